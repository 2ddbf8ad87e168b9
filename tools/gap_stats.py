#!/usr/bin/env python3
"""Idle time in front of one kernel over ALL steps of a rocprofv3 kernel trace: for every dispatch whose name contains `pattern`,
the gap between the end of the latest earlier dispatch (any queue) and its start.  Tells a per-step cost from a one-off (the bench
times every 4th launch of the dominant kernel with an event pair, which itself opens a bubble)."""
import csv
import sys

import numpy as np

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
pat = sys.argv[2] if len(sys.argv) > 2 else "adam_rank_kernel<2"
gaps = []
last_end = 0
for s, e, n in rows:
    if pat in n and last_end:
        gaps.append((s - last_end) / 1e3)
    last_end = max(last_end, e)
g = np.array(gaps[5:])
print(f"{pat}: {len(g)} launches; gap before it: median {np.median(g):.1f} us, p10 {np.percentile(g, 10):.1f}, p90 {np.percentile(g, 90):.1f}")
print("every 4th:", " ".join(f"{x:.0f}" for x in g[:24]))
