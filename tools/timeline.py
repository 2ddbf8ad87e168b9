#!/usr/bin/env python3
"""One steady-state training step as a timeline, from a rocprofv3 --kernel-trace CSV (tuning aid).

  python tools/timeline.py gpurun_out/.../trace_kernel_trace.csv [--step -3]

A step starts at the `zero_kernel` that clears the gradient arena.  Prints every kernel of the chosen step with its
start offset, duration, queue and grid, then the chip-idle gaps and the per-kernel-family totals."""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--step", type=int, default=-3, help="which step (index into the list of steps; negative from the end)")
    ap.add_argument("--marker", default="zero_kernel")
    args = ap.parse_args()
    rows = []
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"])) * max(1, int(r["Grid_Size_Z"])),
                         int(r["Workgroup_Size_X"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"])))
    # memory copies (rocprofv3 --memory-copy-trace writes <prefix>_memory_copy_trace.csv next to the kernel trace)
    mc = args.csv.replace("kernel_trace", "memory_copy_trace")
    import os
    if mc != args.csv and os.path.exists(mc):
        with open(mc) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), -1, "COPY " + r.get("Direction", "") + " " + r.get("Size", ""),
                             0, 0, 0, 0))
    rows.sort()
    # the arena zero fill is the biggest zero_kernel launch of a step
    marks = [i for i, r in enumerate(rows) if r[3].startswith(args.marker) and r[4] >= 256]
    # keep only marks separated by > 1 ms
    starts = []
    for i in marks:
        if not starts or rows[i][0] - rows[starts[-1]][0] > 1_000_000:
            starts.append(i)
    print(f"{len(rows)} dispatches, {len(starts)} steps found")
    k = starts[args.step]
    k_end = starts[args.step + 1] if args.step + 1 < 0 or args.step + 1 < len(starts) and args.step >= 0 else len(rows)
    if args.step == -1:
        k_end = len(rows)
    step = rows[k:k_end]
    t0 = step[0][0]
    print(f"step length (marker to marker): {(rows[k_end][0] - t0) / 1e3 if k_end < len(rows) else float('nan'):.1f} us")
    print(f"{'start':>9} {'dur':>8} {'q':>3} {'grid':>6} {'wg':>5} {'vgpr':>4} {'lds':>6}  kernel")
    fam = defaultdict(float)
    busy = []
    for s, e, q, n, grid, wg, vg, lds in step:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {q:3d} {grid:6d} {wg:5d} {vg:4d} {lds:6d}  {n}")
        fam[n] += (e - s) / 1e3
        busy.append((s, e))
    busy.sort()
    gaps, cur = [], busy[0][1]
    for s, e in busy[1:]:
        if s > cur:
            gaps.append((cur - t0, s - cur))
        cur = max(cur, e)
    print(f"\nchip idle inside the step: {sum(g for _, g in gaps) / 1e3:.1f} us in {len(gaps)} gaps; the largest:")
    for at, g in sorted(gaps, key=lambda x: -x[1])[:8]:
        print(f"   at {at / 1e3:8.1f} us: {g / 1e3:6.1f} us")
    print("\nkernel time by family (us, summed over streams):")
    for n, t in sorted(fam.items(), key=lambda x: -x[1])[:25]:
        print(f"   {t:8.1f}  {n}")
    print(f"   {sum(fam.values()):8.1f}  total")


if __name__ == "__main__":
    main()
