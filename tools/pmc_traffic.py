#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py -> JSON (stdout).

gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE is reported in KB and counts a wide coalesced read at
HALF its bytes (x 1024 x 2); WRITE_SIZE in KB is exact for 16-byte-per-lane stores and float atomics (x 1024).  Means over the
launches of each kernel in the last two steps of the run; `algorithmic` is the byte count DESIGN.md section 4 assigns to one launch."""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path):
    d = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            d[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return d


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)


# (substring of the kernel symbol, grid size in threads or None, label, algorithmic bytes per launch)
B, T, P, H, V, LIVE = 128, 20, 49, 200, 9684, None
WANT = [
    ("adam_rank_kernel<2, true, true, 2, false>", None, "adam_rank head 0 (240000 x 480: W,m,v read + written)", 24.0 * 240000 * 480),
    ("adam_rank_kernel<4, true, true, 1, false>", 4096 * 256, "adam_rank head 1 (120000 x 240)", 24.0 * 120000 * 240),
    ("gemv_t_partial_kernel<2>", None, "hypernet VJP: W2^T dtheta over both big heads (576 MB read)", 4.0 * (240000 * 480 + 120000 * 240)),
    ("rec_pair_fwd_kernel", None, "recurrent forward (pair): G slab + saved activations", None),
    ("rec_pair_bwd_kernel", None, "BPTT (pair)", None),
    ("gemm_bf16x3_kernel<64, 64, false, true, 0>", 6080 * 256, "vocabulary logits GEMM (live rows)", None),
    ("gemm_bf16x3_kernel<64, 64, true, false, 0>", 1824 * 256, "vocabulary weight gradient GEMM (live rows)", None),
    ("gemm_bf16x3_kernel<64, 64, false, false, 0>", 1280 * 256, "dHs = dlogits W_fc GEMM (live rows)", None),
    ("adam_dense_clip_kernel", None, "dense arena Adam (5.2 M parameters: p,m,v,g read, p,m,v written)", 28.0 * 5.2e6),
    ("ce_row_reg_kernel", None, "cross entropy rows (in place)", None),
]


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `bench.py --steps 3 --warmup 2 --no-spinup`; "
                     "bytes = FETCH_SIZE KB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KB x 1024; mean over the launches "
                     "of the last two steps", "kernels": []}
    for pat, grid, label, algo in WANT:
        fk = [k for k in fetch if pat in short(k[0]) and (grid is None or k[1] == grid)]
        wk = [k for k in write if pat in short(k[0]) and (grid is None or k[1] == grid)]
        if not fk or not wk:
            continue
        fv = [v for k in fk for v in fetch[k][-2:]]
        wv = [v for k in wk for v in write[k][-2:]]
        fb = sum(fv) / len(fv) * 1024.0 * 2.0
        wb = sum(wv) / len(wv) * 1024.0
        e = {"kernel": short(fk[0][0]), "grid_threads": fk[0][1], "what": label, "fetch_bytes_corrected": fb, "write_bytes": wb,
             "traffic_bytes": fb + wb}
        if algo:
            e["algorithmic_bytes"] = algo
            e["traffic_over_algorithmic"] = (fb + wb) / algo
        out["kernels"].append(e)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
