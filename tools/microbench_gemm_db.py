#!/usr/bin/env python3
"""A/B of the 64x64 split-bf16 GEMM with one LDS image per slab (default) against ping-pong images (caphn_tune key 23) on the
step's shapes (live rows: 1660 of 2560), interleaved rounds in one process, median; max |error| vs fp64 for both."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from caphn import _lib, ops  # noqa: E402

SHAPES = [  # name, ta, tb, M, N, K, splitk
    ("logits", 0, 1, 1660, 9684, 200, 1), ("dHs", 0, 0, 1660, 200, 9684, 12), ("dW_fc", 1, 0, 9684, 200, 1660, 3),
    ("fc0 fwd", 0, 1, 6272, 200, 2048, 1), ("dW_fc0", 1, 0, 200, 2048, 6272, 10), ("G", 0, 1, 6272, 600, 200, 1),
    ("Xg", 0, 1, 2560, 600, 200, 1), ("Waf/fc2", 0, 1, 6272, 200, 200, 1), ("dY1", 0, 0, 6272, 200, 200, 1),
    ("dctx", 0, 0, 2560, 200, 600, 1), ("dW_ih", 1, 0, 600, 400, 2560, 10), ("dW_hh", 1, 0, 600, 200, 2560, 10),
    ("dWa", 1, 0, 200, 200, 6272, 24),
]


def main():
    lib = _lib.load()
    dev = "cuda"
    # usage: microbench_gemm_db.py [tune key] [value of the variant]   (default: key 23 = ping-pong LDS images, value 7)
    KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    VAR = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    g = torch.Generator(device=dev).manual_seed(0)
    for name, ta, tb, M, N, K, sk in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), generator=g, device=dev)
        B = torch.randn((N, K) if tb else (K, N), generator=g, device=dev) * 0.07
        ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
        out = torch.zeros(M, N, device=dev)
        res = {0: [], VAR: []}
        err = {}
        for rnd in range(5):
            for mode in (0, VAR):
                lib.caphn_tune(KEY, mode)
                if sk > 1:
                    out.zero_()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); ops.gemm(A, B, bool(ta), bool(tb), out=out, splitk=sk); e.record()
                torch.cuda.synchronize()
                res[mode].append(s.elapsed_time(e) * 1e3)
                err[mode] = float((out.double() - ref).abs().max())
        t0, t1 = float(np.median(res[0][1:])), float(np.median(res[VAR][1:]))
        print(f"{name:9s} M={M:5d} N={N:5d} K={K:5d} sk={sk:2d} | default {t0:7.1f} us err {err[0]:.2e} | key {KEY} = {VAR}: {t1:7.1f} us err {err[VAR]:.2e} | x{t0 / t1:.2f}")
    lib.caphn_tune(KEY, 0)


if __name__ == "__main__":
    main()
