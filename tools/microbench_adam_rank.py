#!/usr/bin/env python3
"""Rank-R Adam pass over the canonical head-0 matrix (240000 x 480) for R = 1, 2, 4, 8 ranks' factors, with and without
the fused next-theta GEMV: what the N-GPU data-parallel step costs in this kernel (only 1 GPU is available to us, but
the kernel's R dependence can be measured on it)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from caphn import ops  # noqa: E402


def main(rows=240000, k=480):
    dev = "cuda"
    W = torch.randn(rows, k, device=dev) * 0.01
    m = torch.zeros_like(W); v = torch.zeros_like(W)
    coef = torch.tensor([1.0, 1.0], device=dev)
    sc = torch.tensor(ops.adam_scalars(1e-3, (0.9, 0.999), 1), device=dev, dtype=torch.float32)
    na = torch.randn(k, device=dev); nb = torch.zeros(rows, device=dev); nt = torch.empty(rows, device=dev)
    for R, doms in ((1, 1), (2, 2), (4, 4), (8, 8)):
        g = torch.randn(R, rows, device=dev) * 1e-3
        a = torch.randn(doms, k, device=dev)[torch.arange(R) % doms].contiguous()
        for fused in (False, True):
            kw = dict(next_a=na, next_bias=nb, next_theta=nt) if fused else {}
            ts = []
            for _ in range(6):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); ops.adam_rank(W, m, v, g, a, coef, 1e-3, 1, dev_scalars=sc, **kw); e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e))
            t = float(np.median(ts[1:]))
            print(f"R={R} fused_gemv={int(fused)}: {t*1e3:7.1f} us  {24.0*rows*k/t/1e9:6.2f} TB/s")


if __name__ == "__main__" and "--gram" not in sys.argv:
    main()
    print("head 1 (120000 x 240):")
    main(120000, 240)


def gram():
    """Global-norm term of the rank-R gradients (two R x R Gram matrices per head) for the canonical four heads."""
    dev = "cuda"
    heads = [(480, 240000), (240, 120000), (200, 600), (200, 600)]
    for R in (1, 2, 4, 8):
        pairs = [(torch.randn(R, w, device=dev), torch.randn(R, k, device=dev)) for k, w in heads]
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        ts = []
        for _ in range(6):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.rank_sumsq_multi(pairs, acc); e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ref = sum(float(((g.double().t() @ a.double()) ** 2).sum()) for g, a in pairs[2:])   # dense check on the small heads
        print(f"gram R={R}: {float(np.median(ts[1:]))*1e3:7.1f} us")


if __name__ == "__main__" and "--gram" in sys.argv:
    gram()
