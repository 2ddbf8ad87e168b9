#!/usr/bin/env python3
"""Do an HBM-bound stream kernel (rank-1 Adam over 115 M params) and MFMA-bound fp32 GEMMs overlap
when issued on two HIP streams?  Prints serial vs concurrent wall time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from caphn import ops  # noqa: E402


def main():
    dev = "cuda"
    rows, k = 240000, 480
    W = torch.randn(rows, k, device=dev) * 0.05
    m = torch.zeros_like(W); v = torch.zeros_like(W)
    g = torch.randn(1, rows, device=dev) * 0.01
    a = torch.randn(1, k, device=dev)
    coef = torch.tensor([1.0, 0.0], device=dev)
    X = torch.randn(6272, 2048, device=dev); Wf = torch.randn(200, 2048, device=dev)
    dl = torch.randn(2560, 9684, device=dev); Wo = torch.randn(9684, 200, device=dev)
    out1 = torch.empty(6272, 200, device=dev); out2 = torch.empty(2560, 200, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def adam():
        ops.adam_rank(W, m, v, g, a, coef, 1e-3, 3)

    def gemms():
        for _ in range(2):
            ops.gemm(X, Wf, False, True, out=out1)
            out2.zero_()
            ops.gemm(dl, Wo, False, False, out=out2, splitk=7)

    def wall(fn, n=5):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            adam()
        with torch.cuda.stream(s2):
            gemms()
        cur.wait_stream(s1); cur.wait_stream(s2)

    for _ in range(2):
        ta, tg, tb = wall(adam), wall(gemms), wall(both)
    print(f"adam alone {ta*1e3:.0f} us, gemms alone {tg*1e3:.0f} us, serial sum {(ta+tg)*1e3:.0f} us, concurrent {tb*1e3:.0f} us")


if __name__ == "__main__":
    main()
