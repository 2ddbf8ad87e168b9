#!/usr/bin/env python3
"""A/B the HBM-streaming kernel variants in ONE process (interleaved rounds, median):
forward GEMV (theta = W2 a + b) and rank-1 fused Adam on the canonical 240000x480 matrix."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from caphn import _lib, ops  # noqa: E402


def timeit(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    lib = _lib.load()
    dev = "cuda"
    rows, k = 240000, 480
    W = torch.randn(rows, k, device=dev) * 0.05
    m = torch.zeros_like(W); v = torch.zeros_like(W)
    g = torch.randn(1, rows, device=dev) * 0.01
    a = torch.randn(1, k, device=dev)
    coef = torch.tensor([1.0, 0.0], device=dev)
    shape = ops.HyperShape(200, [(480, 240000), (240, 120000), (200, 600), (200, 600)])
    p = {n: (torch.rand(s, device=dev) - 0.5) * 0.1 for n, s in shape.param_shapes().items()}
    x = torch.randn(200, device=dev)
    theta = torch.empty(shape.theta_size, device=dev)
    acts = torch.zeros(4096, device=dev)
    res = {}
    for rnd in range(4):
        for var in (0, 3, 6):
            lib.caphn_tune(1, var)
            t = timeit(lambda: ops.adam_rank(W, m, v, g, a, coef, 1e-3, 3))
            res.setdefault(("adam", var), []).append(t)
        for var in range(2):
            lib.caphn_tune(0, var)
            t = timeit(lambda: ops.hyper_forward(shape, p, x, theta=theta, acts=acts))
            res.setdefault(("gemv", var), []).append(t)
    lib.caphn_tune(0, 1); lib.caphn_tune(1, 6)      # back to the defaults
    for (kind, var), ts in sorted(res.items()):
        med = float(np.median(ts[1:]))
        nbytes = 24.0 * rows * k if kind == "adam" else 4.0 * (240000 * 480 + 120000 * 240)
        print(f"{kind} variant {var}: median {med*1e3:8.1f} us  min {min(ts)*1e3:8.1f} us  -> {nbytes/med/1e6:7.1f} GB/s")


if __name__ == "__main__":
    main()
