#!/usr/bin/env python3
"""Stand-alone timing of caphn_hyper_forward_acts at the canonical sizes: three launches (caphn_tune(28, 0)) vs one (28, 1),
and of caphn_hyper_backward's tail (caphn_tune(27, 0/1)) -- HIP events over 200 calls each, alone on the chip."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hypernet-image-captioning_amd")):
    sys.path.insert(0, p)
from caphn import ops, _lib  # noqa: E402

DEV = "cuda"


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    he, heads = 200, [(480, 240000), (240, 120000), (200, 600), (200, 600)]
    shape = ops.HyperShape(he, heads)
    torch.manual_seed(0)
    pd = {n: ((torch.rand(s, device=DEV) - 0.5) * 0.2) for n, s in shape.param_shapes().items()}
    x = torch.randn(he, device=DEV)
    theta, acts = ops.hyper_forward(shape, pd, x)
    lib = _lib.load()
    for mode in (0, 1, 0, 1):
        lib.caphn_tune(28, mode)
        print(f"hyper_forward_acts, caphn_tune(28, {mode}): {timeit(lambda: ops.hyper_forward_acts(shape, pd, x, acts)):7.1f} us per call")
    lib.caphn_tune(28, 1)
    dth = torch.randn(theta.numel(), device=DEV)
    grads = {n: torch.empty(s, device=DEV) for n, s in shape.param_shapes().items() if not n.endswith(".2.weight")}
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for mode in (0, 1, 0, 1):
        lib.caphn_tune(27, mode)
        print(f"hyper_backward (VJP + tail), caphn_tune(27, {mode}): {timeit(lambda: ops.hyper_backward(shape, pd, dth, acts, grads, want_x=True, ws=ws), 50):7.1f} us per call")
    lib.caphn_tune(27, 0)


if __name__ == "__main__":
    main()
