#!/usr/bin/env python3
"""Accuracy (vs fp64) and time of the two GEMM back ends on the step's real shapes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from caphn import _lib, ops  # noqa: E402

SHAPES = [  # name, ta, tb, M, N, K, splitk candidates
    ("fc0 fwd", 0, 1, 6272, 200, 2048, (1, 2, 4, 6)), ("logits fwd", 0, 1, 2560, 9684, 200, (1,)),
    ("dW_fc", 1, 0, 9684, 200, 2560, (1, 2, 4)), ("dHs", 0, 0, 2560, 200, 9684, (7, 14, 20)),
    ("dW_fc0", 1, 0, 200, 2048, 6272, (8, 16, 24)), ("G", 0, 1, 6272, 600, 200, (1,)), ("Waf", 0, 1, 6272, 200, 200, (1, 2)),
    ("Xg", 0, 1, 2560, 600, 200, (1,)), ("dW_hh", 1, 0, 600, 200, 2560, (10,)), ("dY1", 0, 0, 6272, 200, 200, (1, 2)),
    ("dW_fc2", 1, 0, 200, 200, 6272, (16, 32)), ("dctx->df", 0, 0, 2560, 200, 600, (1, 2, 4)),
]


CATR_SHAPES = [  # BASELINE config 5 at bs 64: T*bs = 8192 target rows, 49*bs = 3136 memory rows, d 256, ff 2048, V 30522
    ("logits fwd", 0, 1, 8192, 30522, 512, (1,)), ("logits dgrad", 0, 0, 8192, 512, 30522, (1, 4)), ("logits wgrad", 1, 0, 30522, 512, 8192, (1,)),
    ("ffn1 fwd", 0, 1, 8192, 2048, 256, (1,)), ("ffn2 fwd", 0, 1, 8192, 256, 2048, (1, 2, 4)), ("ffn1 dgrad", 0, 0, 8192, 256, 2048, (1, 2, 4)),
    ("ffn2 dgrad", 0, 0, 8192, 2048, 256, (1,)), ("ffn1 wgrad", 1, 0, 2048, 256, 8192, (1, 4, 8)), ("proj fwd", 0, 1, 8192, 256, 256, (1, 2)),
    ("proj wgrad", 1, 0, 256, 256, 8192, (8, 16, 32)), ("enc ffn1 fwd", 0, 1, 3136, 2048, 256, (1,)), ("input_proj", 0, 1, 3136, 256, 2048, (1, 2, 4)),
]


def main():
    global SHAPES
    if "--catr" in sys.argv:
        SHAPES = CATR_SHAPES
    lib = _lib.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for name, ta, tb, M, N, K, sks in SHAPES:
      for sk in sks:
        A = torch.randn((K, M) if ta else (M, K), generator=g, device=dev)
        B = torch.randn((N, K) if tb else (K, N), generator=g, device=dev) * 0.07
        ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
        out = torch.zeros(M, N, device=dev)
        line = f"{name:11s} M={M:5d} N={N:5d} K={K:5d} sk={sk:2d}"
        for mode in (0, 1):
            lib.caphn_tune(2, 1); lib.caphn_tune(6, 1); lib.caphn_tune(7, mode)
            ts = []
            for _ in range(6):
                if sk > 1:
                    out.zero_()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); ops.gemm(A, B, bool(ta), bool(tb), out=out, splitk=sk); e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e))
            err = float((out.double() - ref).abs().max())
            rel = err / float(ref.abs().max())
            t = float(np.median(ts[1:])) * 1e3
            line += f" | {'generic' if mode == 0 else 'fast   '}: {t:7.1f} us {2.0*M*N*K/t/1e6:6.1f} TF err {err:.2e} (rel {rel:.1e})"
        print(line)
    lib.caphn_tune(2, 1); lib.caphn_tune(6, 1); lib.caphn_tune(7, 1)


if __name__ == "__main__":
    main()
