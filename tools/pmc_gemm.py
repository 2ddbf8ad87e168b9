#!/usr/bin/env python3
"""A few launches of the step's biggest GEMM shapes, for rocprofv3 --pmc passes (tuning aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from caphn import _lib, ops  # noqa: E402

SHAPES = [("fc0_fwd", 0, 1, 6272, 200, 2048, 1), ("dW_fc", 1, 0, 9684, 200, 2560, 2), ("dHs", 0, 0, 2560, 200, 9684, 7),
          ("logits", 0, 1, 2560, 9684, 200, 1)]
_lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for name, ta, tb, M, N, K, sk in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), generator=g, device="cuda")
    B = torch.randn((N, K) if tb else (K, N), generator=g, device="cuda") * 0.07
    out = torch.zeros(M, N, device="cuda")
    for _ in range(3):
        ops.gemm(A, B, bool(ta), bool(tb), out=out, splitk=sk)
    torch.cuda.synchronize()
