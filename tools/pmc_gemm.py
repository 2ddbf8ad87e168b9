#!/usr/bin/env python3
"""The step's five biggest GEMM shapes in the three operand modes (split on use / pre-split planes / single bf16 product),
a few launches each, for rocprofv3 --pmc passes; tools/pmc_gemm_summary.py turns the counter CSVs into the table kept
under profiles/.  (Counters in their own runs: --pmc with --kernel-trace only.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from caphn import _lib, ops  # noqa: E402

SHAPES = [("fc0_fwd  feat W^T", 0, 1, 6272, 200, 2048, 1), ("dW_fc0   dY1^T feat", 1, 0, 200, 2048, 6272, 16),
          ("dW_fc    dlog^T Hs", 1, 0, 9684, 200, 1664, 1), ("dHs      dlog fc_w", 0, 0, 1664, 200, 9684, 8),
          ("logits   Hs fc_w^T", 0, 1, 1664, 9684, 200, 1)]
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for name, ta, tb, M, N, K, sk in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), generator=g, device="cuda")
    B = torch.randn((N, K) if tb else (K, N), generator=g, device="cuda") * 0.07
    out = torch.zeros(M, N, device="cuda")
    kp = (K + 7) & ~7
    pa = ops.Planes(A, zero_rows=(kp - K) if ta else 0)
    pb = ops.Planes(B, zero_rows=(kp - K) if not tb else 0)
    for _ in range(3):
        ops.gemm(A, B, bool(ta), bool(tb), out=out, splitk=sk)                                   # MODE 0
    for _ in range(3):
        ops.gemm_planes(pa, pb, bool(ta), bool(tb), out=out, splitk=sk, kp=kp if K % 8 else 0)   # MODE 1
    lib.caphn_tune(11, 1)
    for _ in range(3):
        ops.gemm(A, B, bool(ta), bool(tb), out=out, splitk=sk)                                   # MODE 2
    lib.caphn_tune(11, 0)
    torch.cuda.synchronize()
