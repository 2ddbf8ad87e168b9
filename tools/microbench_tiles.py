#!/usr/bin/env python3
"""Tile size x split-K sweep of the split-bf16 GEMM on the training step's shapes (caphn_tune key 12 forces the tile)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
import torch  # noqa: E402
from caphn import ops, _lib  # noqa: E402

DEV = "cuda:0"
SHAPES = [("logits  Hs fc_w^T", False, True, 1664, 9684, 200), ("dHs     dlog fc_w", False, False, 1664, 200, 9684),
          ("dW_fc   dlog^T Hs", True, False, 9684, 200, 1664), ("dW_fc0  dY1^T feat", True, False, 200, 2048, 6272),
          ("fc0 fwd feat W^T", False, True, 6272, 200, 2048), ("G       f W_ih^T", False, True, 6272, 600, 200),
          ("dctx    dgi W_ih", False, False, 2560, 200, 600), ("dW_ih   dgi^T [Xe|ctx]", True, False, 600, 400, 2560),
          ("dW_hh   dgh^T Hp", True, False, 600, 200, 2560), ("dY1     df fc2", False, False, 6272, 200, 200)]


def timeit(fn, n=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    lib = _lib.load()
    torch.manual_seed(0)
    sks = [1, 2, 4, 8, 16, 32]
    print(f"{'shape':24s} {'M':>5} {'N':>5} {'K':>5} tile " + " ".join(f"sk={k:<4d}" for k in sks))
    for name, ta, tb, M, N, K in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device=DEV)
        B = torch.randn((N, K) if tb else (K, N), device=DEV)
        out = torch.zeros(M, N, device=DEV)
        for tile in (64, 128):
            lib.caphn_tune(12, tile)
            row = []
            for sk in sks:
                if sk > max(1, K // 256):
                    row.append("     -")
                    continue
                row.append(f"{timeit(lambda: ops.gemm(A, B, ta, tb, out=out, splitk=sk)):6.1f}")
            print(f"{name:24s} {M:5d} {N:5d} {K:5d} {tile:4d} " + " ".join(row))
        lib.caphn_tune(12, 0)


if __name__ == "__main__":
    main()
