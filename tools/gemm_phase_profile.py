#!/usr/bin/env python3
"""Where a K-slab of the split-bf16 GEMM spends its time: shader-clock stamps of workgroup (0,0,0), wave 0, per phase of the
main loop (tuning aid).  Needs a library built with -DCAPHN_GEMM_PROFILE
(make -C hypernet-image-captioning_amd/csrc clean all EXTRA=-DCAPHN_GEMM_PROFILE).  Shapes: the step's GEMMs."""
import ctypes as C
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd")); sys.path.insert(0, ROOT)
from caphn import ops, _lib
lib = _lib.load()
fn = lib.caphn_debug_gemm_prof
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
dev = "cuda"
names = ["prologue", "issue look-ahead loads", "fragment reads + MFMA", "barrier after multiply", "wait loads + split + LDS store",
         "barrier after stage"]
shapes = [("vocab logits  NT  2560 x 9684 x 200", 2560, 9684, 200, False, True, 1),
          ("G             NT  6272 x  600 x 200", 6272, 600, 200, False, True, 1),
          ("dHs           NN  2560 x  200 x 9684 (split-K 7)", 2560, 200, 9684, False, False, 7),
          ("dW_fc (vocab) TN  9684 x  200 x 2560 (split-K 2)", 9684, 200, 2560, True, False, 2),
          ("dW_ih         TN   600 x  400 x 2560 (split-K 10)", 600, 400, 2560, True, False, 10)]
shapes += [("vocab logits, C rows padded to 9728 floats (128-byte aligned rows)", 2560, 9684, 200, False, True, 1, 9728),
           ("G, C rows padded to 608 floats", 6272, 600, 200, False, True, 1, 608)]
for sh in shapes:
    label, M, N, K, ta, tb, sk = sh[:7]
    ldc = sh[7] if len(sh) > 7 else N
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.zeros(M, ldc, device=dev)[:, :N]
    for _ in range(3):
        ops.gemm(a, b, ta, tb, out=out, splitk=sk)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 10)()
    fn(buf, 1)
    reps = 20
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        ops.gemm(a, b, ta, tb, out=out, splitk=sk)
    t1.record(); torch.cuda.synchronize()
    fn(buf, 1)
    v = list(buf)
    slabs, launches = v[6], v[7]
    print(f"{label}: {t0.elapsed_time(t1) / reps * 1e3:.1f} us per launch; workgroup 0: {slabs // max(launches, 1)} slabs")
    print(f"   {names[0]:34s} {v[0] / max(launches, 1):9.0f} cycles per launch")
    for n, c in zip(names[1:], v[1:6]):
        print(f"   {n:34s} {c / max(slabs, 1):9.0f} cycles per slab")
    print(f"   {'  of which: pure wait for loads':34s} {v[9] / max(slabs, 1):9.0f} cycles per slab (counted apart from the stage line above)")
    print(f"   {'sum per slab':34s} {(sum(v[1:6]) + v[9]) / max(slabs, 1):9.0f}")
    tot = v[8] / max(launches, 1)
    print(f"   {'workgroup 0 lifetime':34s} {tot:9.0f} cycles; outside prologue and loop (epilogue, setup) "
          f"{tot - v[0] / max(launches, 1) - sum(v[1:6]) / max(launches, 1):9.0f}")
