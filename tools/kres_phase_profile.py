#!/usr/bin/env python3
"""Shader-clock phase stamps of workgroup (0, 0) of the K-resident GEMM (csrc/gemm_kres.hip) on the vocabulary-logits shape.
Needs the profiling build: (cd hypernet-image-captioning_amd/csrc && make prof); CAPHN_LIB_PATH=.../libcaphn_prof.so python tools/kres_phase_profile.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd"))
from caphn import _lib, ops  # noqa: E402


def main():
    lib = _lib.load()
    f = lib.caphn_debug_kres_prof
    f.restype = C.c_int
    f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.caphn_tune(36, 1)
    for name, M, N, K in [("logits", 1660, 9684, 200)]:
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.07
        out = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm(A, B, False, True, out=out)
        torch.cuda.synchronize()
        f(None, 1)
        reps = 10
        for _ in range(reps):
            ops.gemm(A, B, False, True, out=out)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 8)()
        f(buf, 0)
        v = [x / reps for x in buf]
        tiles = v[6]
        names = ["prologue (A fragments, first loads)", "wait loads + split + LDS stores", "barrier after stage", "next loads + fragment reads + MFMAs",
                 "barrier after MFMAs", "epilogue"]
        print(f"{name}: M={M} N={N} K={K}: {tiles:.0f} tiles per workgroup; cycles of workgroup (0,0) wave 0:")
        print(f"   {names[0]:44s} {v[0]:9.0f}")
        for i in range(1, 6):
            print(f"   {names[i]:44s} {v[i]:9.0f}   ({v[i] / tiles:7.0f} per tile)")
        print(f"   total {sum(v[:6]):9.0f} cycles = {sum(v[:6]) / 2.4e3:.1f} us at 2.4 GHz")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.gemm(A, B, False, True, out=out)
        b.record(); torch.cuda.synchronize()
        print(f"   kernel: {a.elapsed_time(b) / 20 * 1e3:.1f} us per launch")


if __name__ == "__main__":
    main()
