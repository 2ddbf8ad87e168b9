#!/usr/bin/env python3
"""When do the workgroups of one GEMM launch start and end?  (tuning aid; library built with -DCAPHN_GEMM_PROFILE)  Each
workgroup stamps the 100 MHz wall clock at entry and after its epilogue; this prints the launch ramp, the lifetimes and how many
workgroups are alive over time."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd")); sys.path.insert(0, ROOT)
from caphn import ops, _lib
lib = _lib.load()
fn = lib.caphn_debug_gemm_wgtimes
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
dev = "cuda"
shapes = [("G   NT 6272x600x200", 6272, 600, 200, False, True, 1, 980), ("logits NT 2560x9684x200", 2560, 9684, 200, False, True, 1, 1520),
          ("dHs NN 2560x200x9684 sk8", 2560, 200, 9684, False, False, 8, 1280), ("dW_fc0 TN 200x2048x6272 sk10", 200, 2048, 6272, True, False, 10, 1280)]
for label, M, N, K, ta, tb, sk, nwg in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.zeros(M, N, device=dev)
    for _ in range(3): ops.gemm(a, b, ta, tb, out=out, splitk=sk)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); ops.gemm(a, b, ta, tb, out=out, splitk=sk); t1.record(); torch.cuda.synchronize()
    buf = (C.c_uint64 * (2 * 8192))()
    fn(buf, nwg)
    t = np.array(buf[:2 * nwg], dtype=np.float64).reshape(nwg, 2) / 100.0     # us
    t -= t[:, 0].min()
    st, en = t[:, 0], t[:, 1]
    life = en - st
    print(f"{label}: {nwg} workgroups, event time {t0.elapsed_time(t1) * 1e3:.1f} us; first start 0, last start {st.max():.1f} us, last end {en.max():.1f} us")
    print(f"   lifetime us: min {life.min():.1f} median {np.median(life):.1f} max {life.max():.1f}; starts: 50% by {np.percentile(st, 50):.1f}, 90% by {np.percentile(st, 90):.1f}, 99% by {np.percentile(st, 99):.1f}")
    grid = np.linspace(0, en.max(), 9)[1:-1]
    print("   alive at t:", ", ".join(f"{g:.1f}us:{int(((st <= g) & (en > g)).sum())}" for g in grid))
