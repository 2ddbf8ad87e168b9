import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "hypernet-image-captioning_amd"))
from caphn import _lib, ops
lib = _lib.load(); lib.caphn_tune(2, 1)
dev = "cuda"
def run(ta, tb, M, N, K, sk, extra):
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.zeros(M, N, device=dev)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = lib.caphn_gemm_f32(ta, tb, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), N, None, None, 0, extra, sk, _lib.stream_ptr())
        e.record(); torch.cuda.synchronize(); assert rc == 0
        ts.append(s.elapsed_time(e))
    return float(np.median(ts[1:])) * 1e3
for name, ta, tb, M, N, K, sk in [("logits", 0, 1, 2560, 9684, 200, 1), ("dHs", 0, 0, 2560, 200, 9684, 7), ("fc0", 0, 1, 6272, 200, 2048, 1)]:
    base = run(ta, tb, M, N, K, sk, 0)
    res = {k: run(ta, tb, M, N, K, sk, v) for k, v in [("noMFMA", 1 << 16), ("noEpiStore", 1 << 17), ("noStage", 1 << 18), ("noGload", 1 << 19), ("noMFMA+noStage", (1<<16)|(1<<18)), ("noGload+noStage+noMFMA", (1<<16)|(1<<18)|(1<<19)), ("all4", 15 << 16)]}
    print(name, f"base {base:.1f} us |", " ".join(f"{k}={v:.1f}" for k, v in res.items()))
