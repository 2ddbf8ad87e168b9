#!/usr/bin/env python3
"""Per-phase shader-clock profile of the pair forward kernel (workgroup 0), canonical size.  Needs a library built with
-DCAPHN_REC_PROFILE (make -C hypernet-image-captioning_amd/csrc clean all EXTRA=-DCAPHN_REC_PROFILE)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd")); sys.path.insert(0, ROOT)
from caphn import ops, _lib
dev = "cuda"
B, T, P, H = 128, 20, 49, 200
dims = ops.DecDims(B, T, P, 2048, 200, 200, H, 9684)
p = {n: (torch.rand(s, device=dev) - 0.5) * 0.14 for n, s in dims.param_shapes().items()}
feats = torch.relu(torch.randn(B, P, 2048, device=dev)) * 0.45
caps = torch.randint(1, 9684, (B, T), device=dev)
ws = ops.decoder_workspace(dims, dev)
# (pair, forward cache mode of caphn_tune key 16): 2 = everything on chip when it fits, 1 = partial, single-workgroup kernels last
for pair, cache in ((1, 2), (1, 1), (0, 1)):
    _lib.load().caphn_tune(9, pair)
    _lib.load().caphn_tune(16, cache)
    print(f"---- pair={pair} cache={cache}")
    for _ in range(3):
        logits, _ = ops.decoder_forward(dims, p, feats, caps, ws)
        lo, dl = ops.cross_entropy_fwd_bwd(logits, caps, 0)
        grads = {n: torch.empty(s, device=dev) for n, s in dims.param_shapes().items()}
        ops.decoder_backward(dims, p, feats, caps, dl, grads, ws)
    torch.cuda.synchronize()
    import ctypes as C
    cd = dims.c()
    ptr = _lib.load().caphn_decoder_profile_ptr(C.byref(cd), C.c_void_p(ws.data_ptr()))
    off = (ptr - ws.data_ptr()) // 8
    prof = ws.view(torch.int64)[off:off + 16].cpu().tolist()
    fw = prof[:8]
    if pair:
        names = ["-", "recv h + barrier", "A matvec (my rows)", "B partial scores", "X1 exchange e",
                 "C softmax", "D1 alpha.G", "D2 gates + send"]
    else:
        names = ["A matvec", "B scores", "C softmax", "D1 alpha.G", "D2 gates", "-", "-", "-"]
    print(f"pair={pair}: forward cycles per step (workgroup 0): total {sum(fw) / T:.0f}")
    for n, v in zip(names, fw):
        print(f"   {n:26s} {v / T:9.0f}")
    bw = prof[8:16]
    if pair:
        namesb = ["loads + cell backward", "d alpha partial", "X1 + softmax bwd", "d(U_a h) partial", "d(U_a h) sum", "tmatvec partner cols + send",
                  "tmatvec own cols", "recv + dh update"]
    else:
        namesb = ["P1 cell pointwise", "P2 dalpha=G.dgi", "P3 softmax bwd", "P4 d(uah) partial", "P4b sum", "P5 matvec^T", "P6 sum", "-"]
    print(f"pair={pair}: backward cycles per step (workgroup 0): total {sum(bw) / T:.0f}")
    for n, v in zip(namesb, bw):
        print(f"   {n:28s} {v / T:9.0f}")
