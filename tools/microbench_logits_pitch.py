import sys, torch, numpy as np
sys.path.insert(0, "hypernet-image-captioning_amd")
from caphn import ops
dev = "cuda"
M, V, H = 1660, 9684, 200
g = torch.Generator(device=dev).manual_seed(0)
Hs = torch.randn(M, H, generator=g, device=dev); W = torch.randn(V, H, generator=g, device=dev) * 0.07
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))
for LD in (9684, 9696, 9728):
    buf = torch.empty(M, LD, device=dev)
    out = buf[:, :V]
    t_log = bench(lambda: ops.gemm(Hs, W, False, True, out=out))
    dl = torch.randn(M, LD, device=dev)[:, :V]
    dHs = torch.zeros(M, H, device=dev)
    t_dhs = bench(lambda: (dHs.zero_(), ops.gemm(dl, W, False, False, out=dHs, splitk=12)))
    dW = torch.zeros(V, H, device=dev)
    t_dw = bench(lambda: (dW.zero_(), ops.gemm(dl, Hs, True, False, out=dW, splitk=3)))
    print(f"row pitch {LD}: logits {t_log:6.1f} us   dHs (+zero) {t_dhs:6.1f} us   dW_fc (+zero) {t_dw:6.1f} us")
