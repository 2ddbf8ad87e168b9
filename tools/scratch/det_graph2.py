import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hypernet-image-captioning_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import caphn_oracle as O
from caphn import _lib
import test_gpu_determinism as TD
lib = _lib.load()
DEV = "cuda:0"
dims = O.Dims(D=64, F=32, E=24, H=32, V=300, he=8)
p = O.init_params(dims, seed=31)
batch = O.synth_batch(dims, B=16, T=9, P=12, seed=32)
feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
x = torch.zeros(dims.he, device=DEV); x[1] = 1.0
assert lib.caphn_tune(13, dims.V) == 0
ta, tb = TD._trainer(dims, p), TD._trainer(dims, p)
for i in range(5):
    la = float(ta.step(feats, caps, x_style=x)[0]); lb = float(tb.step_graphed(feats, caps, x_style=x)[0])
    torch.cuda.synchronize()
    dp = (ta.flat_p - tb.flat_p).abs()
    dg = (ta.flat_g - tb.flat_g).abs()
    print(i, la == lb, "p diff", float(dp.max()), "g diff", float(dg.max()), "coef", ta._coef.tolist(), tb._coef.tolist())
    if float(dg.max()) > 0 or float(dp.max()) > 0:
        for n, (o, sz) in list(ta.offs.items()):
            a = dg[o:o + sz].max().item() if o + sz <= dg.numel() else -1
            b = dp[o:o + sz].max().item() if o + sz <= dp.numel() else -1
            if a > 0 or b > 0: print("    ", n, "g", a, "p", b)
        for j in range(len(ta.W2)):
            print("     W2", j, float((ta.W2[j].data - tb.W2[j].data).abs().max()), "m", float((ta.W2_m[j] - tb.W2_m[j]).abs().max()))
        break
