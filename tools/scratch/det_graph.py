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
for cfg in [(), ((4, 1),), ((9, 0),), ((4, 0),), ((4, 0), (9, 0))]:
    for k, v in ((4, 4), (9, 1)):
        lib.caphn_tune(k, v)
    for k, v in cfg:
        lib.caphn_tune(k, v)
    ta, tb, tc = TD._trainer(dims, p), TD._trainer(dims, p), TD._trainer(dims, p)
    la = [float(ta.step(feats, caps, x_style=x)[0]) for _ in range(6)]
    lc = [float(tc.step(feats, caps, x_style=x)[0]) for _ in range(6)]
    lb = [float(tb.step_graphed(feats, caps, x_style=x)[0]) for _ in range(6)]
    print(cfg, "eager==eager", la == lc, torch.equal(ta.flat_p, tc.flat_p), "eager==graph", la == lb, torch.equal(ta.flat_p, tb.flat_p))
    print("  ", [a - b for a, b in zip(la, lb)])
