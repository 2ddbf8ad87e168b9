"""Is the unsynchronised trajectory different from the synchronised one beyond run-to-run atomics noise?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "hypernet-image-captioning_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import caphn_oracle as O
from caphn.engine import FusedTrainer
from hypernet_attention import HyperNet

class V:
    def __init__(self): self.w2i = {"<s>": 1, "</s>": 2}; self.i2w = {}
    def __call__(self, w): return 4
    def __len__(self): return 9684

dims = O.Dims()
B, T, P = 32, 12, 49
batch = O.synth_batch(dims, B, T, P, seed=2)
feats, caps = batch["features"].cuda(), batch["captions"].cuda()

def run(sync, steps=6, nxt=True):
    torch.manual_seed(11)
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, V()).cuda()
    tr = FusedTrainer(net, lr=1e-3)
    losses = []
    for _ in range(steps):
        kw = dict(next_features=feats, next_captions=caps) if nxt else {}
        losses.append(tr.step(feats, caps, style_token=4, next_style_token=4, **kw))
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return tr.flat_p.clone(), [float(l.flatten()[0]) for l in losses]

for steps in (1, 2, 6):
    a, la = run(True, steps); a2, la2 = run(True, steps); b, lb = run(False, steps); b2, lb2 = run(False, steps)
    c, lc = run(False, steps, nxt=False)
    print(steps, "sync/sync", (a - a2).abs().max().item(), "sync/ahead", (a - b).abs().max().item(),
          "ahead/ahead", (b - b2).abs().max().item(), "sync/ahead-nonext", (a - c).abs().max().item())
    print("   losses", la[-1], la2[-1], lb[-1], lb2[-1], lc[-1])
