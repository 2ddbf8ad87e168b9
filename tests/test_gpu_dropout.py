"""Dropout in the recurrent decoders: `h = self.drop(h)` inside the time loop (models/decoderlstm.py:44,104 for AttentionGru,
:254 for AttentionLstm whose constructor default is p = 0.5) -- the dropped h_t feeds fc, the next step's cell and the next
step's attention.  The kernels draw the mask from a counter-based hash, not from torch's Philox stream, so the reference's
masks cannot be reproduced (parity with the reference is pinned with dropout off); here the oracle is given the kernel's own
mask -- ops.dropout(ones, p, seed) is the same hash -- and logits, loss and every gradient must agree."""
import pytest
import torch
from torch.nn import functional as F

from helpers import maxdiff
from oracle import caphn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEED = 0x5DEECE66D1234567


def _mask(B, T, H, p, seed):
    from caphn import ops
    return ops.dropout(torch.ones(B * T * H, device=DEV), p, seed).view(B, T, H).cpu()


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_module_dropout_matches_oracle_given_the_kernel_mask(cell, monkeypatch):
    from caphn import functional as CF
    from models.decoderlstm import AttentionGru, AttentionLstm
    monkeypatch.setattr(CF, "next_seed", lambda: SEED)
    dims = O.Dims(D=24, F=12, E=10, H=12, V=50, he=6, cell=cell)
    p = O.init_params(dims, seed=5)
    B, T, P, pd = 4, 7, 6, (0.3 if cell == "gru" else 0.5)
    batch = O.synth_batch(dims, B, T, P, seed=6)
    torch.manual_seed(0)
    cw = {n: (torch.rand(s) - 0.5) * 0.5 for n, s in dims.cell_param_shapes()}
    if cell == "gru":
        m = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=pd)
        sd = {k[len("captioner."):]: v for k, v in p.items() if k.startswith("captioner.")}
        sd.update({"gru." + n: v for n, v in cw.items()})
    else:
        m = AttentionLstm(dims.D, dims.E, dims.H, dims.V, feature_out=dims.F)           # reference default p = 0.5
        assert m.drop.p == 0.5
        sd = {k[len("captioner."):].replace("embed.", "embeddings."): v for k, v in p.items() if k.startswith("captioner.")}
        sd.update({"lstm." + n: v for n, v in cw.items()})
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    logits, _ = m(feats, caps, 0.0) if cell == "gru" else m(caps, feats, 0.0)
    loss = F.cross_entropy(logits.view(-1, dims.V), caps.view(-1), ignore_index=0)
    loss.backward()
    mask = _mask(B, T, dims.H, pd, SEED)
    frac = float((mask == 0).float().mean())
    assert abs(frac - pd) < 0.12 and set(mask.unique().tolist()) <= {0.0, float(torch.tensor(1.0 / (1.0 - pd)))}
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    cq = {k: v.clone().requires_grad_(True) for k, v in cw.items()}
    ref, _ = O.decoder_forward(dims, q, cq, batch["features"], batch["captions"], drop_mask=mask)
    rl = O.caption_loss(ref, batch["captions"])
    rl.backward()
    assert maxdiff(logits.detach().cpu(), ref.detach()) < 3e-6 and abs(float(loss) - float(rl)) < 3e-6
    got = dict(m.named_parameters())
    for k, v in q.items():
        if k.startswith("captioner."):
            n = k[len("captioner."):]
            if cell == "lstm":
                n = n.replace("embed.", "embeddings.")
            assert maxdiff(got[n].grad.cpu(), v.grad) < 3e-6, k
    for k, v in cq.items():
        assert maxdiff(got[cell + "." + k].grad.cpu(), v.grad) < 3e-6, k
    # eval mode: dropout is the identity
    m.eval()
    with torch.no_grad():
        le, _ = m(feats, caps, 0.0) if cell == "gru" else m(caps, feats, 0.0)
    r0, _ = O.decoder_forward(dims, p, cw, batch["features"], batch["captions"])
    assert maxdiff(le.cpu(), r0) < 3e-6 and maxdiff(le.cpu(), logits.detach().cpu()) > 1e-3


def test_fused_engine_with_decoder_dropout():
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionGru
    from caphn.engine import FusedTrainer

    class V:
        w2i = {"<pad>": 0}

        def __call__(self, w):
            return 3
    dims = O.Dims(D=24, F=12, E=12, H=12, V=40, he=5)
    p = O.init_params(dims, seed=3)
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, V(), cc=True, hyper_emb=dims.he)
    net.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.25)
    net.load_state_dict(p, strict=False)
    net = net.to(DEV).train()
    tr = FusedTrainer(net, lr=1e-3, max_norm=5.0)
    batch = O.synth_batch(dims, B=3, T=6, P=5, seed=100)
    x = torch.zeros(dims.he); x[2] = 1.0
    loss = tr.forward_backward(batch["features"].to(DEV), batch["captions"].to(DEV), x_style=x.to(DEV), validate=True)
    mask = _mask(3, 6, dims.H, 0.25, tr.last_dropout_seed)
    ref_loss, _, _, _, g = O.forward_backward(dims, p, x, batch["features"], batch["captions"], drop_mask=mask)
    assert abs(float(loss[0]) - float(ref_loss)) < 3e-6
    assert maxdiff(tr.flat_g[:tr.theta_size].cpu(), g["dtheta"]) < 3e-6
    for n in ("captioner.fc.weight", "captioner.attention.U_a.weight", "captioner.feature_fc.0.weight", "hn_base.0.weight"):
        assert maxdiff(tr.grad(n).cpu(), g[n]) < 3e-6, n
    s1 = tr.last_dropout_seed
    tr.optimizer_step()
    tr.step(batch["features"].to(DEV), batch["captions"].to(DEV), x_style=x.to(DEV))
    assert tr.last_dropout_seed != s1                      # a fresh mask every step
    net.eval()                                             # eval mode: no dropout
    l_eval = tr.forward_backward(batch["features"].to(DEV), batch["captions"].to(DEV), x_style=x.to(DEV))
    p2 = {k: v.detach().cpu() for k, v in net.state_dict().items() if not k.startswith("captioner.gru.")}
    r2, *_ = O.forward_backward(dims, p2, x, batch["features"], batch["captions"])
    assert abs(float(l_eval[0]) - float(r2)) < 3e-6
