"""Optimiser-state checkpoint / resume of the fused trainers (the reference checkpoints through Lightning's
ModelCheckpoint, cc_train_hypernet.py:393, which saves torch.optim.Adam's state_dict): save after two steps, load into
a freshly built model + trainer, two more steps == four uninterrupted steps; and the saved dict loads into a
torch.optim.Adam built over the reference's parameter list (configure_optimizers)."""
import numpy as np
import pytest
import torch

from helpers import TINY_DIMS, load_case, maxdiff, style_args
from oracle import caphn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _V:
    w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "factual": 4}

    def __call__(self, w):
        return self.w2i.get(w, 3)


def _net(dims, p, cc):
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionGru
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, _V(), cc=cc, hyper_emb=dims.he)
    net.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
    net.load_state_dict(p, strict=False)
    return net.to(DEV)


def _flat(tr):
    return torch.cat([tr.flat_p] + [w.data.flatten() for w in tr.W2]).cpu()


@pytest.mark.parametrize("name", ["gru_tiny_cc", "gru_tiny_flickr"])
def test_fused_trainer_resume_equals_uninterrupted(name):
    from caphn.engine import FusedTrainer
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    kw = dict(style_token=tok, next_style_token=tok) if tok is not None else dict(x_style=x.to(DEV))

    a = FusedTrainer(_net(dims, p, tok is None), lr=1e-3, max_norm=float(g["clip_max_norm"]))
    for _ in range(4):
        a.step(feats, caps, **kw)
    b = FusedTrainer(_net(dims, p, tok is None), lr=1e-3, max_norm=float(g["clip_max_norm"]))
    for _ in range(2):
        b.step(feats, caps, **kw)
    ckpt = {"model": {k: v.detach().cpu().clone() for k, v in b.net.state_dict().items()}, "opt": b.state_dict()}
    assert int(ckpt["opt"]["state"][0]["step"]) == 2
    c_net = _net(dims, p, tok is None)
    c = FusedTrainer(c_net, lr=123.0, max_norm=1e9)                 # hyper-parameters come back from the checkpoint too
    c.step(feats, caps, **kw)                                       # a trainer that already ran (prefetched theta pending)
    c_net.load_state_dict(ckpt["model"], strict=False)              # torch-side write into the arena views
    c.load_state_dict(ckpt["opt"])
    assert c.step_count == 2 and c.lr == 1e-3
    for _ in range(2):
        c.step(feats, caps, **kw)
    # not bit-exact: fp32 atomics in the split-K weight gradients, and Adam turns a last-bit difference of a ~0 gradient
    # into up to lr * 1e-3 per step
    assert maxdiff(_flat(a), _flat(c)) < 2e-5
    assert maxdiff(a.flat_m.cpu(), c.flat_m.cpu()) < 2e-6 and maxdiff(a.W2_v[0].cpu(), c.W2_v[0].cpu()) < 2e-6
    # without the optimiser state the resumed run is a different one (zero moments, step-1 bias correction)
    d_net = _net(dims, p, tok is None)
    d = FusedTrainer(d_net, lr=1e-3, max_norm=float(g["clip_max_norm"]))
    d_net.load_state_dict(ckpt["model"], strict=False)
    for _ in range(2):
        d.step(feats, caps, **kw)
    assert maxdiff(_flat(a), _flat(d)) > 1e-4
    # the saved dict is torch.optim.Adam's: it loads into the optimiser the reference builds
    opt = c_net.configure_optimizers()[0][0]
    opt.load_state_dict({k: ckpt["opt"][k] for k in ("state", "param_groups")})
    names = ckpt["opt"]["param_names"]
    params = opt.param_groups[0]["params"]
    assert len(params) == len(names)
    sd = dict(c_net.named_parameters())
    for i, n in enumerate(names):
        assert params[i] is sd[n], n                                 # same order as hypernet_attention.py:124-130
        assert opt.state[params[i]]["exp_avg"].shape == params[i].shape


def test_fused_plain_trainer_resume_equals_uninterrupted():
    from caphn.engine_plain import FusedPlainTrainer
    from hypernet import HyperNet
    d = O.PlainDims(E=12, H=10, V=41, L=2, cell="gru")
    p = O.init_plain_params(d, seed=7)
    rng = np.random.default_rng(3)
    imgs = torch.from_numpy(rng.standard_normal((5, d.E)).astype(np.float32)).to(DEV)
    caps = torch.from_numpy(rng.integers(0, d.V, size=(5, 7))).to(DEV)
    h0 = torch.from_numpy(rng.random((5, d.H), dtype=np.float32)).to(DEV)

    def build():
        torch.manual_seed(0)             # image_encoder.fc is not part of the oracle's parameter set: same draw in every build
        net = HyperNet(d.E, d.H, d.V, _V(), num_layers=d.L, type="gru")
        net.load_state_dict({k: v.clone() for k, v in p.items()}, strict=False)
        return net.to(DEV)
    a = FusedPlainTrainer(build(), lr=1e-3)
    for _ in range(4):
        a.step(imgs, caps, 4, h0, next_style_token=4)
    b = FusedPlainTrainer(build(), lr=1e-3)
    for _ in range(2):
        b.step(imgs, caps, 4, h0, next_style_token=4)
    model, opt = {k: v.detach().cpu().clone() for k, v in b.net.state_dict().items()}, b.state_dict()
    c_net = build()
    c = FusedPlainTrainer(c_net, lr=5.0)
    c_net.load_state_dict(model, strict=False)
    c.load_state_dict(opt)
    assert c.step_count == 2
    for _ in range(2):
        c.step(imgs, caps, 4, h0, next_style_token=4)
    assert maxdiff(a.flat_p.cpu(), c.flat_p.cpu()) < 2e-5 and maxdiff(a.W2[0].data.cpu(), c.W2[0].data.cpu()) < 2e-5


def test_fused_adam_resume_equals_uninterrupted():
    from caphn.optim import FusedAdam
    torch.manual_seed(0)
    w0 = [torch.randn(7, 5), torch.randn(11)]
    grads = [[torch.randn(7, 5), torch.randn(11)] for _ in range(4)]

    def run(steps, opt=None, ps=None):
        if ps is None:
            ps = [torch.nn.Parameter(w.clone().to(DEV)) for w in w0]
            opt = FusedAdam(ps, lr=1e-2, max_norm=1.0)
        for gs in steps:
            for q, gq in zip(ps, gs):
                q.grad = gq.to(DEV)
            opt.step()
        return opt, ps
    _, pa = run(grads)
    ob, pb = run(grads[:2])
    sd = ob.state_dict()
    pc = [torch.nn.Parameter(q.detach().clone()) for q in pb]
    oc = FusedAdam(pc, lr=9.0)
    oc.load_state_dict(sd)
    run(grads[2:], oc, pc)
    for x, y in zip(pa, pc):
        assert torch.equal(x.detach().cpu(), y.detach().cpu())
    # and the same numbers as torch.optim.Adam + clip_grad_norm_
    pt = [torch.nn.Parameter(w.clone()) for w in w0]
    ot = torch.optim.Adam(pt, lr=1e-2)
    for gs in grads:
        for q, gq in zip(pt, gs):
            q.grad = gq.clone()
        torch.nn.utils.clip_grad_norm_(pt, 1.0)
        ot.step()
    for x, y in zip(pa, pt):
        assert maxdiff(x.detach().cpu(), y.detach()) < 2e-6
