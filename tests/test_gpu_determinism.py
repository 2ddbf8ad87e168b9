"""Bit-exact re-run of the whole backward (SURVEY section 5's determinism test).  By default the split-K weight gradients and the
embedding scatter accumulate with fp32 atomics, so gradients differ in their last bits from run to run; caphn_tune(13, V) turns
split-K off and computes the embedding gradient by a destination-major scan: then two runs of the same step give bit-identical
gradients, equal (to rounding) to the default mode's."""
import pytest
import torch

from oracle import caphn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _V:
    w2i = {"<pad>": 0}

    def __call__(self, w):
        return 3


def _trainer(dims, p):
    from hypernet_attention import HyperNet
    from models.decoderlstm import AttentionGru
    from caphn.engine import FusedTrainer
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, _V(), cc=True, hyper_emb=dims.he)
    net.captioner = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
    net.load_state_dict(p, strict=False)
    return FusedTrainer(net.to(DEV), lr=1e-3, max_norm=5.0)


def test_gradients_are_bit_identical_in_deterministic_mode():
    from caphn import _lib
    lib = _lib.load()
    dims = O.Dims(D=512, F=64, E=64, H=64, V=2000, he=8)
    p = O.init_params(dims, seed=21)
    batch = O.synth_batch(dims, B=32, T=14, P=49, seed=22)
    batch["captions"][:, 3] = 17                 # one token in every caption: 32 source rows collide on one embedding row
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    x = torch.zeros(dims.he, device=DEV); x[2] = 1.0
    tr = _trainer(dims, p)
    tr.forward_backward(feats, caps, x_style=x, validate=True)
    ref = tr.flat_g.clone()
    try:
        assert lib.caphn_tune(13, dims.V) == 0
        runs = []
        for _ in range(3):
            tr.forward_backward(feats, caps, x_style=x)
            runs.append(tr.flat_g.clone())
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[1], runs[2])
        scale = float(ref.abs().max())
        assert float((runs[0] - ref).abs().max()) < 2e-6 * max(1.0, scale)
        # the collision row really carries 32 contributions
        ge = tr.grad("captioner.embed.weight")
        assert float(ge[17].abs().sum()) > 0
        # and an optimiser step from identical gradients is identical too
        tr.optimizer_step()
        p1 = tr.flat_p.clone()
        tr2 = _trainer(dims, p)
        tr2.forward_backward(feats, caps, x_style=x)
        tr2.optimizer_step()
        assert torch.equal(p1, tr2.flat_p)
    finally:
        assert lib.caphn_tune(13, 0) == 0


def test_graph_replay_equals_eager_bit_for_bit_in_deterministic_mode():
    """A race detector for the composites' side streams: with the atomics out of the way (caphn_tune 13), five steps replayed from
    a captured hipGraph and five eagerly launched steps must leave bit-identical parameters -- any ordering bug between the
    forked branches (fork modes, milestones, the pair kernels' hand-offs) would show as a difference."""
    from caphn import _lib
    lib = _lib.load()
    dims = O.Dims(D=64, F=32, E=24, H=32, V=300, he=8)
    p = O.init_params(dims, seed=31)
    batch = O.synth_batch(dims, B=16, T=9, P=12, seed=32)
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    x = torch.zeros(dims.he, device=DEV); x[1] = 1.0
    try:
        assert lib.caphn_tune(13, dims.V) == 0
        ta, tb = _trainer(dims, p), _trainer(dims, p)
        la = [float(ta.step(feats, caps, x_style=x)[0]) for _ in range(5)]
        lb = [float(tb.step_graphed(feats, caps, x_style=x)[0]) for _ in range(5)]
        assert len(tb._graphs) == 1
        assert la == lb
        assert torch.equal(ta.flat_p, tb.flat_p)
    finally:
        assert lib.caphn_tune(13, 0) == 0
