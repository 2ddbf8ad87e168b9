"""Fused training step of hypernet.py (caphn.engine_plain.FusedPlainTrainer) against the oracle's plain_train_step:
losses and every parameter after three Adam steps, for GRU / LSTM, one and several layers, E > H and H > E (then an
extra layer reads further into theta than the first cell does, so heads past the first four are live), pooled
2048-d features through image_encoder.fc and ready embeddings, with and without the next step's theta produced in
the optimiser pass.  Adam divides by sqrt(v): a gradient that differs in its last bit moves a parameter by up to
lr * 1e-3 in the first steps, hence the tolerance relative to lr."""
import numpy as np
import pytest
import torch

from oracle import caphn_oracle as O
from helpers import maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda"
LR = 1e-3


class _Vocab:
    w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "factual": 4}

    def __call__(self, w):
        return self.w2i.get(w, 3)


def _net(d, p):
    from hypernet import HyperNet
    net = HyperNet(d.E, d.H, d.V, _Vocab(), num_layers=d.L, type=d.cell if d.cell == "gru" else "lstm")
    res = net.load_state_dict(p, strict=False)
    assert not res.unexpected_keys
    return net.to(DEV)


@pytest.mark.parametrize("cell,L,E,H,through_fc,prefetch", [
    ("gru", 2, 12, 10, True, True), ("gru", 1, 12, 10, False, False), ("lstm", 2, 10, 12, True, True),
    ("gru", 3, 16, 20, False, True), ("lstm", 1, 12, 12, True, False)])
def test_three_steps_match_oracle(cell, L, E, H, through_fc, prefetch):
    from caphn.engine_plain import FusedPlainTrainer
    d = O.PlainDims(E=E, H=H, V=41, L=L, cell=cell)
    p = O.init_plain_params(d, seed=7)
    rng = np.random.default_rng(3)
    p["image_encoder.fc.weight"] = torch.from_numpy(rng.uniform(-0.02, 0.02, size=(E, 2048)).astype(np.float32))
    p["image_encoder.fc.bias"] = torch.from_numpy(rng.uniform(-0.02, 0.02, size=(E,)).astype(np.float32))
    net = _net(d, {k: v.clone() for k, v in p.items()})
    tr = FusedPlainTrainer(net, lr=LR)
    used = max(sum(int(np.prod(sh)) for _, sh in d.cell_param_shapes()[4 * l:4 * l + 4]) for l in range(L))
    offs = np.cumsum([0] + [w for _, w in O.plain_head_layout(d)])[:-1]
    assert tr.n_live == int((offs < used).sum()) and (tr.n_live > 4) == (L > 1 and H > E)
    B, T = 5, 7
    state = {}
    toks = [4, 4, 6]
    for s in range(3):
        imgs = torch.from_numpy(rng.standard_normal((B, 2048 if through_fc else E)).astype(np.float32))
        caps = torch.from_numpy(rng.integers(0, d.V, size=(B, T)))
        h0 = torch.from_numpy(rng.random((B, H), dtype=np.float32))
        c0 = torch.zeros(B, H) if cell == "lstm" else None
        ref = O.plain_train_step(d, p, state, s + 1, imgs, caps, h0, c0, style_token=toks[s], lr=LR)
        nxt = toks[s + 1] if (prefetch and s + 1 < len(toks)) else None
        out = tr.step(imgs.to(DEV), caps.to(DEV), toks[s], h0.to(DEV), c0.to(DEV) if c0 is not None else None,
                      next_style_token=nxt)
        assert abs(float(out[0]) - float(ref)) < 5e-6, (s, float(out[0]), float(ref))
    sd = dict(net.named_parameters())
    live = tr.n_live
    for name, want in p.items():
        got = sd[name].detach().cpu()
        tol = 2e-2 * LR if name.startswith(O.PLAIN_OPTIMISED_PREFIXES) else 0.0
        assert maxdiff(got, want) <= tol, (name, maxdiff(got, want))
    # heads whose output the captioner never reads were not touched (Adam with zero gradient and state is the identity)
    p0 = O.init_plain_params(d, seed=7)
    for i in range(live, len(net.hn_heads)):
        assert torch.equal(sd[f"hn_heads.{i}.2.weight"].detach().cpu(), p0[f"hn_heads.{i}.2.weight"])
        assert torch.equal(p[f"hn_heads.{i}.2.weight"], p0[f"hn_heads.{i}.2.weight"])       # ... in the oracle too
    assert torch.equal(sd["captioner.fc_out.weight"].detach().cpu(), p0["captioner.fc_out.weight"])   # no optimiser entry


def test_needs_cuda_model():
    from caphn.engine_plain import FusedPlainTrainer
    from caphn._lib import CaphnError
    from hypernet import HyperNet
    with pytest.raises(CaphnError):
        FusedPlainTrainer(HyperNet(8, 6, 20, _Vocab(), num_layers=1, type="gru"))


@pytest.mark.parametrize("cell,L", [("gru", 2), ("lstm", 1)])
def test_sampled_branch_in_the_fused_trainer_given_the_ids_drawn(cell, L):
    """hypernet.py:135-140: np.random.binomial picks teacher forcing or the sampled branch per step.  FusedPlainTrainer.step(
    teacher_forcing=False) runs later.py:418-431's branch on libcaphn; the reference's draws (torch.multinomial) cannot be
    reproduced, so parity is: GIVEN the ids the kernel drew, loss and every parameter after each Adam step equal the oracle's
    step whose cell inputs are those ids and whose loss is taken against the caption."""
    from caphn.engine_plain import FusedPlainTrainer
    d = O.PlainDims(E=12, H=10, V=41, L=L, cell=cell)
    p = O.init_plain_params(d, seed=11)
    net = _net(d, {k: v.clone() for k, v in p.items()})
    tr = FusedPlainTrainer(net, lr=LR)
    rng = np.random.default_rng(5)
    B, T = 6, 7
    state = {}
    for s, tf in enumerate([False, True, False]):
        imgs = torch.from_numpy(rng.standard_normal((B, d.E)).astype(np.float32))
        caps = torch.from_numpy(rng.integers(0, d.V, size=(B, T)))
        h0 = torch.from_numpy(rng.random((B, d.H), dtype=np.float32))
        c0 = torch.zeros(B, d.H) if cell == "lstm" else None
        out = tr.step(imgs.to(DEV), caps.to(DEV), 4, h0.to(DEV), c0.to(DEV) if c0 is not None else None,
                      teacher_forcing=tf, seed=1000 + s)
        eff = None
        if not tf:
            ids = tr.last_chosen.cpu()
            assert bool((ids[:, 0] == -1).all()) and bool((ids[:, 1:] >= 0).all())
            eff = caps.clone()
            eff[:, :T - 1] = ids[:, 1:]
        else:
            assert tr.last_chosen is None
        ref = O.plain_train_step(d, p, state, s + 1, imgs, caps, h0, c0, style_token=4, lr=LR, input_captions=eff)
        assert abs(float(out[0]) - float(ref)) < 2e-5, (s, float(out[0]), float(ref))
    got = dict(net.named_parameters())
    for n in p:
        if n.startswith(O.PLAIN_OPTIMISED_PREFIXES):
            assert maxdiff(got[n].detach().cpu(), p[n]) < 2e-2 * LR + 1e-6, n
