"""AttentionGru(num_layers > 1): the extra GRUCells applied as h = layer(h, h) to the initial state and after the attention
cell at every step (models/decoderlstm.py:34-36, :65-67, :101-103, greedy_search :152-163).  Golden vectors come from the
reference module itself (tools/make_golden.py layers_case, num_layers = 3)."""
import pytest
import torch
from torch.nn import functional as F

from helpers import load_case, maxdiff
from oracle import caphn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DIMS = O.Dims(D=32, F=16, E=16, H=16, V=50, he=16)


def _module(g, p=0.0):
    from models.decoderlstm import AttentionGru
    L = int(g["num_layers"])
    m = AttentionGru(DIMS.D, DIMS.F, DIMS.E, DIMS.H, DIMS.V, num_layers=L, p=p)
    res = m.load_state_dict({k[2:]: v for k, v in g.items() if k.startswith("p/")})
    assert not res.missing_keys and not res.unexpected_keys
    return m.to(DEV)


def test_teacher_forced_forward_backward_against_the_reference_module():
    g, _ = load_case("gru_layers")
    m = _module(g)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    logits, alphas = m(feats, caps.long(), 0.0)
    assert maxdiff(logits.detach().cpu(), g["logits"]) < 2e-5          # logits are O(10) here (fc scaled by 8)
    assert maxdiff(alphas.detach().cpu(), g["alphas"]) < 2e-6
    loss = F.cross_entropy(logits.view(-1, DIMS.V), caps.view(-1).long(), ignore_index=0)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    n = 0
    for name, q in m.named_parameters():
        ref = g["g/" + name]
        tol = 2e-5 * max(1.0, float(ref.abs().max()))
        assert maxdiff(q.grad.cpu(), ref) < tol, name
        n += 1
    assert n == 15 + 4 + 4 * (int(g["num_layers"]) - 1)


def test_free_running_forward():
    g, _ = load_case("gru_layers")
    m = _module(g).eval()
    with torch.no_grad():
        lf, af = m(g["features"].to(DEV), g["captions"].to(DEV).long(), 1.0)
    assert torch.equal(lf.argmax(-1).cpu(), g["logits_free"].argmax(-1))
    assert maxdiff(lf.cpu(), g["logits_free"]) < 5e-5 and maxdiff(af.cpu(), g["alphas_free"]) < 5e-6


def test_sampled_training_gradients_match_the_oracle():
    """sample_prob = 1 in training mode: backward over the sampled token ids (caphn_decoder_forward_sampled_train) with layers."""
    g, _ = load_case("gru_layers")
    m = _module(g).train()
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    logits, _ = m(feats, caps.long(), 1.0)
    loss = F.cross_entropy(logits.view(-1, DIMS.V), caps.view(-1).long(), ignore_index=0)
    loss.backward()
    p, cw, layers = _oracle_params(g)
    T = caps.shape[1]
    ref, _ = O.decoder_forward(DIMS, p, cw, g["features"], g["captions"], use_sampling=[True] * T, layers=layers)
    rl = O.caption_loss(ref, g["captions"])
    rl.backward()
    assert maxdiff(logits.detach().cpu(), ref.detach()) < 5e-5
    got = dict(m.named_parameters())
    for l, lw in enumerate(layers):
        for n, v in lw.items():
            assert maxdiff(got[f"layers.{l}.{n}"].grad.cpu(), v.grad) < 2e-5 * max(1.0, float(v.grad.abs().max())), (l, n)
    for n, v in cw.items():
        assert maxdiff(got["gru." + n].grad.cpu(), v.grad) < 2e-5 * max(1.0, float(v.grad.abs().max())), n


def _oracle_params(g):
    p = {"captioner." + k[2:]: v.clone().requires_grad_(True) for k, v in g.items()
         if k.startswith("p/") and not k.startswith("p/gru.") and not k.startswith("p/layers.")}
    cw = {n: g["p/gru." + n].clone().requires_grad_(True) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
    layers = [{n: g[f"p/layers.{l}.{n}"].clone().requires_grad_(True) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
              for l in range(int(g["num_layers"]) - 1)]
    return p, cw, layers


def test_greedy_search_against_the_reference_module():
    g, _ = load_case("gru_layers")
    m = _module(g).eval()
    with torch.no_grad():
        m.fc.bias.copy_(g["fc_bias_search"].to(DEV))
        ff = m.feature_fc(g["features"].to(DEV))
        sents, wts = m.greedy_search(ff, 2, 12)
        one, w1 = m.greedy_search(ff[2:3], 2, 12)
    lens = set()
    for b in range(ff.shape[0]):
        assert sents[b] == g[f"greedy_{b}"].tolist(), b
        assert maxdiff(torch.cat([w.cpu() for w in wts[b]], 0), g[f"greedy_w_{b}"]) < 5e-6
        lens.add(len(sents[b]))
    assert len(lens) > 1 and one == g["greedy_2"].tolist() and len(w1) == len(one)


def test_dropout_with_layers_matches_oracle_given_the_kernel_mask(monkeypatch):
    from caphn import functional as CF, ops
    seed, pd = 0x1234ABCD5678, 0.25
    monkeypatch.setattr(CF, "next_seed", lambda: seed)
    g, _ = load_case("gru_layers")
    m = _module(g, p=pd).train()
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    B, T = caps.shape
    logits, _ = m(feats, caps.long(), 0.0)
    loss = F.cross_entropy(logits.view(-1, DIMS.V), caps.view(-1).long(), ignore_index=0)
    loss.backward()
    mask = ops.dropout(torch.ones(B * T * DIMS.H, device=DEV), pd, seed).view(B, T, DIMS.H).cpu()
    assert 0.1 < float((mask == 0).float().mean()) < 0.4
    p, cw, layers = _oracle_params(g)
    ref, _ = O.decoder_forward(DIMS, p, cw, g["features"], g["captions"], drop_mask=mask, layers=layers)
    rl = O.caption_loss(ref, g["captions"])
    rl.backward()
    assert maxdiff(logits.detach().cpu(), ref.detach()) < 5e-5 and abs(float(loss) - float(rl)) < 2e-5
    got = dict(m.named_parameters())
    for k, v in p.items():
        assert maxdiff(got[k[len("captioner."):]].grad.cpu(), v.grad) < 2e-5 * max(1.0, float(v.grad.abs().max())), k
    for l, lw in enumerate(layers):
        for n, v in lw.items():
            assert maxdiff(got[f"layers.{l}.{n}"].grad.cpu(), v.grad) < 2e-5 * max(1.0, float(v.grad.abs().max())), (l, n)


def test_layers_are_rejected_for_the_lstm_cell():
    from caphn import ops
    from caphn._lib import CaphnError
    d = ops.DecDims(2, 3, 4, 8, 8, 8, 8, 20, cell="lstm", raw=True, layers=2)
    with pytest.raises(CaphnError):
        ops.decoder_workspace(d, torch.device(DEV))


@pytest.mark.parametrize("L,B,T,P,dims", [(2, 5, 9, 10, O.Dims(D=37, F=13, E=11, H=19, V=83, he=5)),
                                          (4, 3, 4, 5, O.Dims(D=24, F=12, E=10, H=12, V=40, he=4)),
                                          (2, 9, 6, 49, O.Dims(D=64, F=200, E=200, H=200, V=500, he=8))])
def test_layers_odd_and_canonical_widths_against_the_oracle(L, B, T, P, dims):
    """Nothing a multiple of 4 (scalar load paths), the maximum layer count, and the canonical F = E = H = 200 (pair kernels are
    not used with layers: one launch window per step on the one-workgroup kernels) -- module API vs the oracle (itself pinned on
    the reference's three-layer vectors)."""
    from models.decoderlstm import AttentionGru
    torch.manual_seed(L * 7 + B)
    m = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, num_layers=L, p=0.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    batch = O.synth_batch(dims, B, T, P, seed=3 + L)
    feats, caps = batch["features"], batch["captions"]
    m = m.to(DEV).train()
    logits, alphas = m(feats.to(DEV), caps.to(DEV), 0.0)
    loss = F.cross_entropy(logits.view(-1, dims.V), caps.to(DEV).view(-1), ignore_index=0)
    loss.backward()
    p = {"captioner." + k: v.clone().requires_grad_(True) for k, v in sd.items() if not k.startswith(("gru.", "layers."))}
    cw = {k[4:]: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("gru.")}
    layers = [{n: sd[f"layers.{l}.{n}"].clone().requires_grad_(True) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
              for l in range(L - 1)]
    ref, ra = O.decoder_forward(dims, p, cw, feats, caps, layers=layers)
    rl = O.caption_loss(ref, caps)
    rl.backward()
    assert maxdiff(logits.detach().cpu(), ref.detach()) < 5e-6 and maxdiff(alphas.detach().cpu(), ra.detach()) < 2e-6
    assert abs(float(loss.detach()) - float(rl.detach())) < 3e-6
    got = dict(m.named_parameters())
    for k, v in p.items():
        assert maxdiff(got[k[len("captioner."):]].grad.cpu(), v.grad) < 5e-6, k
    for k, v in cw.items():
        assert maxdiff(got["gru." + k].grad.cpu(), v.grad) < 5e-6, k
    for l, lw in enumerate(layers):
        for n, v in lw.items():
            assert maxdiff(got[f"layers.{l}.{n}"].grad.cpu(), v.grad) < 5e-6, (l, n)
