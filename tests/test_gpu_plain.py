"""GPU parity of the N3 row: hypernet.py's wider hypernet + later.py's non-attention DecoderGRU / DecoderRNN
(caphn_plain_forward / _backward, caphn_hyper_* with d_in / d_mid) against vectors produced by the reference's
own later.py classes and utils.py injection (child-offset restart included), through the C ABI and through the
drop-in modules.  fp32 tolerance 2e-6 absolute on O(1) values (5e-6 where split-K atomics reorder long sums)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import caphn_oracle as O
from helpers import GOLDEN, load_case, maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda"
ATOL = 2e-6
CASES = ["plain_gru_l2", "plain_gru_b3", "plain_lstm_l2"]


def _dims(name):
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        m = json.load(f)[name]
    return O.PlainDims(**m["dims"]), m


def _check_common(g, d, logits, loss):
    cols = g["cols"].long()
    assert maxdiff(logits.cpu()[:, :, cols], g["logits_cols"]) < ATOL
    assert torch.equal(logits.argmax(-1).cpu(), g["tokens"])
    assert abs(float(loss) - float(g["loss"])) < ATOL


@pytest.mark.parametrize("name", CASES)
def test_c_abi_forward_backward(name):
    from caphn import ops
    d, m = _dims(name)
    g, _ = load_case(name)
    p = {k: v.to(DEV) for k, v in O.init_plain_params(d, int(g["seed"])).items()}
    shape = ops.HyperShape(8 * d.E, [tuple(x) for x in O.plain_head_layout(d)], d_in=d.E, d_mid=4 * d.E)
    hp = {k: v for k, v in p.items() if k.startswith("hn_")}
    x = p["captioner.embed.weight"][int(g["style_token"])].clone()
    theta, acts = ops.hyper_forward(shape, hp, x)
    assert maxdiff(theta.cpu(), g["theta"]) < ATOL
    # utils.py:62-68 -- every layer reads theta from offset 0
    cells = O.plain_inject(d, theta)
    params = {"embed.weight": p["captioner.embed.weight"], "fc_out.weight": p["captioner.fc_out.weight"],
              "fc_out.bias": p["captioner.fc_out.bias"]}
    for li, cw in enumerate(cells):
        for n, t in cw.items():
            params[("lstm_cell." if li == 0 else f"layers.{li - 1}.") + n] = t.contiguous()
    B, T = g["captions"].shape
    pd = ops.PlainDims(B, T, d.E, d.H, d.V, d.L, d.cell)
    feats, caps, h0 = g["features"].to(DEV), g["captions"].to(DEV), g["h0"].to(DEV)
    c0 = torch.zeros_like(h0) if d.cell == "lstm" else None
    ws = ops.plain_workspace(pd, DEV)
    logits = ops.plain_forward(pd, params, feats, caps, h0, c0, ws)
    out, dlogits = ops.cross_entropy_fwd_bwd(logits.view(B * T, d.V), caps.view(-1), ignore_index=-100)
    _check_common(g, d, logits, out[0])
    grads = {n: torch.empty(s, device=DEV) for n, s in pd.param_shapes().items()}
    dfeat = torch.empty_like(feats)
    ops.plain_backward(pd, params, feats, caps, h0, c0, dlogits.view(B, T, d.V), grads, ws, dfeat)
    assert maxdiff(dfeat.cpu(), g["dfeatures"]) < ATOL
    rows = g["rows"].long()
    for key, nm in (("gfc_w", "fc_out.weight"), ("gfc_b", "fc_out.bias")):
        assert maxdiff(grads[nm].cpu()[rows], g[key + "_rows"]) < ATOL, key
        assert abs(float(grads[nm].double().norm()) - float(g[key + "_norm"])) < 1e-5, key
    dtheta = torch.zeros_like(theta)
    for li in range(d.L):
        off = 0
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            full = ("lstm_cell." if li == 0 else f"layers.{li - 1}.") + n
            assert maxdiff(grads[full].cpu(), g["glit/" + full]) < ATOL, full
            dtheta[off:off + grads[full].numel()] += grads[full].flatten()
            off += grads[full].numel()
    assert maxdiff(dtheta.cpu(), g["dtheta"]) < ATOL
    hg = {n: torch.empty(s, device=DEV) for n, s in shape.param_shapes().items()}
    gx = ops.hyper_backward(shape, hp, dtheta, acts, hg, want_x=True)
    for k, v in g.items():
        if k.startswith("gint/hn_"):
            assert maxdiff(hg[k[5:]].cpu(), v) < ATOL, k
        elif k.startswith("gint_rows/"):
            assert maxdiff(hg[k[10:]].cpu()[:32], v) < ATOL, k
            assert abs(float(hg[k[10:]].double().norm()) - float(g["gint_norm/" + k[10:]])) < 1e-5, k
    # the style row of embed receives gx on top of the decoder's gradient (hypernet.py:127-131)
    tok = int(g["style_token"])
    ge = grads["embed.weight"].clone(); ge[tok] += gx
    assert maxdiff(ge.cpu()[rows], g["gembed_rows"]) < ATOL
    assert abs(float(ge.double().norm()) - float(g["gembed_norm"])) < 1e-5


class _Vocab:
    w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "factual": 4}

    def __call__(self, w):
        return self.w2i.get(w, 3)


@pytest.mark.parametrize("name", CASES)
def test_module_api_hypernet_py(name):
    """hypernet.HyperNet(embed, hidden, vocab_size, vocab, num_layers, type) -> training_step protocol."""
    from hypernet import HyperNet
    d, m = _dims(name)
    g, _ = load_case(name)
    p = O.init_plain_params(d, int(g["seed"]))
    net = HyperNet(d.E, d.H, d.V, _Vocab(), num_layers=d.L, type=d.cell if d.cell == "gru" else "lstm")
    res = net.load_state_dict(p, strict=False)
    assert not res.unexpected_keys
    assert all(k.startswith(("captioner.lstm_cell.", "captioner.layers.", "image_encoder.")) for k in res.missing_keys), res
    assert [(h[0].out_features, h[2].out_features) for h in net.hn_heads] == [tuple(x) for x in m["heads"]]
    net = net.to(DEV)
    feats = g["features"].to(DEV).requires_grad_(True)
    caps = g["captions"].to(DEV)
    style = torch.tensor([int(g["style_token"])], device=DEV)
    cap = net(net.captioner.embed(style))
    assert cap.lstm_cell.registered_parameters_name == ["weight_ih", "weight_hh", "bias_ih", "bias_hh"]
    torch.manual_seed(int(g["seed"]))                  # DecoderGRU draws torch.rand for h0 (later.py:397)
    logits = cap(feats, caps, True)
    loss = F.cross_entropy(logits.view(-1, d.V), caps.view(-1))
    _check_common(g, d, logits.detach(), loss.detach())
    loss.backward()
    assert maxdiff(feats.grad.cpu(), g["dfeatures"]) < ATOL
    rows = g["rows"].long()
    assert maxdiff(net.captioner.embed.weight.grad.cpu()[rows], g["gembed_rows"]) < ATOL
    assert maxdiff(net.captioner.fc_out.weight.grad.cpu()[rows], g["gfc_w_rows"]) < ATOL
    assert maxdiff(net.captioner.fc_out.bias.grad.cpu()[rows], g["gfc_b_rows"]) < ATOL
    sd = dict(net.named_parameters())
    for k, v in g.items():
        if k.startswith("gint/hn_"):
            assert maxdiff(sd[k[5:]].grad.cpu(), v) < ATOL, k
        elif k.startswith("gint_rows/"):
            assert maxdiff(sd[k[10:]].grad.cpu()[:32], v) < ATOL, k
    # teacher_forcing=False (later.py:418-431): a word drawn per step; the ids fed are valid and the call is differentiable
    cap = net(net.captioner.embed(style))              # fresh theta: the first backward freed the hypernet's graph
    out = cap(feats, caps, False)
    ids = cap.last_sampled
    assert out.shape == logits.shape and ids.shape == caps.shape
    assert bool((ids[:, 0] == -1).all()) and bool((ids[:, 1:] >= 0).all()) and bool((ids[:, 1:] < d.V).all())
    F.cross_entropy(out.view(-1, d.V), caps.view(-1)).backward()


@pytest.mark.parametrize("cell,L", [("gru", 1), ("gru", 3), ("lstm", 2)])
def test_medium_size_vs_oracle(cell, L):
    """B=32, T=12, E=64, H=48 (odd multiples, H not a multiple of 32), V=777: C ABI vs the oracle."""
    from caphn import ops
    d = O.PlainDims(E=64, H=48, V=777, L=L, cell=cell)
    p = O.init_plain_params(d, 5)
    rng = np.random.default_rng(11)
    B, T = 32, 12
    feats = torch.from_numpy(rng.standard_normal((B, d.E), dtype=np.float32))
    caps = torch.from_numpy(rng.integers(0, d.V, size=(B, T)))
    h0 = torch.from_numpy(rng.random((B, d.H), dtype=np.float32))
    c0 = torch.zeros(B, d.H) if cell == "lstm" else None
    loss, logits, theta, grads, dtheta = O.plain_forward_backward(d, p, feats, caps, h0, c0, 4)
    cells = O.plain_inject(d, theta.to(DEV))
    params = {"embed.weight": p["captioner.embed.weight"].to(DEV), "fc_out.weight": p["captioner.fc_out.weight"].to(DEV),
              "fc_out.bias": p["captioner.fc_out.bias"].to(DEV)}
    for li, cw in enumerate(cells):
        for n, t in cw.items():
            params[("lstm_cell." if li == 0 else f"layers.{li - 1}.") + n] = t.contiguous()
    pd = ops.PlainDims(B, T, d.E, d.H, d.V, L, cell)
    ws = ops.plain_workspace(pd, DEV)
    f, c, h = feats.to(DEV), caps.to(DEV), h0.to(DEV)
    cz = c0.to(DEV) if c0 is not None else None
    lg = ops.plain_forward(pd, params, f, c, h, cz, ws)
    assert maxdiff(lg.cpu(), logits) < 5e-6
    l2, dl = ops.cross_entropy_fwd_bwd(lg.view(B * T, d.V), c.view(-1), ignore_index=-100)
    assert abs(float(l2[0]) - float(loss)) < 5e-6
    gr = {n: torch.empty(s, device=DEV) for n, s in pd.param_shapes().items()}
    df = torch.empty_like(f)
    ops.plain_backward(pd, params, f, c, h, cz, dl.view(B, T, d.V), gr, ws, df)
    assert maxdiff(df.cpu(), grads["features"]) < 5e-6
    assert maxdiff(gr["fc_out.weight"].cpu(), grads["captioner.fc_out.weight"]) < 5e-6
    dth = torch.zeros(d.theta_size(), device=DEV)
    for li in range(L):
        off = 0
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            t = gr[("lstm_cell." if li == 0 else f"layers.{li - 1}.") + n]
            dth[off:off + t.numel()] += t.flatten(); off += t.numel()
    assert maxdiff(dth.cpu(), dtheta) < 5e-6


def test_hypernet_py_literal_configuration():
    """hypernet.py:203 -- HyperNet(200, 150, len(vocab), vocab, 2, type='gru') at B=128, T=20: 2.785 G hypernet
    parameters (11.1 GB), heads (11250, 90000), 3 x (8437, 67500), 4 x (450, 450) -- row widths that are neither
    multiples of four nor below the register-tiled kernels' 2048-column limit.  Size-independent properties against
    plain PyTorch on the same device: theta, the rank-1 structure of the big gradient, dead heads."""
    from hypernet import HyperNet
    torch.manual_seed(0)
    E, H, V, L, B, T = 200, 150, 9684, 2, 128, 20
    with torch.device(DEV):
        net = HyperNet(E, H, V, _Vocab(), num_layers=L, type="gru")
    assert [(h[0].out_features, h[2].out_features) for h in net.hn_heads] == \
        [(11250, 90000), (8437, 67500), (450, 450), (450, 450), (8437, 67500), (8437, 67500), (450, 450), (450, 450)]
    assert sum(p.numel() for p in net.hn_heads.parameters()) + sum(p.numel() for p in net.hn_base.parameters()) == 2784955161
    feats = torch.randn(B, E, device=DEV, requires_grad=True)
    caps = torch.randint(0, V, (B, T), device=DEV)
    style = torch.tensor([4], device=DEV)
    cap = net(net.captioner.embed(style))
    logits = cap(feats, caps, True)
    loss = F.cross_entropy(logits.view(-1, V), caps.view(-1))
    loss.backward()
    with torch.no_grad():
        x = net.captioner.embed.weight[4:5]
        base = F.leaky_relu(F.linear(F.leaky_relu(F.linear(x, net.hn_base[0].weight, net.hn_base[0].bias)),
                                     net.hn_base[2].weight, net.hn_base[2].bias))
        acts = [F.leaky_relu(F.linear(base, h[0].weight, h[0].bias)) for h in net.hn_heads]
        theta_ref = torch.cat([F.linear(a, h[2].weight, h[2].bias).flatten() for a, h in zip(acts, net.hn_heads)])
    names = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
    got = torch.cat([getattr(net.captioner.lstm_cell, n).detach().flatten() for n in names])
    assert float((got - theta_ref[:got.numel()]).abs().max()) < 2e-6
    # utils.py:68: the extra layer re-reads theta from offset 0
    lay = torch.cat([getattr(net.captioner.layers[0], n).detach().flatten() for n in names])
    assert torch.equal(lay, theta_ref[:lay.numel()].reshape(-1)) or float((lay - theta_ref[:lay.numel()]).abs().max()) < 2e-6
    g, db = net.hn_heads[0][2].weight.grad, net.hn_heads[0][2].bias.grad
    rows = torch.randint(0, g.shape[0], (64,), device=DEV)
    assert float((g[rows] - db[rows, None] * acts[0][0][None, :]).abs().max()) < 1e-9
    assert float(g.abs().max()) > 0
    assert all(float(net.hn_heads[i][2].weight.grad.abs().sum()) == 0.0 for i in range(4, 8))
    assert bool(torch.isfinite(loss))


@pytest.mark.parametrize("cell,L", [("gru", 2), ("lstm", 1)])
def test_sampled_branch_equals_teacher_forcing_over_the_drawn_ids(cell, L):
    """teacher_forcing=False (later.py:418-431 / :290-301): x_t = embed(w), w ~ softmax(out_{t-1}).  The kernel's draws come from
    its own counter-based stream (torch.multinomial's Philox stream cannot be reproduced: parity with the reference's DRAWS is
    unpinned); given the ids it drew, logits and every gradient must equal the oracle's teacher-forced pass over those ids
    with the loss still taken against the caption -- which is the reference's autograd graph (nothing flows through the draw)."""
    from caphn import ops
    d = O.PlainDims(E=40, H=36, V=333, L=L, cell=cell)
    p = O.init_plain_params(d, 8)
    rng = np.random.default_rng(12)
    B, T = 16, 9
    feats = torch.from_numpy(rng.standard_normal((B, d.E), dtype=np.float32))
    caps = torch.from_numpy(rng.integers(0, d.V, size=(B, T)))
    h0 = torch.from_numpy(rng.random((B, d.H), dtype=np.float32))
    c0 = torch.zeros(B, d.H) if cell == "lstm" else None
    _, _, theta, _, _ = O.plain_forward_backward(d, p, feats, caps, h0, c0, 4)
    cells = [{n: (3.0 * t).contiguous() for n, t in cw.items()} for cw in O.plain_inject(d, theta.to(DEV))]     # sharper softmax
    params = {"embed.weight": p["captioner.embed.weight"].to(DEV), "fc_out.weight": (4.0 * p["captioner.fc_out.weight"]).to(DEV),
              "fc_out.bias": p["captioner.fc_out.bias"].to(DEV)}
    for li, cw in enumerate(cells):
        for n, t in cw.items():
            params[("lstm_cell." if li == 0 else f"layers.{li - 1}.") + n] = t
    pd = ops.PlainDims(B, T, d.E, d.H, d.V, L, cell)
    ws = ops.plain_workspace(pd, DEV)
    f, c, h = feats.to(DEV), caps.to(DEV), h0.to(DEV)
    cz = c0.to(DEV) if c0 is not None else None
    lg, ids = ops.plain_forward_sampled(pd, params, f, h, cz, ws, seed=0xABCDEF)
    lg2, ids2 = ops.plain_forward_sampled(pd, params, f, h, cz, ops.plain_workspace(pd, DEV), seed=0xABCDEF)
    assert torch.equal(ids, ids2) and torch.equal(lg, lg2)                      # a seed fixes the draws
    _, ids3 = ops.plain_forward_sampled(pd, params, f, h, cz, ops.plain_workspace(pd, DEV), seed=0xABCDF0)
    assert not torch.equal(ids, ids3)
    assert bool((ids[:, 0] == -1).all()) and bool((ids[:, 1:] >= 0).all()) and bool((ids[:, 1:] < d.V).all())
    assert len(torch.unique(ids[:, 1:])) > 20                                   # not an argmax in disguise
    # oracle: teacher forcing over the drawn ids (the id fed at step t sits in column t - 1 of the caption it replaces)
    eff = caps.clone()
    eff[:, :T - 1] = ids[:, 1:].cpu()
    q = {k: v.clone() for k, v in p.items()}
    q["captioner.fc_out.weight"] = 4.0 * q["captioner.fc_out.weight"]
    q = {k: v.requires_grad_(True) for k, v in q.items()}
    ocells = [{n: t.detach().cpu().clone().requires_grad_(True) for n, t in cw.items()} for cw in cells]
    fr = feats.clone().requires_grad_(True)
    ref = O.plain_decoder_forward(d, q, ocells, fr, eff, h0, c0)
    assert maxdiff(lg.cpu(), ref.detach()) < 2e-5
    rl = torch.nn.functional.cross_entropy(ref.view(-1, d.V), caps.view(-1))
    rl.backward()
    l2, dl = ops.cross_entropy_fwd_bwd(lg.view(B * T, d.V), c.view(-1), ignore_index=-100)
    assert abs(float(l2[0]) - float(rl)) < 1e-5
    gr = {n: torch.empty(s, device=DEV) for n, s in pd.param_shapes().items()}
    df = torch.empty_like(f)
    ops.plain_backward(pd, params, f, c, h, cz, dl.view(B, T, d.V), gr, ws, df)
    assert maxdiff(df.cpu(), fr.grad) < 1e-5
    assert maxdiff(gr["fc_out.weight"].cpu(), q["captioner.fc_out.weight"].grad) < 1e-5
    assert maxdiff(gr["embed.weight"].cpu(), q["captioner.embed.weight"].grad) < 1e-5
    for li, cw in enumerate(ocells):
        for n, t in cw.items():
            assert maxdiff(gr[("lstm_cell." if li == 0 else f"layers.{li - 1}.") + n].cpu(), t.grad) < 1e-5, (li, n)


def test_sampled_words_follow_the_softmax():
    """The draw itself: 4096 captions share one logits row; the histogram of the words drawn at step 1 matches softmax(out_0)."""
    from caphn import ops
    d = O.PlainDims(E=16, H=16, V=50, L=1, cell="gru")
    p = O.init_plain_params(d, 3)
    B, T = 4096, 2
    rng = np.random.default_rng(5)
    feats = torch.from_numpy(rng.standard_normal((1, d.E), dtype=np.float32)).repeat(B, 1).to(DEV)
    h0 = torch.from_numpy(rng.random((1, d.H), dtype=np.float32)).repeat(B, 1).to(DEV)
    params = {"embed.weight": p["captioner.embed.weight"].to(DEV), "fc_out.weight": (6.0 * p["captioner.fc_out.weight"]).to(DEV),
              "fc_out.bias": p["captioner.fc_out.bias"].to(DEV)}
    g = torch.Generator().manual_seed(1)
    for n, shape in (("weight_ih", (3 * d.H, d.E)), ("weight_hh", (3 * d.H, d.H)), ("bias_ih", (3 * d.H,)), ("bias_hh", (3 * d.H,))):
        params["lstm_cell." + n] = ((torch.rand(shape, generator=g) - 0.5) * 1.5).to(DEV)
    pd = ops.PlainDims(B, T, d.E, d.H, d.V, 1, "gru")
    lg, ids = ops.plain_forward_sampled(pd, params, feats, h0, None, ops.plain_workspace(pd, DEV), seed=77)
    prob = torch.softmax(lg[0, 0].double().cpu(), 0)
    assert float((lg[:, 0] - lg[0, 0]).abs().max()) == 0.0
    hist = torch.bincount(ids[:, 1].cpu(), minlength=d.V).double() / B
    assert float(prob.max()) < 0.6 and int((prob > 0.02).sum()) >= 5            # a real distribution, several likely words
    # binomial standard error of a frequency is sqrt(p (1 - p) / B) <= 0.0078: allow 4.5 sigma
    assert float((hist - prob).abs().max()) < 4.5 * 0.0078, (hist, prob)
