"""Pins oracle/caphn_oracle.py against golden vectors produced by the reference's own
modules (tools/make_golden.py).  CPU only.  Tolerances: fp32, same torch build, only the
summation order of hoisted/merged ops differs -> 2e-6 absolute on O(1) values."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import caphn_oracle as O
from helpers import GOLDEN, TINY_DIMS, load_case, maxdiff, style_args

GRU_CASES = ["gru_tiny_flickr", "gru_tiny_cc", "gru_odd_cc"]
ATOL = 2e-6


@pytest.mark.parametrize("name", GRU_CASES)
def test_forward_and_grads(name):
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    loss, logits, alphas, theta, grads = O.forward_backward(dims, p, x, g["features"], g["captions"],
                                                            style_token=tok)
    assert maxdiff(theta, g["theta"]) < ATOL
    assert maxdiff(logits, g["logits"]) < ATOL
    assert maxdiff(alphas, g["alphas"]) < ATOL
    assert abs(float(loss) - float(g["loss"])) < ATOL
    assert maxdiff(grads["dtheta"], g["dtheta"]) < ATOL
    n_lit = n_int = 0
    for k, v in g.items():
        if k.startswith("glit/"):
            nm = k[5:]
            if tok is not None and nm == "captioner.embed.weight":
                continue   # the oracle returns the intended (decoder + style-row) grad for embed
            assert maxdiff(grads[nm], v) < ATOL, nm
            n_lit += 1
        if k.startswith("gint/"):
            assert maxdiff(grads[k[5:]], v) < ATOL, k
            n_int += 1
    assert n_lit >= 14 and n_int >= 20


@pytest.mark.parametrize("name", GRU_CASES)
def test_literal_detached_mode(name):
    """detach_theta=True reproduces the reference's literal behaviour: no hypernet grads
    (utils.py:57; meta.json hyper_grad_none_literal)."""
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    _, _, _, _, grads = O.forward_backward(dims, p, x, g["features"], g["captions"], style_token=tok,
                                           detach_theta=True)
    assert all(grads[k] is None for k in grads if k.startswith("hn_"))
    assert maxdiff(grads["captioner.embed.weight"], g["glit/captioner.embed.weight"]) < ATOL
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    assert meta[name]["hyper_grad_none_literal"] is True
    assert meta[name]["registered"] == [n for n, _ in dims.cell_param_shapes()]
    assert meta[name]["n_set"] == dims.theta_size
    assert [tuple(h) for h in meta[name]["heads"]] == O.head_layout(dims)


@pytest.mark.parametrize("name", GRU_CASES)
def test_caption_quirks(name):
    """caps[:,0] and caps[:,T-1] never influence logits; caps[:,1] first matters at t=2
    (models/decoderlstm.py:82-88 in-place zero of a view)."""
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    xs = p["captioner.embed.weight"][tok][None] if tok is not None else x
    cellw = O.split_theta(dims, O.hyper_forward(p, xs))
    base, _ = O.decoder_forward(dims, p, cellw, g["features"], g["captions"])
    q0, _ = O.decoder_forward(dims, p, cellw, g["features"], g["captions_q0"])
    q1, _ = O.decoder_forward(dims, p, cellw, g["features"], g["captions_q1"])
    assert maxdiff(q0, base) == 0.0
    assert maxdiff(g["logits_q0"], g["logits"]) == 0.0          # the reference agrees
    assert maxdiff(q1, g["logits_q1"]) < ATOL
    assert maxdiff(q1[:, :2], base[:, :2]) == 0.0
    assert maxdiff(q1[:, 2], base[:, 2]) > 1e-4


@pytest.mark.parametrize("name", GRU_CASES)
def test_free_running(name):
    """sample_prob=1.0: greedy argmax feedback for t>=1 (models/decoderlstm.py:89-96)."""
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    xs = p["captioner.embed.weight"][tok][None] if tok is not None else x
    cellw = O.split_theta(dims, O.hyper_forward(p, xs))
    T = g["captions"].shape[1]
    logits, alphas = O.decoder_forward(dims, p, cellw, g["features"], g["captions"],
                                       use_sampling=[True] * T)
    assert maxdiff(logits, g["logits_free"]) < ATOL
    assert maxdiff(alphas, g["alphas_free"]) < ATOL
    assert torch.equal(logits.argmax(-1), g["tokens_free"])


@pytest.mark.parametrize("name", GRU_CASES)
def test_scheduled_sampling_mixed_pattern(name):
    """sample_prob = 0.5 with the reference's per-step numpy draws (models/decoderlstm.py:79-80)."""
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    xs = p["captioner.embed.weight"][tok][None] if tok is not None else x
    cellw = O.split_theta(dims, O.hyper_forward(p, xs))
    pattern = [bool(v) for v in g["mixed_pattern"]]
    assert any(pattern[1:]) and not all(pattern[1:])
    logits, alphas = O.decoder_forward(dims, p, cellw, g["features"], g["captions"], use_sampling=pattern)
    assert maxdiff(logits, g["logits_mixed"]) < ATOL and maxdiff(alphas, g["alphas_mixed"]) < ATOL


def test_lstm_free_running_and_mixed():
    """AttentionLstm's sampling rule differs from the GRU's: the sampled embedding is produced after fc and
    only consumed by a later sampling step (models/decoderlstm.py:236-251)."""
    dims = TINY_DIMS["lstm_tiny"]
    g, p = load_case("lstm_tiny")
    cellw = O.split_theta(dims, g["theta"])
    T = g["captions"].shape[1]
    lf, af = O.decoder_forward(dims, p, cellw, g["features"], g["captions"], use_sampling=[True] * T, use_feature_fc=False)
    assert maxdiff(lf, g["logits_free"]) < ATOL and maxdiff(af, g["alphas_free"]) < ATOL
    assert torch.equal(lf.argmax(-1), g["tokens_free"])
    pattern = [bool(v) for v in g["mixed_pattern"]]
    lm, am = O.decoder_forward(dims, p, cellw, g["features"], g["captions"], use_sampling=pattern, use_feature_fc=False)
    assert maxdiff(lm, g["logits_mixed"]) < ATOL and maxdiff(am, g["alphas_mixed"]) < ATOL


@pytest.mark.parametrize("name", GRU_CASES)
def test_clip_and_adam(name):
    """clip_coef + adam_step against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam."""
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    p2 = {k: v.clone() for k, v in p.items()}
    state = {}
    # feed the reference's own gradients: Adam's first step is lr * g / (|g| + eps), which turns
    # rounding noise on near-zero gradients (|g| ~ eps = 1e-8) into visible differences, so the
    # optimiser restatement is pinned on identical inputs
    gold = {k[5:]: v for k, v in g.items() if k.startswith("glit/")}
    gold.update({k[5:]: v for k, v in g.items() if k.startswith("gint/")})
    loss, tot, _, _ = O.train_step(dims, p2, state, 1, x, g["features"], g["captions"], lr=1e-3,
                                   max_norm=float(g["clip_max_norm"]), style_token=tok,
                                   grads_override=gold)
    assert abs(tot - float(g["clip_total_norm"])) < 1e-5 * max(1.0, tot)
    n = 0
    for k, v in g.items():
        if k.startswith("padam/"):
            assert maxdiff(p2[k[6:]], v) < 1e-7, k
            n += 1
    assert n == len(O.trainable_names(p))
    if name == "gru_tiny_cc":
        assert tot > float(g["clip_max_norm"])          # clipping really active in this case


def test_lstm_cell_variant():
    """Reference AttentionLstm (models/decoderlstm.py:188-261, num_features=F, p=0,
    sample_prob=0) with hypernet-injected lstm weights."""
    dims = TINY_DIMS["lstm_tiny"]
    g, p = load_case("lstm_tiny")
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    theta = O.hyper_forward(q, g["x_style"])
    assert maxdiff(theta, g["theta"]) < ATOL
    tl = theta.detach().clone().requires_grad_(True)
    logits, alphas = O.decoder_forward(dims, q, O.split_theta(dims, tl), g["features"], g["captions"],
                                       use_feature_fc=False)
    loss = O.caption_loss(logits, g["captions"])
    loss.backward()
    assert maxdiff(logits, g["logits"]) < ATOL
    assert maxdiff(alphas, g["alphas"]) < ATOL
    assert abs(float(loss) - float(g["loss"])) < ATOL
    assert maxdiff(tl.grad, g["dtheta"]) < ATOL
    for k, v in g.items():
        if k.startswith("glit/"):
            assert maxdiff(q[k[5:]].grad, v) < ATOL, k


def test_full_size_samples():
    """B=128,T=20 full-size reference run: sampled logits rows, argmax tokens, theta / dtheta
    samples, gradient norms.  Parameters regenerated from the seed."""
    path = os.path.join(GOLDEN, "gru_full.npz")
    z = np.load(path)
    dims = O.Dims()
    seed, B, T, P = int(z["seed"]), int(z["B"]), int(z["T"]), int(z["P"])
    p = O.init_params(dims, seed)
    batch = O.synth_batch(dims, B, T, P, seed=seed + 1)
    tok = int(z["style_token"])
    loss, logits, alphas, theta, grads = O.forward_backward(dims, p, None, batch["features"],
                                                            batch["captions"], style_token=tok)
    assert abs(float(loss) - float(z["loss"])) < 5e-6
    for i, (b, t) in enumerate(z["logit_rows_bt"]):
        assert np.abs(logits[b, t].numpy() - z["logit_rows"][i]).max() < 5e-6
        assert np.abs(alphas[b, t].numpy() - z["alphas_rows"][i]).max() < 1e-6
    am = logits.argmax(-1).numpy()
    safe = z["argmax_margin"] > 1e-5
    assert (am[safe] == z["argmax_tokens"][safe]).all() and safe.mean() > 0.99
    assert np.abs(theta.numpy()[z["theta_idx"]] - z["theta_vals"]).max() < 2e-6
    assert np.abs(grads["dtheta"].numpy()[z["theta_idx"]] - z["dtheta_vals"]).max() < 2e-6
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    for k, n in meta["gru_full"]["grad_norms"].items():
        mine = float(grads[k].double().norm())
        assert abs(mine - n) < 1e-4 * max(n, 1e-3), (k, mine, n)
        idx = z["gidx/" + k]
        assert np.abs(grads[k].flatten().numpy()[idx] - z["gval/" + k]).max() < 2e-6, k


def test_greedy_and_beam_search():
    """Oracle greedy_search / beam_search vs the reference's AttentionGru.greedy_search and the beam loop of
    HyperNet.test_step run around the reference's sub-modules (tools/make_golden.py search_case)."""
    dims = TINY_DIMS["gru_search"]
    g, p = load_case("gru_search")
    theta = O.hyper_forward(p, g["x_style"])
    assert maxdiff(theta, g["theta"]) < ATOL
    cellw = O.split_theta(dims, theta)
    k, end, max_sentence = int(g["beam"]), int(g["end_token"]), int(g["max_sentence"])
    feats = g["features"]
    n_unfinished = 0
    for n in range(feats.shape[0]):
        f_post = O._feature_fc(p, feats[n:n + 1])
        sent, weights = O.greedy_search(p, cellw, f_post, end, max_sentence)
        L = int(g["greedy_len"][n])
        assert sent == g["greedy_tokens"][n, :L].tolist()
        assert maxdiff(torch.cat(weights, 0), g["greedy_alphas"][n, :L]) < ATOL
        best, score, complete, cscores, margin = O.beam_search(p, cellw, feats[n:n + 1], k, end)
        assert abs(margin - float(g["beam_margin"][n])) < 1e-5
        if int(g["beam_finished"][n]):
            bl = int(g["beam_len"][n])
            assert best == g["beam_seq"][n, :bl].tolist()
            assert abs(score - float(g["beam_score"][n])) < 1e-5
        else:
            n_unfinished += 1
            assert best is None
    assert n_unfinished >= 1          # the fixture covers the reference's `step > 50` exit
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        meta = json.load(f)["gru_search"]
    assert len(set(meta["greedy_len"])) >= 3 and len(set(meta["beam_len"])) >= 3


PLAIN_CASES = ["plain_gru_l2", "plain_gru_b3", "plain_lstm_l2"]


def plain_dims(name):
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        m = json.load(f)[name]
    return O.PlainDims(**m["dims"]), m


@pytest.mark.parametrize("name", PLAIN_CASES)
def test_plain_decoder_hypernet_py(name):
    """N3: oracle restatement of hypernet.py + later.py vs the reference's own DecoderGRU / DecoderRNN classes
    (later.py source executed by tools/make_golden.py) and its flip/set_all_parameters incl. the child-offset
    restart: theta, logits (sampled columns, lse, argmax), loss, every gradient."""
    d, m = plain_dims(name)
    g, _ = load_case(name)
    p = O.init_plain_params(d, int(g["seed"]))
    assert m["heads"] == [list(x) for x in O.plain_head_layout(d)] and m["n_set"] == m["theta_size"] == d.theta_size()
    c0 = torch.zeros_like(g["h0"]) if d.cell == "lstm" else None
    loss, logits, theta, grads, dtheta = O.plain_forward_backward(d, p, g["features"], g["captions"], g["h0"], c0,
                                                                  int(g["style_token"]))
    assert maxdiff(theta, g["theta"]) < ATOL
    cols = g["cols"].long()
    assert maxdiff(logits[:, :, cols], g["logits_cols"]) < ATOL
    assert maxdiff(torch.logsumexp(logits, -1), g["logits_lse"]) < 2e-5
    assert torch.equal(logits.argmax(-1), g["tokens"])
    assert abs(float(loss) - float(g["loss"])) < ATOL
    assert maxdiff(dtheta, g["dtheta"]) < ATOL
    n0 = sum(int(np.prod(sh)) for _, sh in d.cell_param_shapes()[:4])
    assert float(dtheta[n0:].abs().sum()) == 0.0          # the extra layers' own slices are never used (utils.py:68)
    assert maxdiff(grads["features"], g["dfeatures"]) < ATOL
    rows = g["rows"].long()
    for key, nm in (("gembed", "captioner.embed.weight"), ("gfc_w", "captioner.fc_out.weight"), ("gfc_b", "captioner.fc_out.bias")):
        assert maxdiff(grads[nm][rows], g[key + "_rows"]) < ATOL, key
        assert abs(float(grads[nm].double().norm()) - float(g[key + "_norm"])) < 1e-5, key
    for k, v in g.items():
        if k.startswith("gint/"):
            assert maxdiff(grads[k[5:]], v) < ATOL, k
        elif k.startswith("gint_rows/"):
            assert maxdiff(grads[k[10:]][:32], v) < ATOL, k
            assert abs(float(grads[k[10:]].double().norm()) - float(g["gint_norm/" + k[10:]])) < 1e-5, k
    # literal leaf gradients of the overlapping views, layer by layer
    theta_l = theta.clone().requires_grad_(True)
    cells = O.plain_inject(d, theta_l)
    leaves = [{n: t.detach().clone().requires_grad_(True) for n, t in cw.items()} for cw in cells]
    lg = O.plain_decoder_forward(d, p, leaves, g["features"], g["captions"], g["h0"], c0)
    torch.nn.functional.cross_entropy(lg.reshape(-1, d.V), g["captions"].reshape(-1)).backward()
    for li, cw in enumerate(leaves):
        for n, t in cw.items():
            assert maxdiff(t.grad, g["glit/" + ("lstm_cell." if li == 0 else f"layers.{li - 1}.") + n]) < ATOL


def test_plain_train_step_protocol():
    """hypernet.py's optimiser step in the oracle: the loss falls, fc_out (no optimiser entry) and the heads whose theta
    slices nothing reads (child-offset restart) stay bit-identical, everything else in the optimiser's list moves."""
    d = O.PlainDims(E=12, H=10, V=30, L=2, cell="gru")
    p = O.init_plain_params(d, seed=1)
    p0 = {k: v.clone() for k, v in p.items()}
    g = torch.Generator().manual_seed(0)
    imgs, caps, h0 = torch.randn(3, 12, generator=g), torch.randint(0, 30, (3, 5), generator=g), torch.rand(3, 10, generator=g)
    st = {}
    losses = [float(O.plain_train_step(d, p, st, s, imgs, caps, h0, lr=1e-2)) for s in (1, 2, 3)]
    assert losses[2] < losses[0]
    assert torch.equal(p["captioner.fc_out.weight"], p0["captioner.fc_out.weight"])
    for i in range(4, 8):
        assert torch.equal(p[f"hn_heads.{i}.2.weight"], p0[f"hn_heads.{i}.2.weight"])
        assert torch.equal(p[f"hn_heads.{i}.0.bias"], p0[f"hn_heads.{i}.0.bias"])
    for n in ("hn_heads.0.2.weight", "hn_heads.3.2.bias", "hn_base.0.weight", "captioner.embed.weight"):
        assert not torch.equal(p[n], p0[n])


def test_layers_case_oracle_matches_reference_module():
    """AttentionGru(num_layers = 3): the oracle's `layers` path against the reference module's own outputs and gradients."""
    g, _ = load_case("gru_layers")
    dims = O.Dims(D=32, F=16, E=16, H=16, V=50, he=16)
    names = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
    p = {"captioner." + k[2:]: v.clone().requires_grad_(True) for k, v in g.items()
         if k.startswith("p/") and not k.startswith("p/gru.") and not k.startswith("p/layers.")}
    cw = {n: g["p/gru." + n].clone().requires_grad_(True) for n in names}
    layers = [{n: g[f"p/layers.{l}.{n}"].clone().requires_grad_(True) for n in names} for l in range(int(g["num_layers"]) - 1)]
    assert len(layers) == 2
    logits, alphas = O.decoder_forward(dims, p, cw, g["features"], g["captions"], layers=layers)
    assert maxdiff(logits, g["logits"]) < 2e-5 and maxdiff(alphas, g["alphas"]) < 2e-6
    loss = O.caption_loss(logits, g["captions"])
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for k, v in p.items():
        ref = g["g/" + k[len("captioner."):]]
        assert maxdiff(v.grad, ref) < 2e-5 * max(1.0, float(ref.abs().max())), k
    for l, lw in enumerate(layers):
        for n, v in lw.items():
            ref = g[f"g/layers.{l}.{n}"]
            assert maxdiff(v.grad, ref) < 2e-5 * max(1.0, float(ref.abs().max())), (l, n)
    T = g["captions"].shape[1]
    with torch.no_grad():
        lf, af = O.decoder_forward(dims, p, cw, g["features"], g["captions"], use_sampling=[True] * T, layers=layers)
    assert maxdiff(lf, g["logits_free"]) < 5e-5 and maxdiff(af, g["alphas_free"]) < 5e-6
    q = {k: v.detach().clone() for k, v in p.items()}
    q["captioner.fc.bias"] = g["fc_bias_search"]
    ff = O._feature_fc(q, g["features"])
    for b in range(g["features"].shape[0]):
        sent, wts = O.greedy_search(q, {k: v.detach() for k, v in cw.items()}, ff[b:b + 1], 2, 12,
                                    layers=[{k: v.detach() for k, v in lw.items()} for lw in layers])
        assert sent == g[f"greedy_{b}"].tolist()
        assert maxdiff(torch.cat(wts, 0), g[f"greedy_w_{b}"]) < 5e-6
