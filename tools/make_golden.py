#!/usr/bin/env python3
"""Generate golden vectors under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference; the GPU box never sees it).
It imports, from the reference tree:
    models.decoderlstm.AttentionGru / AttentionLstm   (models/decoderlstm.py:11,188)
    models.attention.BahdanauAttention                (models/attention.py:5)
    utils.flip_parameters_to_tensors / set_all_parameters   (utils.py:24,44)
after placing inert stub modules in sys.modules for packages that are merely not
installed here (nltk, rouge_metric, torchvision) -- none is touched by the hot path.
hypernet_attention.HyperNet.__init__ cannot run offline (it fetches metric scripts and
ResNet weights, hypernet_attention.py:47,50), so its 45-line constructor rule (:55-99)
and forward (:111-121) are restated here around the imported reference functions.

Parameters come from oracle.init_params (numpy PCG64 seeds) and are LOADED INTO the
reference modules, so reference and oracle/HIP paths compute with identical weights.
Only data is written: inputs, parameters (tiny cases), expected outputs.
"""
import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CAPHN_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

SEARCH_END_BUMP = 1.2936  # added to fc.bias[</s>] so random-init captions terminate (config found by random search: varied lengths, one unfinished beam)
SEARCH_SHARPEN = (6.0, 4.0, 8.0)


def _import_reference():
    from transformers import BertTokenizer  # noqa: F401  (utils.py:20 imports it)
    for name in ["nltk", "nltk.translate", "nltk.translate.bleu_score",
                 "nltk.translate.meteor_score", "rouge_metric", "torchvision",
                 "torchvision.models"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    for attr in ("corpus_bleu", "sentence_bleu", "SmoothingFunction"):
        setattr(sys.modules["nltk.translate.bleu_score"], attr, None)
    for attr in ("single_meteor_score", "meteor_score"):
        setattr(sys.modules["nltk.translate.meteor_score"], attr, None)
    sys.modules["rouge_metric"].PyRouge = None
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.path.insert(0, REF)
    from models.decoderlstm import AttentionGru, AttentionLstm
    from utils import flip_parameters_to_tensors, set_all_parameters
    sys.path.pop(0)
    return AttentionGru, AttentionLstm, flip_parameters_to_tensors, set_all_parameters


AttentionGru, AttentionLstm, flip_parameters_to_tensors, set_all_parameters = _import_reference()
sys.path.insert(0, REPO)
from oracle import caphn_oracle as O  # noqa: E402


def build_ref_hypernet(cell_module, he):
    """hypernet_attention.py:55-99 restated (N=1, M=500) over cell_module.named_parameters()."""
    N, M = 1, 500
    hn_base = nn.Sequential(nn.Linear(he, N * he), nn.LeakyReLU(),
                            nn.Linear(N * he, N * he), nn.LeakyReLU())
    heads = []
    for name, W in cell_module.named_parameters():
        w_size = len(W.flatten())
        if w_size < N * he:
            heads.append(nn.Sequential(nn.Linear(N * he, N), nn.LeakyReLU(), nn.Linear(w_size, w_size)))
        elif w_size // M < N * he:
            heads.append(nn.Sequential(nn.Linear(N * he, N * he), nn.LeakyReLU(), nn.Linear(N * he, w_size)))
        else:
            heads.append(nn.Sequential(nn.Linear(N * he, w_size // M), nn.LeakyReLU(),
                                       nn.Linear(w_size // M, w_size)))
    return hn_base, nn.ModuleList(heads)


def load_into(module, prefix, p):
    sd = {k[len(prefix):]: v.clone() for k, v in p.items() if k.startswith(prefix)}
    missing = module.load_state_dict(sd, strict=False)
    return missing


def ref_gru_step(dims, p, features, captions, x_style=None, style_token=None,
                 sample_prob=0.0, want_grads=True):
    """hypernet_attention.py:111-121 + :146,183 with reference modules."""
    cap = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
    hn_base, hn_heads = build_ref_hypernet(cap.gru, dims.he)
    order = [n for n, _ in cap.gru.named_parameters()]
    # load shared parameters (cap.gru keeps its own init; it is overwritten by theta)
    msd = {k[len("captioner."):]: v.clone() for k, v in p.items() if k.startswith("captioner.")}
    res = cap.load_state_dict(msd, strict=False)
    assert all(k.startswith("gru.") for k in res.missing_keys), res
    assert not res.unexpected_keys, res
    hn_base.load_state_dict({k[len("hn_base."):]: v.clone() for k, v in p.items() if k.startswith("hn_base.")})
    hn_heads.load_state_dict({k[len("hn_heads."):]: v.clone() for k, v in p.items() if k.startswith("hn_heads.")})
    if style_token is not None:
        x = cap.embed(torch.tensor([style_token]))            # hypernet_attention.py:139-142
    else:
        x = x_style
    base_feat = hn_base(x)
    heads_out = torch.cat([h(base_feat).flatten() for h in hn_heads], dim=0)
    flip_parameters_to_tensors(cap.gru)
    n_set = set_all_parameters(cap.gru, heads_out.reshape(1, -1))
    reg = list(cap.gru.registered_parameters_name)
    logits, alphas = cap(features, captions.long(), sample_prob)
    loss = F.cross_entropy(logits.view(-1, dims.V), captions.view(-1).long(), ignore_index=0)
    out = {"theta": heads_out.detach(), "logits": logits.detach(), "alphas": alphas.detach(),
           "loss": loss.detach(), "n_set": n_set, "order": order, "registered": reg}
    if not want_grads:
        return out, None
    loss.backward()
    lit = {"captioner." + n: q.grad.detach().clone() for n, q in cap.named_parameters()
           if not n.startswith("gru.") and q.grad is not None}
    hyper_none = all(q.grad is None for q in list(hn_base.parameters()) + list(hn_heads.parameters()))
    dtheta = torch.cat([getattr(cap.gru, n).grad.flatten() for n in reg])
    # (ii) intended grads: VJP of the hypernet with dtheta
    heads_out.backward(dtheta)
    intended = {}
    for n, q in hn_base.named_parameters():
        intended["hn_base." + n] = q.grad.detach().clone()
    for n, q in hn_heads.named_parameters():
        intended["hn_heads." + n] = q.grad.detach().clone()
    if style_token is not None:
        # embed.weight.grad now holds decoder grad + the style-row VJP
        lit_plus = cap.embed.weight.grad.detach().clone()
        intended["captioner.embed.weight"] = lit_plus
    out.update({"hyper_grad_none_literal": hyper_none, "dtheta": dtheta})
    return out, (lit, intended)


def save_npz(path, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        flat[k] = v
    np.savez_compressed(path, **flat)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def tiny_case(name, dims, B, T, P, seed, flickr, max_norm):
    p = O.init_params(dims, seed)
    torch.manual_seed(seed)
    batch = O.synth_batch(dims, B, T, P, seed=seed + 1)
    feats, caps = batch["features"], batch["captions"]
    if flickr:
        style_token, x_style = 4 + batch["domain"], None
    else:
        style_token = None
        x_style = torch.zeros(dims.he)
        x_style[batch["domain"] % dims.he] = 1.0       # one-hot row, 1-D (cc_train_hypernet.py:142-144)
    out, (lit, intended) = ref_gru_step(dims, p, feats, caps, x_style, style_token)
    arrs = {"features": feats, "captions": caps, "style_token": np.int64(-1 if style_token is None else style_token)}
    if x_style is not None:
        arrs["x_style"] = x_style
    for k, v in p.items():
        arrs["p/" + k] = v
    for k in ("theta", "logits", "alphas", "loss", "dtheta"):
        arrs[k] = out[k]
    for k, v in lit.items():
        arrs["glit/" + k] = v
    for k, v in intended.items():
        arrs["gint/" + k] = v
    # quirk vectors: caps[:,0] / caps[:,T-1] never influence logits,
    # caps[:,1] first matters at t = 2
    caps_a = caps.clone(); caps_a[:, 0] = (caps_a[:, 0] + 3) % dims.V; caps_a[:, T - 1] = (caps_a[:, T - 1] + 5) % dims.V
    oa, _ = ref_gru_step(dims, p, feats, caps_a, x_style, style_token, want_grads=False)
    caps_b = caps.clone(); caps_b[:, 1] = (caps_b[:, 1] + 7) % dims.V
    ob, _ = ref_gru_step(dims, p, feats, caps_b, x_style, style_token, want_grads=False)
    arrs["captions_q0"] = caps_a; arrs["logits_q0"] = oa["logits"]
    arrs["captions_q1"] = caps_b; arrs["logits_q1"] = ob["logits"]
    # free-running (sample_prob = 1.0 -> np.random.random() < 1.0 always true for t>=1)
    of, _ = ref_gru_step(dims, p, feats, caps, x_style, style_token, sample_prob=1.0, want_grads=False)
    arrs["logits_free"] = of["logits"]; arrs["alphas_free"] = of["alphas"]
    arrs["tokens_free"] = of["logits"].argmax(-1)
    # scheduled sampling with p = 0.5: the draws come from numpy's global RNG, one per timestep (:80)
    np.random.seed(4321)
    draws = np.random.random(T)
    arrs["mixed_pattern"] = np.array([bool(t > 0 and draws[t] < 0.5) for t in range(T)])
    np.random.seed(4321)
    om, _ = ref_gru_step(dims, p, feats, caps, x_style, style_token, sample_prob=0.5, want_grads=False)
    arrs["logits_mixed"] = om["logits"]; arrs["alphas_mixed"] = om["alphas"]
    # one clip + Adam step with torch's own implementations on the intended grads
    names = O.trainable_names(p)
    g_all = dict(lit); g_all.update(intended)
    plist = []
    for n in names:
        q = nn.Parameter(p[n].clone()); q.grad = g_all[n].clone(); plist.append(q)
    opt = torch.optim.Adam(plist, lr=1e-3)
    tot = torch.nn.utils.clip_grad_norm_(plist, max_norm)
    opt.step()
    arrs["clip_total_norm"] = tot.detach()
    arrs["clip_max_norm"] = np.float64(max_norm)
    for n, q in zip(names, plist):
        arrs["padam/" + n] = q.detach()
    save_npz(os.path.join(OUT, name + ".npz"), **arrs)
    return {"order": out["order"], "registered": out["registered"], "n_set": int(out["n_set"]),
            "hyper_grad_none_literal": bool(out["hyper_grad_none_literal"]),
            "heads": O.head_layout(dims)}


def lstm_case(name, seed):
    """Reference AttentionLstm (models/decoderlstm.py:188-261) at num_features = F, p = 0,
    sample_prob = 0: pins the oracle's LSTM loop (use_feature_fc=False).  Hypernet
    injection through flip/set_all_parameters on .lstm."""
    dims = O.Dims(D=12, F=12, E=8, H=8, V=40, he=8, cell="lstm")
    B, T, P = 3, 6, 5
    p = O.init_params(dims, seed)
    batch = O.synth_batch(dims, B, T, P, seed=seed + 1)
    feats, caps = batch["features"], batch["captions"]
    m = AttentionLstm(dims.F, dims.E, dims.H, dims.V, p=0.0)
    sd = {"embeddings.weight": p["captioner.embed.weight"]}
    for k in ("fc.weight", "fc.bias", "attention.W_a.weight", "attention.W_a.bias",
              "attention.U_a.weight", "attention.U_a.bias", "attention.v_a.weight",
              "attention.v_a.bias", "init_h.weight", "init_h.bias", "init_c.weight", "init_c.bias"):
        sd[k] = p["captioner." + k]
    res = m.load_state_dict(sd, strict=False)
    assert all(k.startswith("lstm.") for k in res.missing_keys), res
    hn_base, hn_heads = build_ref_hypernet(m.lstm, dims.he)
    hn_base.load_state_dict({k[len("hn_base."):]: v.clone() for k, v in p.items() if k.startswith("hn_base.")})
    hn_heads.load_state_dict({k[len("hn_heads."):]: v.clone() for k, v in p.items() if k.startswith("hn_heads.")})
    x = torch.zeros(dims.he); x[batch["domain"]] = 1.0
    theta = torch.cat([h(hn_base(x)).flatten() for h in hn_heads])
    flip_parameters_to_tensors(m.lstm)
    set_all_parameters(m.lstm, theta.reshape(1, -1))
    reg = list(m.lstm.registered_parameters_name)
    logits, alphas = m(caps, feats, 0.0)
    loss = F.cross_entropy(logits.view(-1, dims.V), caps.view(-1), ignore_index=0)
    loss.backward()
    dtheta = torch.cat([getattr(m.lstm, n).grad.flatten() for n in reg])
    arrs = {"features": feats, "captions": caps, "x_style": x, "theta": theta, "logits": logits,
            "alphas": alphas, "loss": loss, "dtheta": dtheta}
    for k, v in p.items():
        arrs["p/" + k] = v
    name_map = {"embeddings.weight": "captioner.embed.weight"}
    for n, q in m.named_parameters():
        if n.startswith("lstm."):
            continue
        arrs["glit/" + name_map.get(n, "captioner." + n)] = q.grad
    # free running: the reference's default sample_prob = 1.0 (:224), every step t >= 1 samples
    with torch.no_grad():
        lf, af = m(caps, feats, 1.0)
    arrs["logits_free"] = lf; arrs["alphas_free"] = af; arrs["tokens_free"] = lf.argmax(-1)
    # a mixed pattern: force the per-step draws through numpy's global RNG exactly as the reference consumes it
    import numpy as _np
    _np.random.seed(1234)
    draws = _np.random.random(T)
    pattern = [bool(t > 0 and draws[t] < 0.5) for t in range(T)]
    _np.random.seed(1234)
    with torch.no_grad():
        lm, am = m(caps, feats, 0.5)
    arrs["mixed_pattern"] = _np.array(pattern); arrs["logits_mixed"] = lm; arrs["alphas_mixed"] = am
    save_npz(os.path.join(OUT, name + ".npz"), **arrs)
    return {"registered": reg, "heads": O.head_layout(dims)}


def full_case(name, seed):
    """Full-size GRU (B=128,T=20,P=49,D=2048,F=E=H=200,V=9684,he=200): parameters are
    regenerated from the seed on the test side; only samples and float64 checksums of the
    reference's outputs are stored."""
    dims = O.Dims()
    B, T, P = 128, 20, 49
    p = O.init_params(dims, seed)
    batch = O.synth_batch(dims, B, T, P, seed=seed + 1)
    feats, caps = batch["features"], batch["captions"]
    style_token = 4 + batch["domain"]
    out, (lit, intended) = ref_gru_step(dims, p, feats, caps, None, style_token)
    rng = np.random.default_rng(seed + 2)
    arrs = {"seed": np.int64(seed), "B": np.int64(B), "T": np.int64(T), "P": np.int64(P),
            "style_token": np.int64(style_token), "loss": out["loss"]}
    bt = np.stack([rng.integers(0, B, 6), rng.integers(0, T, 6)], 1)
    arrs["logit_rows_bt"] = bt
    arrs["logit_rows"] = np.stack([out["logits"][b, t].numpy() for b, t in bt])
    arrs["logits_sum"] = np.float64(out["logits"].double().sum())
    arrs["logits_sumsq"] = np.float64((out["logits"].double() ** 2).sum())
    arrs["argmax_tokens"] = out["logits"].argmax(-1).numpy().astype(np.int32)
    # top-2 margin, so the argmax bit-exactness test can skip genuine near-ties
    top2 = out["logits"].topk(2, dim=-1).values
    arrs["argmax_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
    arrs["alphas_rows"] = np.stack([out["alphas"][b, t].numpy() for b, t in bt])
    ti = rng.integers(0, dims.theta_size, 512)
    arrs["theta_idx"] = ti; arrs["theta_vals"] = out["theta"].numpy()[ti]
    arrs["theta_sumsq"] = np.float64((out["theta"].double() ** 2).sum())
    arrs["dtheta_vals"] = out["dtheta"].numpy()[ti]
    arrs["dtheta_sumsq"] = np.float64((out["dtheta"].double() ** 2).sum())
    gall = dict(lit); gall.update(intended)
    norms = {}
    for k, g in gall.items():
        norms[k] = float(g.double().norm())
        flat = g.flatten()
        idx = rng.integers(0, flat.numel(), 64)
        arrs["gidx/" + k] = idx
        arrs["gval/" + k] = flat.numpy()[idx]
    save_npz(os.path.join(OUT, name + ".npz"), **arrs)
    return {"grad_norms": norms, "heads": O.head_layout(dims)}



def lstm_full_case(name, seed):
    """BASELINE config 3 at full size: HyperNet(cell='lstm') -- a hypernet-generated LSTMCell behind the same feature_fc,
    B=128, T=20, P=49, D=2048, F=E=H=200, V=9684, he=200 (257 M hypernet parameters).  The reference never wraps its
    AttentionLstm in a hypernet and that class has no feature_fc, so no reference module exists at this configuration:
    the vectors come from the ORACLE, after asserting here that the oracle reproduces the reference's own AttentionLstm +
    flip/set_all_parameters run (tests/golden/lstm_tiny.npz, written by lstm_case above from the reference's class)."""
    z = np.load(os.path.join(OUT, "lstm_tiny.npz"))
    td = O.Dims(D=12, F=12, E=8, H=8, V=40, he=8, cell="lstm")
    tp = {k[2:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith("p/")}
    th = O.hyper_forward(tp, torch.from_numpy(z["x_style"]))
    lg, al = O.decoder_forward(td, tp, O.split_theta(td, th), torch.from_numpy(z["features"]), torch.from_numpy(z["captions"]),
                               use_feature_fc=False)
    pin = {"theta": float((th - torch.from_numpy(z["theta"])).abs().max()),
           "logits": float((lg - torch.from_numpy(z["logits"])).abs().max()),
           "alphas": float((al - torch.from_numpy(z["alphas"])).abs().max())}
    assert pin["theta"] < 2e-6 and pin["logits"] < 2e-6 and pin["alphas"] < 1e-6, pin
    dims = O.Dims(cell="lstm")
    B, T, P = 128, 20, 49
    p = O.init_params(dims, seed)
    batch = O.synth_batch(dims, B, T, P, seed=seed + 1)
    feats, caps = batch["features"], batch["captions"]
    style_token = 4 + batch["domain"]
    loss, logits, alphas, theta, grads = O.forward_backward(dims, p, None, feats, caps, style_token=style_token)
    rng = np.random.default_rng(seed + 2)
    arrs = {"seed": np.int64(seed), "B": np.int64(B), "T": np.int64(T), "P": np.int64(P),
            "style_token": np.int64(style_token), "loss": loss}
    bt = np.stack([rng.integers(0, B, 6), rng.integers(0, T, 6)], 1)
    arrs["logit_rows_bt"] = bt
    arrs["logit_rows"] = np.stack([logits[b, t].numpy() for b, t in bt])
    arrs["logits_sum"] = np.float64(logits.double().sum())
    arrs["logits_sumsq"] = np.float64((logits.double() ** 2).sum())
    arrs["argmax_tokens"] = logits.argmax(-1).numpy().astype(np.int32)
    top2 = logits.topk(2, dim=-1).values
    arrs["argmax_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
    arrs["alphas_rows"] = np.stack([alphas[b, t].numpy() for b, t in bt])
    ti = rng.integers(0, dims.theta_size, 512)
    dtheta = grads.pop("dtheta")
    arrs["theta_idx"] = ti; arrs["theta_vals"] = theta.numpy()[ti]
    arrs["theta_sumsq"] = np.float64((theta.double() ** 2).sum())
    arrs["dtheta_vals"] = dtheta.numpy()[ti]
    arrs["dtheta_sumsq"] = np.float64((dtheta.double() ** 2).sum())
    norms = {}
    for k in O.trainable_names(p):
        g = grads[k]
        norms[k] = float(g.double().norm())
        flat = g.flatten()
        idx = rng.integers(0, flat.numel(), 64)
        arrs["gidx/" + k] = idx
        arrs["gval/" + k] = flat.numpy()[idx]
    save_npz(os.path.join(OUT, name + ".npz"), **arrs)
    return {"grad_norms": norms, "heads": O.head_layout(dims),
            "generated_by": "oracle/caphn_oracle.py (no reference module exists at this configuration); the oracle was asserted "
                            "equal to the reference's AttentionLstm + flip/set_all_parameters on lstm_tiny in the same run",
            "oracle_vs_reference_on_lstm_tiny_max_abs": pin}


def layers_case(name, seed, num_layers=3):
    """Reference AttentionGru(num_layers > 1) (models/decoderlstm.py:34-36, :65-67, :101-103) with ORDINARY cell parameters (no
    hypernet: the extra GRUCells are not generated in hypernet_attention.py, which builds its hypernet over captioner.gru only).
    Teacher-forced logits / alphas / loss / all parameter gradients, the free-running forward, and greedy_search."""
    dims = O.Dims(D=32, F=16, E=16, H=16, V=50, he=16)
    B, T, P = 4, 7, 7
    p = O.init_params(dims, seed)
    batch = O.synth_batch(dims, B, T, P, seed=seed + 1)
    feats, caps = batch["features"], batch["captions"]
    torch.manual_seed(seed)
    cap = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, num_layers=num_layers, p=0.0)
    msd = {k[len("captioner."):]: v.clone() for k, v in p.items() if k.startswith("captioner.")}
    res = cap.load_state_dict(msd, strict=False)
    assert all(k.startswith("gru.") or k.startswith("layers.") for k in res.missing_keys), res
    rng = np.random.default_rng(seed + 7)
    with torch.no_grad():
        for n, q in cap.named_parameters():
            if n.startswith("gru.") or n.startswith("layers."):
                q.copy_(torch.from_numpy(rng.uniform(-1.2, 1.2, size=tuple(q.shape)).astype(np.float32)))
        # three stacked random cells squash h towards a fixed point: sharpen the token path so that decoded sequences differ
        cap.embed.weight.mul_(8.0); cap.fc.weight.mul_(8.0)
    arrs = {"features": feats, "captions": caps, "num_layers": np.int64(num_layers)}
    for n, q in cap.named_parameters():
        arrs["p/" + n] = q.detach().clone()
    logits, alphas = cap(feats, caps.long(), 0.0)
    loss = F.cross_entropy(logits.view(-1, dims.V), caps.view(-1).long(), ignore_index=0)
    loss.backward()
    arrs.update({"logits": logits, "alphas": alphas, "loss": loss})
    for n, q in cap.named_parameters():
        arrs["g/" + n] = q.grad.detach().clone()
    with torch.no_grad():
        lf, af = cap(feats, caps.long(), 1.0)          # np.random.random() < 1.0: every step t >= 1 samples
    arrs["logits_free"] = lf; arrs["alphas_free"] = af
    top2 = lf.topk(2, dim=-1)[0]
    margin_free = float((top2[..., 0] - top2[..., 1])[:, :-1].min())
    # greedy_search per image on feature_fc outputs, with </s> made reachable
    cap.eval()
    with torch.no_grad():
        cap.fc.bias[2] += 4.0
        arrs["fc_bias_search"] = cap.fc.bias.detach().clone()
        ff = cap.feature_fc(feats)
        lens = []
        for b in range(B):
            sent, wts = cap.greedy_search(ff[b:b + 1], 2, 12)
            arrs[f"greedy_{b}"] = np.array(sent, dtype=np.int64)
            arrs[f"greedy_w_{b}"] = torch.cat(wts, 0)
            lens.append(len(sent))
    save_npz(os.path.join(OUT, name + ".npz"), **arrs)
    return {"num_layers": num_layers, "B": B, "T": T, "P": P, "greedy_lengths": lens, "loss": float(loss),
            "min_top2_margin_free": margin_free}


def ref_captioner(dims, p, x_style):
    """Reference AttentionGru with theta from the restated hypernet injected by the reference's own
    flip_parameters_to_tensors / set_all_parameters (as HyperNet.forward does, hypernet_attention.py:111-121)."""
    cap = AttentionGru(dims.D, dims.F, dims.E, dims.H, dims.V, p=0.0)
    hn_base, hn_heads = build_ref_hypernet(cap.gru, dims.he)
    msd = {k[len("captioner."):]: v.clone() for k, v in p.items() if k.startswith("captioner.")}
    cap.load_state_dict(msd, strict=False)
    hn_base.load_state_dict({k[len("hn_base."):]: v.clone() for k, v in p.items() if k.startswith("hn_base.")})
    hn_heads.load_state_dict({k[len("hn_heads."):]: v.clone() for k, v in p.items() if k.startswith("hn_heads.")})
    base_feat = hn_base(x_style)
    theta = torch.cat([h(base_feat).flatten() for h in hn_heads], dim=0)
    flip_parameters_to_tensors(cap.gru)
    set_all_parameters(cap.gru, theta.reshape(1, -1))
    return cap.eval(), theta.detach()


def ref_beam_search(cap, features, k, vocab_size, end_token):
    """The beam loop of HyperNet.test_step (hypernet_attention.py:251-306) RESTATED here -- the class cannot be
    constructed offline (see module docstring) -- around the reference's own sub-modules feature_fc /
    init_hidden / embed / attention / gru / fc.  Metric logging (:313-320) is dropped."""
    encoder_out = cap.feature_fc(features)
    encoder_dim = encoder_out.size(-1)
    encoder_out = encoder_out.view(1, -1, encoder_dim)
    num_pixels = encoder_out.size(1)
    encoder_out = encoder_out.expand(k, num_pixels, encoder_dim)
    k_prev_words = torch.LongTensor([[0]] * k)
    seqs = k_prev_words
    top_k_scores = torch.zeros(k, 1)
    complete_seqs, complete_seqs_scores = [], []
    step, compute, margin = 1, False, float("inf")
    h = cap.init_hidden(encoder_out)
    while True:
        embeddings = cap.embed(k_prev_words).squeeze(1)
        if k_prev_words[0][0] == 0:
            embeddings[:] = 0
        context, _ = cap.attention(encoder_out, h)
        h = cap.gru(torch.cat([embeddings, context], 1), h)
        scores = F.log_softmax(cap.fc(h), dim=1)
        scores = top_k_scores.expand_as(scores) + scores
        flat = scores[0] if step == 1 else scores.view(-1)
        vals, _ = flat.topk(k + 1, 0, True, True)
        margin = min(margin, float((vals[:-1] - vals[1:]).min()))
        top_k_scores, top_k_words = flat.topk(k, 0, True, True)
        prev_word_inds = top_k_words // vocab_size
        next_word_inds = top_k_words % vocab_size
        seqs = torch.cat([seqs[prev_word_inds], next_word_inds.unsqueeze(1)], dim=1)
        incomplete_inds = [ind for ind, w in enumerate(next_word_inds) if w != end_token]
        complete_inds = list(set(range(len(next_word_inds))) - set(incomplete_inds))
        if len(complete_inds) > 0:
            compute = True
            complete_seqs.extend(seqs[complete_inds].tolist())
            complete_seqs_scores.extend(top_k_scores[complete_inds])
        k -= len(complete_inds)
        if k == 0:
            break
        seqs = seqs[incomplete_inds]
        h = h[prev_word_inds[incomplete_inds]]
        encoder_out = encoder_out[prev_word_inds[incomplete_inds]]
        top_k_scores = top_k_scores[incomplete_inds].unsqueeze(1)
        k_prev_words = next_word_inds[incomplete_inds].unsqueeze(1)
        if step > 50:
            compute = False
            break
        step += 1
    best = complete_seqs[complete_seqs_scores.index(max(complete_seqs_scores))] if compute else None
    return best, [float(x) for x in complete_seqs_scores], complete_seqs, margin, step


def search_case(name, dims, n_images, P, seed, end_bump, k=3, max_sentence=12, sharpen=(1.0, 1.0, 1.0), write=True):
    """Greedy search by the reference's AttentionGru.greedy_search (models/decoderlstm.py:138-175) and the
    beam loop above, image by image.  fc.bias[</s>] is raised by end_bump so that random-init captions end."""
    p = O.init_params(dims, seed)
    p["captioner.fc.bias"] = p["captioner.fc.bias"].clone()
    p["captioner.fc.bias"][2] += end_bump
    # random-init logits are nearly flat and image-independent: scale fc / embed / feature_fc so that captions differ
    p["captioner.fc.weight"] = p["captioner.fc.weight"] * sharpen[0]
    p["captioner.embed.weight"] = p["captioner.embed.weight"] * sharpen[1]
    p["captioner.feature_fc.2.weight"] = p["captioner.feature_fc.2.weight"] * sharpen[2]
    torch.manual_seed(seed)
    feats = O.synth_batch(dims, n_images, 4, P, seed=seed + 1)["features"]
    x_style = torch.zeros(dims.he); x_style[seed % dims.he] = 1.0
    L = 52
    with torch.no_grad():
        cap, theta = ref_captioner(dims, p, x_style)
        arrs = {"features": feats, "x_style": x_style, "theta": theta, "end_token": np.int64(2), "beam": np.int64(k),
                "max_sentence": np.int64(max_sentence)}
        for kk, v in p.items():
            arrs["p/" + kk] = v
        g_tok = np.zeros((n_images, max_sentence), np.int64); g_len = np.zeros(n_images, np.int64)
        g_alpha = np.zeros((n_images, max_sentence, P), np.float32)
        b_seq = np.zeros((n_images, L), np.int64); b_len = np.zeros(n_images, np.int64)
        b_score = np.zeros(n_images, np.float32); b_fin = np.zeros(n_images, np.int64); b_margin = np.zeros(n_images, np.float64)
        b_steps = np.zeros(n_images, np.int64)
        comp = []
        for n in range(n_images):
            f_post = cap.feature_fc(feats[n:n + 1])
            sent, weights = cap.greedy_search(f_post, end_sentence=2, max_sentence=max_sentence)
            g_tok[n, :len(sent)] = sent; g_len[n] = len(sent)
            for t, wgt in enumerate(weights):
                g_alpha[n, t] = wgt[0].numpy()
            best, cs, cseqs, margin, steps = ref_beam_search(cap, feats[n:n + 1], k, dims.V, 2)
            b_margin[n] = margin; b_steps[n] = steps; b_fin[n] = best is not None
            if best is not None:
                b_seq[n, :len(best)] = best; b_len[n] = len(best); b_score[n] = max(cs)
            comp.append({"scores": cs, "seqs": cseqs})
    arrs.update({"greedy_tokens": g_tok, "greedy_len": g_len, "greedy_alphas": g_alpha, "beam_seq": b_seq, "beam_len": b_len,
                 "beam_score": b_score, "beam_finished": b_fin, "beam_margin": b_margin, "beam_steps": b_steps})
    if write:
        save_npz(os.path.join(OUT, name + ".npz"), **arrs)
    return {"greedy_len": g_len.tolist(), "beam_len": b_len.tolist(), "beam_finished": b_fin.tolist(),
            "beam_margin": b_margin.tolist(), "beam_steps": b_steps.tolist(), "complete": comp}



def _load_later_classes():
    """later.py holds DecoderGRU / DecoderRNN but has no import statements (it cannot be imported as a module).
    Its source is executed here in a namespace that supplies the names its authors had in scope.  Inert stand-ins:
    `open` / `pickle.load` (the constructors read data/vocab.pkl only to keep a vocabulary for text dumps; nothing is
    unpickled here) and `cap_to_text_gt` (id -> text, result unused by forward, later.py:449-450)."""
    import io
    ns = {"torch": torch, "nn": nn, "F": F, "np": np,
          "pickle": types.SimpleNamespace(load=lambda f: None),
          "open": lambda *a, **k: io.BytesIO(b""),
          "cap_to_text_gt": lambda *a, **k: ""}
    with open(os.path.join(REF, "later.py")) as f:
        src = f.read()
    exec(compile(src, os.path.join(REF, "later.py"), "exec"), ns)
    return ns["DecoderGRU"], ns["DecoderRNN"]


def plain_case(name, d, B, T, seed, style_token=4):
    """hypernet.py's step with the reference's own DecoderGRU / DecoderRNN (later.py) and its own
    flip_parameters_to_tensors / set_all_parameters (utils.py) on the whole captioner -- child-offset restart
    included; HyperNet.__init__ (hypernet.py:55-93) is restated (the class needs downloads).  V must be 9684:
    later.py:445 hard-codes it."""
    assert d.V == 9684
    DecoderGRU, DecoderRNN = _load_later_classes()
    p = O.init_plain_params(d, seed)
    Cls = DecoderGRU if d.cell == "gru" else DecoderRNN
    cap = Cls(d.E, d.H, d.V, num_layers=d.L, dropout=False)
    E = d.E
    hn_base = nn.Sequential(nn.Linear(E, 4 * E), nn.LeakyReLU(), nn.Linear(4 * E, 8 * E), nn.LeakyReLU())
    heads, order = [], []
    for pname, W in cap.named_parameters():
        if pname in ('embed.weight', 'fc_out.weight', 'fc_out.bias'):
            continue
        order.append(pname)
        w_size = len(W.flatten())
        if w_size < 8 * E:
            heads.append(nn.Sequential(nn.Linear(8 * E, w_size), nn.LeakyReLU(), nn.Linear(w_size, w_size)))
        elif w_size // 8 < 8 * E:
            heads.append(nn.Sequential(nn.Linear(8 * E, 8 * E), nn.LeakyReLU(), nn.Linear(8 * E, w_size)))
        else:
            heads.append(nn.Sequential(nn.Linear(8 * E, w_size // 8), nn.LeakyReLU(), nn.Linear(w_size // 8, w_size)))
    hn_heads = nn.ModuleList(heads)
    assert order == [n for n, _ in d.cell_param_shapes()], order
    assert [(h[0].out_features, h[2].out_features) for h in hn_heads] == O.plain_head_layout(d)
    hn_base.load_state_dict({k[len("hn_base."):]: v.clone() for k, v in p.items() if k.startswith("hn_base.")})
    hn_heads.load_state_dict({k[len("hn_heads."):]: v.clone() for k, v in p.items() if k.startswith("hn_heads.")})
    res = cap.load_state_dict({k[len("captioner."):]: v.clone() for k, v in p.items() if k.startswith("captioner.")}, strict=False)
    assert not res.unexpected_keys and all(k.startswith(("lstm_cell.", "layers.")) for k in res.missing_keys), res
    rng = np.random.default_rng(seed + 1)
    feats = torch.from_numpy(rng.standard_normal((B, E), dtype=np.float32)).requires_grad_(True)
    caps = torch.from_numpy(rng.integers(0, d.V, size=(B, T)))
    caps[:, 0] = 1
    caps[0, T - 1] = 0                                  # a <pad> target: this loss has no ignore_index (hypernet.py:146)
    x = cap.embed(torch.tensor([style_token]))          # hypernet.py:127-131
    base_feat = hn_base(x)
    theta = torch.cat([h(base_feat).flatten() for h in hn_heads], dim=0)
    flip_parameters_to_tensors(cap)
    n_set = set_all_parameters(cap, theta.reshape(1, -1))
    torch.manual_seed(seed)
    h0 = torch.rand(size=(B, d.H)) if d.cell == "gru" else torch.zeros(B, d.H)
    torch.manual_seed(seed)                             # DecoderGRU.forward draws the same torch.rand (later.py:397)
    logits = cap(feats, caps, True)
    loss = F.cross_entropy(logits.view(-1, d.V), caps.view(-1).long())
    loss.backward()
    # literal leaf grads of the injected Parameters, then dtheta by the same overlapping slicing, then the hypernet VJP
    cells = [cap.lstm_cell] + (list(cap.layers) if cap.layers else [])
    dtheta = torch.zeros_like(theta)
    lit = {}
    for li, cell in enumerate(cells):
        off = 0
        for nm in cell.registered_parameters_name:
            gq = getattr(cell, nm).grad
            lit[("lstm_cell." if li == 0 else f"layers.{li - 1}.") + nm] = gq.detach().clone()
            dtheta[off:off + gq.numel()] += gq.flatten()
            off += gq.numel()
    theta.backward(dtheta)
    cols = np.sort(rng.choice(d.V, size=192, replace=False)); cols[:3] = [0, 1, 2]; cols = np.sort(np.unique(cols))
    rows = np.unique(np.concatenate([caps.numpy().reshape(-1), [style_token]]))
    arrs = {"features": feats.detach(), "captions": caps, "h0": h0, "style_token": np.int64(style_token),
            "seed": np.int64(seed), "theta": theta.detach(), "dtheta": dtheta, "loss": loss.detach(),
            "cols": cols, "logits_cols": logits.detach()[:, :, cols], "tokens": logits.detach().argmax(-1),
            "logits_lse": torch.logsumexp(logits.detach(), -1), "logits_sum": np.float64(logits.detach().double().sum()),
            "dfeatures": feats.grad.detach(), "rows": rows,
            "gembed_rows": cap.embed.weight.grad.detach()[rows], "gembed_norm": np.float64(cap.embed.weight.grad.double().norm()),
            "gfc_w_rows": cap.fc_out.weight.grad.detach()[rows], "gfc_w_norm": np.float64(cap.fc_out.weight.grad.double().norm()),
            "gfc_b_rows": cap.fc_out.bias.grad.detach()[rows], "gfc_b_norm": np.float64(cap.fc_out.bias.grad.double().norm())}
    for k, v in lit.items():
        arrs["glit/" + k] = v
    for n, q in hn_base.named_parameters():
        arrs["gint/hn_base." + n] = q.grad.detach()
    for n, q in hn_heads.named_parameters():
        if q.grad.numel() > 20000:       # rank-1 (dtheta slice x hidden activations): keep 32 rows and the norm
            arrs["gint_rows/hn_heads." + n] = q.grad.detach()[:32]
            arrs["gint_norm/hn_heads." + n] = np.float64(q.grad.double().norm())
        else:
            arrs["gint/hn_heads." + n] = q.grad.detach()
    save_npz(os.path.join(OUT, name + ".npz"), **arrs)
    return {"order": order, "n_set": int(n_set), "heads": O.plain_head_layout(d), "theta_size": int(theta.numel()),
            "dims": {"E": d.E, "H": d.H, "V": d.V, "L": d.L, "cell": d.cell}, "B": B, "T": T,
            "dtheta_tail_abs_sum": float(dtheta[sum(int(np.prod(sh)) for _, sh in d.cell_param_shapes()[:4]):].abs().sum())}


PLAIN_CASES = {   # name: (dims, B, T, seed); E=12,H=10 -> head branches 2 and 1; E=4,H=24 -> branches 3 and 2
    "plain_gru_l2": (dict(E=12, H=10, V=9684, L=2, cell="gru"), 3, 5, 71),
    "plain_gru_b3": (dict(E=4, H=24, V=9684, L=1, cell="gru"), 2, 4, 72),
    "plain_lstm_l2": (dict(E=12, H=10, V=9684, L=2, cell="lstm"), 3, 5, 73),
}

# --------------------------------------------------------------------------
# N4: the reference's own baseline/transformer.py (pure torch, imported by path) on seeded inputs
# --------------------------------------------------------------------------
CATR_CASES = {   # name: (pre_norm, enc_layers, dec_layers, seed)
    "catr_prenorm": (True, 2, 2, 91),
    "catr_postnorm": (False, 1, 2, 92),
}


def catr_case(name, pre_norm, enc_layers, dec_layers, seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_catr_transformer", os.path.join(REF, "baseline", "transformer.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    class Cfg:        # the fields baseline/transformer.py reads from baseline/configuration.py, at test size; dropout off
        hidden_dim = 32; pad_token_id = 0; max_position_embeddings = 12; layer_norm_eps = 1e-12; dropout = 0.0
        vocab_size = 50; nheads = 4; dim_feedforward = 64
    Cfg.enc_layers, Cfg.dec_layers, Cfg.pre_norm = enc_layers, dec_layers, pre_norm
    torch.manual_seed(seed)
    net = ref.build_transformer(Cfg).double()
    g = torch.Generator().manual_seed(seed + 1)
    bs, h, w, T = 3, 3, 5, Cfg.max_position_embeddings
    src = torch.randn(bs, Cfg.hidden_dim, h, w, generator=g, dtype=torch.float64)
    pos = torch.randn(bs, Cfg.hidden_dim, h, w, generator=g, dtype=torch.float64)
    mask = torch.zeros(bs, h, w, dtype=torch.bool)
    mask[1, :, 4:] = True
    mask[2, 2:, :] = True
    tgt = torch.randint(1, Cfg.vocab_size, (bs, T), generator=g)
    tgt_mask = torch.zeros(bs, T, dtype=torch.bool)
    tgt[1, 9:] = 0; tgt_mask[1, 9:] = True
    tgt[2, 5:] = 0; tgt_mask[2, 5:] = True
    src.requires_grad_(True)
    hs = net(src, mask, pos, tgt, tgt_mask)
    R = torch.randn(hs.shape, generator=g, dtype=torch.float64)
    (hs * R).sum().backward()
    arrs = {"src": src.detach().float(), "pos": pos.float(), "mask": mask, "tgt": tgt, "tgt_mask": tgt_mask, "R": R.float(),
            "hs": hs.detach(), "dsrc": src.grad}
    for k, v in net.state_dict().items():
        arrs["p/" + k] = v.float()
    for k, v in net.named_parameters():
        arrs["g/" + k] = v.grad if v.grad is not None else torch.zeros_like(v)
    save_npz(os.path.join(OUT, name + ".npz"), **arrs)
    return {"pre_norm": pre_norm, "enc_layers": enc_layers, "dec_layers": dec_layers, "hidden_dim": Cfg.hidden_dim, "nheads": Cfg.nheads,
            "dim_feedforward": Cfg.dim_feedforward, "vocab_size": Cfg.vocab_size, "max_position_embeddings": T,
            "n_params": int(sum(p.numel() for p in net.parameters())), "hs_abs_max": float(hs.abs().max())}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    meta = {"torch": torch.__version__, "reference": "zacharie12/Hypernet-image-captioning @ /root/reference"}
    if "--only-catr" in sys.argv:              # the CATR transformer vectors only
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        for nm, args in CATR_CASES.items():
            meta[nm] = catr_case(nm, *args)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(json.dumps({k: meta[k] for k in CATR_CASES}, indent=1))
        return
    if "--only-lstm-full" in sys.argv:         # BASELINE config 3 at full size (oracle-generated, oracle pinned on lstm_tiny)
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["lstm_full"] = lstm_full_case("lstm_full", seed=2025)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(json.dumps({k: v for k, v in meta["lstm_full"].items() if k != "grad_norms"}, indent=1))
        return
    if "--only-layers" in sys.argv:            # AttentionGru(num_layers = 3)
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["gru_layers"] = layers_case("gru_layers", seed=57)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(json.dumps(meta["gru_layers"], indent=1))
        return
    if "--only-plain" in sys.argv:             # refresh the hypernet.py / later.py vectors only
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        for nm, (dd, B, T, sd) in PLAIN_CASES.items():
            meta[nm] = plain_case(nm, O.PlainDims(**dd), B, T, sd)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(json.dumps({k: meta[k] for k in PLAIN_CASES}, indent=1)[:3000])
        return
    if "--only-search" in sys.argv:            # refresh the search vectors, keep everything else as committed
        with open(os.path.join(OUT, "meta.json")) as f:
            meta = json.load(f)
        meta["gru_search"] = search_case("gru_search", O.Dims(D=32, F=16, E=16, H=16, V=50, he=16), n_images=6, P=7,
                                         seed=462, end_bump=SEARCH_END_BUMP, sharpen=SEARCH_SHARPEN)
        with open(os.path.join(OUT, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(json.dumps(meta["gru_search"], indent=1)[:3000])
        return
    tiny = O.Dims(D=32, F=16, E=16, H=16, V=50, he=16)
    meta["gru_tiny_flickr"] = tiny_case("gru_tiny_flickr", tiny, B=3, T=6, P=7, seed=11, flickr=True, max_norm=5.0)
    # he = 2 forces head branch 3 (w//500 >= he) for weight_ih and a 1-D CC-style x; max_norm small
    # enough that clipping is active
    tiny_cc = O.Dims(D=32, F=16, E=16, H=16, V=50, he=2)
    meta["gru_tiny_cc"] = tiny_case("gru_tiny_cc", tiny_cc, B=4, T=7, P=7, seed=23, flickr=False, max_norm=0.05)
    # odd sizes: nothing a multiple of 4/16/64
    odd = O.Dims(D=37, F=13, E=11, H=19, V=83, he=5)
    meta["gru_odd_cc"] = tiny_case("gru_odd_cc", odd, B=5, T=9, P=10, seed=31, flickr=False, max_norm=5.0)
    meta["lstm_tiny"] = lstm_case("lstm_tiny", seed=41)
    meta["gru_layers"] = layers_case("gru_layers", seed=57)
    meta["gru_search"] = search_case("gru_search", tiny, n_images=6, P=7, seed=462, end_bump=SEARCH_END_BUMP, sharpen=SEARCH_SHARPEN)
    for nm, (dd, B, T, sd) in PLAIN_CASES.items():
        meta[nm] = plain_case(nm, O.PlainDims(**dd), B, T, sd)
    for nm, args in CATR_CASES.items():
        meta[nm] = catr_case(nm, *args)
    if os.environ.get("CAPHN_GOLDEN_FULL", "1") == "1":
        meta["gru_full"] = full_case("gru_full", seed=2024)
        meta["lstm_full"] = lstm_full_case("lstm_full", seed=2025)
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in meta.items() if k != "gru_full"}, indent=1)[:2000])


if __name__ == "__main__":
    main()
