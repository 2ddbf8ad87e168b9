"""Full-size (BASELINE config: B=128, T=20, P=49, D=2048, F=E=H=200, V=9684, he=200) parity of the
fused HIP step against samples/checksums of the reference's own run (tests/golden/gru_full.npz),
plus size-independent properties."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import caphn_oracle as O
from helpers import GOLDEN, maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Vocab:
    w2i = {"<pad>": 0}

    def __call__(self, w):
        return 3


@pytest.fixture(scope="module")
def full():
    from hypernet_attention import HyperNet
    from caphn.engine import FusedTrainer
    z = np.load(os.path.join(GOLDEN, "gru_full.npz"))
    dims = O.Dims()
    seed, B, T, P = int(z["seed"]), int(z["B"]), int(z["T"]), int(z["P"])
    p = O.init_params(dims, seed)
    batch = O.synth_batch(dims, B, T, P, seed=seed + 1)
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, _Vocab())
    res = net.load_state_dict(p, strict=False)
    assert all(k.startswith("captioner.gru.") for k in res.missing_keys) and not res.unexpected_keys
    net = net.to(DEV)
    tr = FusedTrainer(net, lr=1e-3, max_norm=5.0)
    return z, dims, p, batch, net, tr


def test_full_size_forward_backward(full):
    z, dims, p, batch, net, tr = full
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    tok = int(z["style_token"])
    loss = tr.forward_backward(feats, caps, style_token=tok, validate=True)
    # forward logits were overwritten in place by d logits; recompute the forward for the logits check
    buf = tr._buffers(*[int(z[k]) for k in ("B", "T", "P")])
    from caphn import ops
    import dataclasses
    params = tr._dec_tensors(tr._theta, grads=False)
    full_dims = dataclasses.replace(buf["dims"], rows=False)      # every logits row, as the module API returns them
    logits, alphas = ops.decoder_forward(full_dims, params, feats, caps, buf["ws"])
    assert abs(float(loss[0]) - float(z["loss"])) < 5e-6
    lg = logits.cpu().numpy()
    for i, (b, t) in enumerate(z["logit_rows_bt"]):
        assert np.abs(lg[b, t] - z["logit_rows"][i]).max() < 5e-6          # north star: 1e-6 class, fp32
        assert np.abs(alphas[b, t].cpu().numpy() - z["alphas_rows"][i]).max() < 1e-6
    # 24.8 M logits: a mean offset of 1e-8 per element (summation order / split-bf16 truncation) is 0.25 on the sum
    assert abs(float(logits.double().sum()) - float(z["logits_sum"])) < 0.25      # 1e-8 mean offset per logit
    assert abs(float((logits.double() ** 2).sum()) / float(z["logits_sumsq"]) - 1) < 1e-6
    # token argmax bit-exact wherever the reference's own top-2 margin exceeds fp32 noise
    am = logits.argmax(-1).cpu().numpy()
    safe = z["argmax_margin"] > 2e-5
    assert safe.mean() > 0.99 and (am[safe] == z["argmax_tokens"][safe]).all()
    th = tr._theta.cpu().numpy()
    assert np.abs(th[z["theta_idx"]] - z["theta_vals"]).max() < 2e-6
    assert abs(float((tr._theta.double() ** 2).sum()) / float(z["theta_sumsq"]) - 1) < 1e-6
    dth = tr.flat_g[:tr.theta_size]
    assert np.abs(dth.cpu().numpy()[z["theta_idx"]] - z["dtheta_vals"]).max() < 2e-6
    assert abs(float((dth.double() ** 2).sum()) / float(z["dtheta_sumsq"]) - 1) < 1e-4
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    for k, n in meta["gru_full"]["grad_norms"].items():
        if k.startswith("hn_heads.") and k.endswith(".2.weight"):
            got = tr.w2_grad_dense(int(k.split(".")[1]))
        else:
            got = tr.grad(k)
        mine = float(got.double().norm())
        assert abs(mine - n) < 1e-4 * max(n, 1e-3), (k, mine, n)
        idx = z["gidx/" + k]
        assert np.abs(got.flatten().cpu().numpy()[idx] - z["gval/" + k]).max() < 2e-6, k
        del got


def test_full_size_step_properties(full):
    """Size-independent properties at the full configuration: deterministic forward (bit-exact
    re-run), loss decreases over Adam steps on a fixed batch, parameters stay finite, the clip
    coefficient matches a recomputation from the gradient arena."""
    z, dims, p, batch, net, tr = full
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    x = torch.zeros(dims.he, device=DEV); x[3] = 1.0
    from caphn import ops
    import dataclasses
    buf = tr._buffers(128, 20, 49)
    params = tr._dec_tensors(tr._theta, grads=False)
    full_dims = dataclasses.replace(buf["dims"], rows=False)
    l1, a1 = ops.decoder_forward(full_dims, params, feats, caps, buf["ws"])
    l1 = l1.clone()
    l2, _ = ops.decoder_forward(full_dims, params, feats, caps, buf["ws"])
    assert torch.equal(l1, l2)
    assert float((a1.sum(-1) - 1).abs().max()) < 1e-5
    losses = []
    for _ in range(6):
        loss = tr.forward_backward(feats, caps, x_style=x)
        dense = float((tr.flat_g.double() ** 2).sum())
        rank = sum(float((tr.w2_grad_dense(i).double() ** 2).sum()) for i in range(2))
        rank += sum(float((tr.w2_grad_dense(i).double() ** 2).sum()) for i in range(2, 4))
        coef = tr.optimizer_step()
        tot = (dense + rank) ** 0.5
        assert abs(float(coef[1]) - tot) < 1e-4 * tot
        assert abs(float(coef[0]) - min(1.0, 5.0 / (tot + 1e-6))) < 1e-5
        losses.append(float(loss[0]))
    assert losses[-1] < losses[0] - 0.05, losses
    assert all(np.isfinite(losses))
    assert bool(torch.isfinite(tr.flat_p).all()) and all(bool(torch.isfinite(w.data).all()) for w in tr.W2)
