"""The reduced-precision side modes (caphn_tune key 11 / bench.py --dtype): never the default, never the headline.
  bf16    (key 11 = 1): every dense contraction as ONE bf16 MFMA product on operands rounded to bf16 at staging -- 8 significand
          bits over K = 200 ... 2048: measured 3.1e-3 on logits, OUTSIDE the north star's 1e-4 (asserted < 2x the measurement);
  bf16x2  (key 11 = 2): operands as TWO bf16 planes (hi + mid, 16 significand bits), three products, fp32 accumulation --
          asserted INSIDE 1e-4 on the fp32 golden logits at the full configuration, argmax equal wherever the fp32 top-2
          margin exceeds 1e-3.
Recurrent kernels, softmax, loss, Adam and master weights stay fp32 in both.  Reference vectors: tests/golden/gru_full.npz."""
import os

import numpy as np
import pytest
import torch

from oracle import caphn_oracle as O
from helpers import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Vocab:
    w2i = {"<pad>": 0}

    def __call__(self, w):
        return 3


@pytest.fixture(params=[1, 2], ids=["bf16", "bf16x2"])
def bf16_mode(request):
    from caphn import _lib
    lib = _lib.load()
    assert lib.caphn_tune(11, request.param) == 0
    yield request.param
    assert lib.caphn_tune(11, 0) == 0


def test_bf16_side_mode_against_fp32_golden(bf16_mode):
    from hypernet_attention import HyperNet
    from caphn.engine import FusedTrainer
    from caphn import ops
    import dataclasses
    z = np.load(os.path.join(GOLDEN, "gru_full.npz"))
    dims = O.Dims()
    seed, B, T, P = int(z["seed"]), int(z["B"]), int(z["T"]), int(z["P"])
    p = O.init_params(dims, seed)
    batch = O.synth_batch(dims, B, T, P, seed=seed + 1)
    net = HyperNet(dims.F, dims.E, dims.H, dims.V, _Vocab())
    net.load_state_dict(p, strict=False)
    net = net.to(DEV)
    tr = FusedTrainer(net, lr=1e-3, max_norm=5.0)
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    tok = int(z["style_token"])
    loss = tr.forward_backward(feats, caps, style_token=tok, validate=True)
    buf = tr._buffers(B, T, P)
    params = tr._dec_tensors(tr._theta, grads=False)
    logits, _ = ops.decoder_forward(dataclasses.replace(buf["dims"], rows=False), params, feats, caps, buf["ws"])
    lg = logits.cpu().numpy()
    err = max(float(np.abs(lg[b, t] - z["logit_rows"][i]).max()) for i, (b, t) in enumerate(z["logit_rows_bt"]))
    print(f"bf16 side mode: max |logits - fp32 reference| on the sampled rows = {err:.3e}; loss {float(loss[0]):.6f} vs {float(z['loss']):.6f}")
    am = logits.argmax(-1).cpu().numpy()
    agree = float((am == z["argmax_tokens"]).mean())
    print(f"side mode {bf16_mode}: token argmax agrees on {agree * 100:.2f} % of all positions")
    if bf16_mode == 2:
        assert 1e-7 < err < 1e-4                 # the north star's reduced-precision bound on logits
        assert abs(float(loss[0]) - float(z["loss"])) < 2e-5
        safe = z["argmax_margin"] > 1e-3
        assert (am[safe] == z["argmax_tokens"][safe]).all() and agree > 0.999
    else:
        assert 1e-5 < err < 6.5e-3               # really reduced precision; 2x the measured 3.1e-3 (a 5x regression fails)
        assert abs(float(loss[0]) - float(z["loss"])) < 2e-3
        safe = z["argmax_margin"] > 2e-2         # well above the deviation
        assert (am[safe] == z["argmax_tokens"][safe]).all()
        assert agree > 0.97
    # gradients: same direction as the fp32 reference's
    dth = tr.flat_g[:tr.theta_size].cpu().numpy()
    ref = z["dtheta_vals"]; got = dth[z["theta_idx"]]
    cos = float((ref * got).sum() / (np.linalg.norm(ref) * np.linalg.norm(got)))
    assert cos > (0.99999 if bf16_mode == 2 else 0.999)
    l0 = float(loss[0])
    tr.optimizer_step()
    for _ in range(5):
        l = tr.step(feats, caps, style_token=tok)
    assert float(l[0]) < l0 - 0.05
