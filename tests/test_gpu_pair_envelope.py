"""Envelope of the two-workgroups-per-caption recurrent kernels (csrc/recurrent_pair.hip; models/decoderlstm.py:78-108 is the loop
they run): batches that are no multiple of 8, more captions than the chip has CU pairs (a second wave of workgroups), a busy
side stream competing for CUs, a repeated backward on one workspace, and the failure path -- a partner that never answers must end
in CAPHN_ETIMEOUT and NaN, never in a plausible wrong gradient."""
import dataclasses

import pytest
import torch

from oracle import caphn_oracle as O
from helpers import dec_dims, dec_params_from_oracle, maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from caphn import ops as _ops
    return _ops


@pytest.fixture()
def lib():
    from caphn import _lib
    lib = _lib.load()
    yield lib
    lib.caphn_tune(9, 1); lib.caphn_tune(10, 0); lib.caphn_tune(22, 1000000)
    torch.cuda.synchronize()
    lib.caphn_device_error(1)


def _case(B, T=6, P=49, H=200, seed=3, hypernet=False):
    """Canonical recurrent shape (H = 200, P = 49) by default.  The cell weights are what the kernels see; at H = 200 they are drawn
    directly (GRUCell's own U(-1/sqrt(H), 1/sqrt(H))) instead of building the 145 M-parameter hypernet that would generate them."""
    import math
    import numpy as np
    dims = O.Dims(D=48, F=H, E=H, H=H, V=120, he=6)
    batch = O.synth_batch(dims, B=B, T=T, P=P, seed=seed + 1)
    if hypernet:
        p = O.init_params(dims, seed=seed)
        x = torch.zeros(dims.he); x[1] = 1.0
        return dims, p, batch, O.hyper_forward(p, x).detach()
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in O.decoder_param_shapes(dims):
        b = 1.0 if name == "captioner.embed.weight" else 1.0 / math.sqrt(shape[-1] if len(shape) > 1 else H)
        p[name] = torch.from_numpy(rng.uniform(-b, b, size=shape).astype(np.float32))
    theta = torch.from_numpy(rng.uniform(-1, 1, size=dims.theta_size).astype(np.float32)) / math.sqrt(H)
    return dims, p, batch, theta


def _run(ops, dims, p, batch, theta, ws=None, backward=True):
    B, T = batch["captions"].shape
    P = batch["features"].shape[1]
    dd = dec_dims(dims, B, T, P)
    params = dec_params_from_oracle(p, theta, dims, DEV)
    ws = ops.decoder_workspace(dd, DEV) if ws is None else ws
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    logits, alphas = ops.decoder_forward(dd, params, feats, caps, ws)
    out = {"logits": logits.clone(), "alphas": alphas.clone()}
    if backward:
        _, dl = ops.cross_entropy_fwd_bwd(logits, caps, 0)
        grads = {n: torch.full(s, float("nan"), device=DEV) for n, s in dd.param_shapes().items()}
        ops.decoder_backward(dd, params, feats, caps, dl.clone(), grads, ws)
        out["grads"] = grads
        out["ctx"] = (dd, params, feats, caps, dl, ws)
    return out


@pytest.mark.parametrize("B", [129, 136, 256])
def test_pair_kernels_match_single_workgroup_kernels_beyond_one_wave(ops, lib, B):
    """B = 129 / 136: partial group of eight; B = 256: 512 workgroups on 256 CUs -- the second half of the grid only starts as
    the first finishes.  The pair kernels add the two halves' partial sums in a fixed order, so they are deterministic (two
    runs: identical bits) but not bit-equal to the one-workgroup kernels, whose sums run over all k in one sweep: 2e-6."""
    dims, p, batch, theta = _case(B)
    lib.caphn_tune(9, 1)
    a = _run(ops, dims, p, batch, theta)
    a2 = _run(ops, dims, p, batch, theta, backward=False)
    lib.caphn_tune(9, 0)
    s = _run(ops, dims, p, batch, theta)
    torch.cuda.synchronize()
    assert lib.caphn_device_error(0) == 0
    assert torch.equal(a["logits"], a2["logits"]) and torch.equal(a["alphas"], a2["alphas"])
    assert maxdiff(a["logits"], s["logits"]) < 2e-6 and maxdiff(a["alphas"], s["alphas"]) < 1e-6
    for n in a["grads"]:
        assert bool(torch.isfinite(a["grads"][n]).all()), n
        assert maxdiff(a["grads"][n], s["grads"][n]) < 2e-6, n


def test_pair_kernels_against_the_oracle_at_an_odd_batch(ops, lib):
    dims, p, batch, theta = _case(13, T=5, P=11, H=24, hypernet=True)
    x = torch.zeros(dims.he); x[1] = 1.0
    loss, logits_ref, alphas_ref, _, gref = O.forward_backward(dims, p, x, batch["features"], batch["captions"])
    a = _run(ops, dims, p, batch, theta)
    assert maxdiff(a["logits"].cpu(), logits_ref) < 2e-6 and maxdiff(a["alphas"].cpu(), alphas_ref) < 1e-6
    for n, g in a["grads"].items():
        if not n.startswith("gru."):
            assert maxdiff(g.cpu(), gref["captioner." + n]) < 2e-6, n


def test_second_backward_on_the_same_workspace(ops, lib):
    """The backward's exchange area is cleared only by the forward's prep launch: without the per-launch epoch in the tags a second
    backward found the first one's granules already carrying matching tags (stale d alpha / dh partials accepted at random)."""
    dims, p, batch, theta = _case(24, T=7)
    a = _run(ops, dims, p, batch, theta)
    dd, params, feats, caps, dl, ws = a["ctx"]
    for _ in range(3):
        g2 = {n: torch.full_like(t, float("nan")) for n, t in a["grads"].items()}
        ops.decoder_backward(dd, params, feats, caps, dl.clone(), g2, ws)
        for n in g2:
            assert maxdiff(g2[n], a["grads"][n]) < 1e-6, n      # (split-K atomics: not bit-identical run to run)
    torch.cuda.synchronize()
    assert lib.caphn_device_error(0) == 0


def test_pair_kernels_with_a_busy_side_stream(ops, lib):
    """Another queue keeps CUs occupied while the pair kernels run (what an RCCL kernel or a caller's own side stream does): the
    partners of a pair may then start at different times; results must not change and nothing may time out."""
    dims, p, batch, theta = _case(128, T=8)
    quiet = _run(ops, dims, p, batch, theta)
    side = torch.cuda.Stream()
    A = torch.randn(4096, 2048, device=DEV); Bm = torch.randn(4096, 2048, device=DEV)
    out = torch.empty(4096, 4096, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(40):
            ops.gemm(A, Bm, False, True, out=out)
    busy = _run(ops, dims, p, batch, theta)
    torch.cuda.synchronize()
    assert lib.caphn_device_error(0) == 0
    assert torch.equal(busy["logits"], quiet["logits"])
    for n in busy["grads"]:
        assert maxdiff(busy["grads"][n], quiet["grads"][n]) < 1e-6, n


def test_a_partner_that_never_answers_is_reported_not_hidden(ops, lib):
    from caphn._lib import CaphnError
    dims, p, batch, theta = _case(16, T=4)
    good = _run(ops, dims, p, batch, theta, backward=False)
    lib.caphn_tune(22, 2000)          # give up after 2 ms instead of 1 s
    lib.caphn_tune(10, 2)             # test mode: half 1 of every pair never sends
    bad = _run(ops, dims, p, batch, theta, backward=False)
    torch.cuda.synchronize()
    assert lib.caphn_device_error(0) == -4
    assert not bool(torch.isfinite(bad["logits"]).all())           # NaN, not stale bits
    with pytest.raises(CaphnError, match="CAPHN_ETIMEOUT"):       # sticky: the next call that launches work says so
        ops.zero_(torch.empty(16, device=DEV))
    assert lib.caphn_device_error(1) == -4 and lib.caphn_device_error(0) == 0
    lib.caphn_tune(10, 0); lib.caphn_tune(22, 1000000)
    again = _run(ops, dims, p, batch, theta, backward=False)
    torch.cuda.synchronize()
    assert lib.caphn_device_error(0) == 0 and torch.equal(again["logits"], good["logits"])
