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


@pytest.mark.parametrize("H,B", [(200, 9), (12, 3)])
def test_prep_ahead_and_weights_packed_by_the_adam_pass(ops, lib, H, B):
    """caphn_decoder_pair_prep (exchange areas, d Hs, U_a rows) + caphn_adam_rank_multi_f32 with caphn_rank_job::next_pack (the W_hh
    rows, written while the pass produces the next theta) leave the pair kernels' packed weight copy exactly as the forward's own prep
    launch builds it from that theta -- and a forward given dims.precomputed bit 128 on top of them reproduces the ordinary forward and
    backward bit for bit (models/decoderlstm.py:78-108; the optimiser of cc_train_hypernet.py:110-120 in front of it)."""
    dims, p, batch, theta = _case(B, H=H, P=49 if H == 200 else 5, seed=8)
    Bc, T = batch["captions"].shape
    P = batch["features"].shape[1]
    dd = dec_dims(dims, Bc, T, P)
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    # an "Adam pass" over a stand-in second layer W2 [3 H H, k] whose next theta = W2' a + b is the next W_hh
    g = torch.Generator().manual_seed(H)
    k, rows = 16, 3 * H * H
    W = (torch.randn(rows, k, generator=g) * 0.05).to(DEV)
    m, v = torch.zeros_like(W), torch.zeros_like(W)
    gf, af = (torch.randn(1, rows, generator=g) * 0.01).to(DEV), torch.randn(1, k, generator=g).to(DEV)
    na, nb = torch.randn(k, generator=g).to(DEV), (torch.randn(rows, generator=g) * 0.05).to(DEV)
    whh_next = torch.zeros(rows, device=DEV)
    coef = torch.tensor([1.0, 0.0], device=DEV)
    ws_a = ops.decoder_workspace(dd, DEV)
    ws_a.zero_()
    pk = ops.decoder_pair_pack_desc(dd, ws_a)
    assert pk is not None and pk.H == H
    params0 = dec_params_from_oracle(p, theta, dims, DEV)
    ops.decoder_pair_prep(dd, params0, ws_a)
    ops.adam_rank_multi([[W, m, v, gf, af, na, nb, whh_next]], coef, 1e-3, 1, packs=[pk])
    # the same theta with that W_hh through the ordinary forward (its own prep launch packs)
    theta2 = theta.clone().to(DEV)
    off = 3 * H * (dims.E + dims.F)
    theta2[off:off + rows] = whh_next
    params = dec_params_from_oracle(p, theta2.cpu(), dims, DEV)
    ws_b = ops.decoder_workspace(dd, DEV)
    ws_b.zero_()
    ref_logits, _ = ops.decoder_forward(dd, params, feats, caps, ws_b)
    o = pk.wp - ws_a.data_ptr()
    n = 2 * pk.hrows * pk.pitch * 4
    assert torch.equal(ws_a[o:o + n], ws_b[o:o + n])
    # and the forward that trusts them (bit 128; bit 8: the prep cleared d Hs)
    got_logits, _ = ops.decoder_forward(dataclasses.replace(dd, pre=128), params, feats, caps, ws_a)
    assert torch.equal(got_logits, ref_logits)
    _, dl = ops.cross_entropy_fwd_bwd(ref_logits.clone(), caps, 0)
    ga = {n_: torch.full(s, float("nan"), device=DEV) for n_, s in dd.param_shapes().items()}
    gb = {n_: torch.full(s, float("nan"), device=DEV) for n_, s in dd.param_shapes().items()}
    lib.caphn_tune(13, dims.V)               # deterministic gradients: bit for bit
    try:
        ops.decoder_backward(dd, params, feats, caps, dl.clone(), ga, ws_a)
        ops.decoder_backward(dd, params, feats, caps, dl.clone(), gb, ws_b)
    finally:
        lib.caphn_tune(13, 0)
    for n_ in ga:
        assert torch.equal(ga[n_], gb[n_]), n_
    assert ops.device_error() == 0
