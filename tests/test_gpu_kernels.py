"""GPU parity tests of the individual C-ABI entry points against fp64/torch references and the
oracle.  Run with -m gpu on an MI355X.  Tolerances are fp32: products are exact fp32 (MFMA
f32 / VALU fma), only summation order differs from the CPU reference."""
import math

import numpy as np
import pytest
import torch

from oracle import caphn_oracle as O
from helpers import (TINY_DIMS, load_case, maxdiff, style_args, dec_params_from_oracle,
                     hyper_params_from_oracle, hyper_shape, dec_dims, rel_err)

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from caphn import ops as _ops
    return _ops


@pytest.fixture(params=[1, 0], ids=["split_bf16", "fp32_mfma"])
def gemm_backend(request):
    """Both GEMM back ends must meet the same fp32 tolerances (1 = default split-bf16, 0 = fp32 MFMA)."""
    from caphn import _lib
    lib = _lib.load()
    assert lib.caphn_tune(2, request.param) == 0
    yield request.param
    lib.caphn_tune(2, 1)


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (37, 83, 19), (130, 200, 200), (2560, 200, 333), (300, 1000, 64)])
def test_gemm_layouts(ops, gemm_backend, ta, tb, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    tol = 2e-6 * math.sqrt(K) * 4
    out = ops.gemm(A.to(DEV), B.to(DEV), ta, tb)
    assert maxdiff(out.cpu(), ref) < tol
    out = ops.gemm(A.to(DEV), B.to(DEV), ta, tb, bias=bias.to(DEV), relu=True)
    assert maxdiff(out.cpu(), torch.relu(ref + bias.double())) < tol
    # accumulate + mask
    C0 = torch.randn(M, N, generator=g)
    mask = torch.randn(M, N, generator=g)
    out = ops.gemm(A.to(DEV), B.to(DEV), ta, tb, out=C0.to(DEV).clone(), accumulate=True, mask=mask.to(DEV))
    assert maxdiff(out.cpu(), (ref + C0.double()) * (mask > 0)) < tol
    # split-K (atomic accumulate into zeros) with bias
    out = ops.gemm(A.to(DEV), B.to(DEV), ta, tb, bias=bias.to(DEV), splitk=3)
    assert maxdiff(out.cpu(), ref + bias.double()) < tol


@pytest.mark.parametrize("M,N", [(300, 1100), (1660, 9684), (129, 1024)])
def test_gemm_k_resident_variant(ops, M, N):
    """caphn_tune key 36 (off by default): the K = 200 NT products with A's fragments resident in registers (csrc/gemm_kres.hip) against
    fp64 and against the general kernel; plain, bias + ReLU, and through a live-row map (the vocabulary projection of the fused
    step: models/decoderlstm.py:105 over the rows whose target is not <pad>)."""
    import caphn._lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + N)
    A, B, bias = torch.randn(M, 200, generator=g).to(DEV), (torch.randn(N, 200, generator=g) * 0.07).to(DEV), torch.randn(N, generator=g).to(DEV)
    ref = A.double() @ B.double().t()
    try:
        lib.caphn_tune(36, 0)
        base = ops.gemm(A, B, False, True, bias=bias, relu=True)
        lib.caphn_tune(36, 1)
        out = ops.gemm(A, B, False, True)
        assert float((out.double() - ref).abs().max()) < 1e-6 * math.sqrt(200) * 4
        out2 = ops.gemm(A, B, False, True, bias=bias, relu=True)
        assert float((out2.double() - torch.relu(ref + bias.double())).abs().max()) < 1e-6 * math.sqrt(200) * 4
        assert maxdiff(out2.cpu(), base.cpu()) < 5e-6
    finally:
        lib.caphn_tune(36, 0)


def test_gemm_k_resident_variant_in_the_decoder_forward(ops):
    """The same variant behind caphn_decoder_forward's live-row vocabulary projection (row map from caphn_decoder_prepare_rows)."""
    import dataclasses
    import caphn._lib as L
    from helpers import dec_dims, dec_params_from_oracle
    lib = L.load()
    dims = O.Dims(D=32, F=200, E=200, H=200, V=1100, he=6)
    p = O.init_params(dims, seed=4)
    x = torch.zeros(dims.he); x[2] = 1.0
    theta = O.hyper_forward(p, x).detach()
    batch = O.synth_batch(dims, B=5, T=6, P=7, seed=5)
    dd = dataclasses.replace(dec_dims(dims, 5, 6, 7), rows=True)
    params = dec_params_from_oracle(p, theta, dims, DEV)
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    out = []
    try:
        for mode in (0, 1):
            lib.caphn_tune(36, mode)
            ws = ops.decoder_workspace(dd, DEV)
            ops.decoder_prepare_rows(dd, caps, 0, ws)
            logits = torch.zeros(5, 6, dims.V, device=DEV)
            ops.decoder_forward(dd, params, feats, caps, ws, logits=logits, want_alphas=False)
            out.append(logits.clone())
    finally:
        lib.caphn_tune(36, 0)
    live = (caps != 0)
    assert int(live.sum()) > 0 and maxdiff(out[0][live].cpu(), out[1][live].cpu()) < 2e-6
    assert torch.equal(out[0][~live], out[1][~live])            # dead rows: untouched by either


@pytest.mark.parametrize("K", [96, 200])
def test_gemm_large_tile_config(ops, gemm_backend, K):
    """>= 1024 128x128 tiles selects the 128x128 kernel (BK = 32 for K = 96, BK = 40 for K = 200)."""
    g = torch.Generator().manual_seed(K)
    A, B = torch.randn(4096, K, generator=g), torch.randn(4000, K, generator=g)
    bias = torch.randn(4000, generator=g)
    out = ops.gemm(A.to(DEV), B.to(DEV), False, True, bias=bias.to(DEV))
    ref = (A.to(DEV).double() @ B.to(DEV).double().t() + bias.to(DEV).double())
    assert float((out.double() - ref).abs().max()) < 1e-6 * math.sqrt(K) * 4


def test_gemm_split_bf16_is_fp32_accurate(ops):
    """The split-bf16 back end keeps fp32-class accuracy on data with a wide dynamic range
    (per-element magnitudes from 1e-9 to 1e3), where a plain bf16 GEMM would be off by 1e-2 relative."""
    from caphn import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    A = torch.randn(300, 777, generator=g) * torch.pow(10.0, torch.randint(-9, 4, (300, 777), generator=g).float())
    B = torch.randn(260, 777, generator=g) * torch.pow(10.0, torch.randint(-6, 2, (260, 777), generator=g).float())
    ref = A.double() @ B.double().t()
    scale = (A.double().abs() @ B.double().abs().t())          # sum_k |a||b|: the natural error scale
    errs = {}
    for mode in (0, 1):
        lib.caphn_tune(2, mode)
        out = ops.gemm(A.to(DEV), B.to(DEV), False, True)
        errs[mode] = float(((out.cpu().double() - ref).abs() / scale).max())
    lib.caphn_tune(2, 1)
    # fp32 accumulation over K = 777 widely scaled terms: ~1e-6 of sum |a||b| for BOTH back ends (a plain
    # bf16 GEMM would sit at ~4e-3); the split must not be worse than the fp32 MFMA
    assert errs[0] < 3e-6 and errs[1] < 1.25 * errs[0] + 1e-7, errs


def test_gemm_strided_views(ops):
    """leading dimensions larger than the logical width (W_ih[:, E:] style views)."""
    g = torch.Generator().manual_seed(5)
    W = torch.randn(48, 40, generator=g).to(DEV)      # [3H, E+F]
    X = torch.randn(21, 24, generator=g).to(DEV)
    out = ops.gemm(X, W[:, 16:], False, True)          # X @ W[:,16:]^T
    assert maxdiff(out.cpu(), X.cpu().double() @ W[:, 16:].cpu().double().t()) < 1e-5
    big = torch.zeros(48, 40, device=DEV)
    D = torch.randn(21, 48, generator=g).to(DEV)
    ops.gemm(D, X, True, False, out=big[:, 16:])       # dW[:,16:] = D^T X
    assert maxdiff(big[:, 16:].cpu(), D.cpu().double().t() @ X.cpu().double()) < 1e-5
    assert float(big[:, :16].abs().max()) == 0.0


def test_colsum(ops):
    g = torch.Generator().manual_seed(9)
    for M, N in [(5, 3), (2560, 600), (6272, 200), (100, 9684)]:
        A = torch.randn(M, N, generator=g)
        assert maxdiff(ops.colsum(A.to(DEV)).cpu(), A.double().sum(0)) < 1e-5 * math.sqrt(M) * 4


@pytest.fixture(params=[1, 0], ids=["tail-fused", "tail-5-launches"])
def hyper_tail(request):
    """caphn_tune key 27: the tail of caphn_hyper_backward as one launch with counter barriers / as five launches (default)."""
    import caphn._lib as L
    L.load().caphn_tune(27, request.param)
    yield request.param
    L.load().caphn_tune(27, 0)


@pytest.mark.parametrize("name", ["gru_tiny_flickr", "gru_tiny_cc", "gru_odd_cc"])
def test_hyper_forward_backward_tiny(ops, name, hyper_tail):
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    x, tok = style_args(g)
    xs = p["captioner.embed.weight"][tok] if tok is not None else x
    shape = hyper_shape(dims)
    hp = hyper_params_from_oracle(p, DEV)
    theta, acts = ops.hyper_forward(shape, hp, xs.to(DEV))
    assert maxdiff(theta.cpu(), g["theta"]) < 1e-6
    grads = {n: torch.empty(s, device=DEV) for n, s in shape.param_shapes().items()}
    gx = ops.hyper_backward(shape, hp, g["dtheta"].to(DEV), acts, grads, want_x=True)
    for n in grads:
        assert maxdiff(grads[n].cpu(), g["gint/" + n]) < 1e-6, n
    # dx against autograd on the oracle
    xr = xs.clone().reshape(-1).requires_grad_(True)
    O.hyper_forward(p, xr).backward(g["dtheta"])
    assert maxdiff(gx.cpu(), xr.grad) < 1e-6


def test_hyper_forward_backward_canonical_shape(ops, hyper_tail):
    """he=200 with the canonical head widths (480/240/200/200) but fewer rows: exercises the
    wave-per-row dwordx4 paths (k >= 128) incl. the 2-chunk 480 case and row tails."""
    torch.manual_seed(3)
    he, heads = 200, [(480, 4001), (240, 1999), (200, 600), (200, 600)]
    shape = ops.HyperShape(he, heads)
    p = {n: (torch.rand(s) - 0.5) * 0.2 for n, s in shape.param_shapes().items()}
    x = torch.randn(he)
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    theta_ref = O.hyper_forward(q, xr)
    dth = torch.randn(theta_ref.numel())
    theta_ref.backward(dth)
    pd = {k: v.to(DEV) for k, v in p.items()}
    theta, acts = ops.hyper_forward(shape, pd, x.to(DEV))
    assert maxdiff(theta.cpu(), theta_ref.detach()) < 5e-6
    grads = {n: torch.empty(s, device=DEV) for n, s in shape.param_shapes().items()}
    gx = ops.hyper_backward(shape, pd, dth.to(DEV), acts, grads, want_x=True)
    for n in grads:
        assert rel_err(grads[n], q[n].grad) < 2e-5, n
    assert rel_err(gx, xr.grad) < 2e-5


def test_hyper_small_layers_in_one_launch_bit_identical(ops):
    """caphn_tune key 28: hn_base and the heads' first layers as ONE launch (every workgroup recomputes the two 200 x 200 layers
    for itself) == the three-launch form bit for bit (same per-row arithmetic), theta included."""
    import caphn._lib as L
    torch.manual_seed(11)
    for he, heads in [(200, [(480, 4001), (240, 1999), (200, 600), (200, 600)]), (256, [(132, 77), (512, 300)])]:
        shape = ops.HyperShape(he, heads)
        pd = {n: ((torch.rand(s) - 0.5) * 0.3).to(DEV) for n, s in shape.param_shapes().items()}
        x = torch.randn(he).to(DEV)
        out = []
        try:
            for mode in (0, 1):
                L.load().caphn_tune(28, mode)
                theta, acts = ops.hyper_forward(shape, pd, x)
                out.append((theta.clone(), acts.clone()))
        finally:
            L.load().caphn_tune(28, 1)
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
        assert float(out[1][1].abs().sum()) > 0


def test_hyper_backward_fused_tail_hand_offs_under_load(ops):
    """The one-launch tail exchanges dz / dzb2 / dzb0 between workgroups through write-through stores and agent-scope loads behind
    counter barriers.  Alternating between two different d theta (a stale value of the other run would show), with GEMMs keeping
    the chip busy on a second stream (uneven load), every gradient word must equal the five-launch form's up to summation order,
    and repeated runs of the fused form must be bit-identical."""
    import caphn._lib as L
    torch.manual_seed(5)
    he, heads = 200, [(480, 4001), (240, 1999), (200, 600), (200, 600)]
    shape = ops.HyperShape(he, heads)
    pd = {n: ((torch.rand(s) - 0.5) * 0.2).to(DEV) for n, s in shape.param_shapes().items()}
    x = torch.randn(he).to(DEV)
    theta, acts = ops.hyper_forward(shape, pd, x)
    dths = [torch.randn(theta.numel()).to(DEV), (torch.randn(theta.numel()) * 3).to(DEV)]
    names = [n for n in shape.param_shapes() if not n.endswith(".2.weight")]

    def run(dth):
        grads = {n: torch.full(s, float("nan"), device=DEV) for n, s in shape.param_shapes().items() if not n.endswith(".2.weight")}
        gx = ops.hyper_backward(shape, pd, dth, acts, grads, want_x=True)
        return {**{n: grads[n].clone() for n in names if n in grads}, "x": gx.clone()}
    try:
        L.load().caphn_tune(27, 0)
        ref = [run(d) for d in dths]
        L.load().caphn_tune(27, 1)
        side = torch.cuda.Stream()
        A = torch.randn(4096, 4096, device=DEV)
        first = [None, None]
        for it in range(24):
            with torch.cuda.stream(side):
                for _ in range(3):
                    ops.gemm(A, A)
            got = run(dths[it & 1])
            torch.cuda.synchronize()
            for n, t in got.items():
                assert torch.isfinite(t).all(), n
                assert rel_err(t, ref[it & 1][n]) < 1e-5, (it, n)
            if first[it & 1] is None:
                first[it & 1] = got
            else:
                for n, t in got.items():
                    assert torch.equal(t, first[it & 1][n]), (it, n)
        assert ops.device_error() == 0
    finally:
        L.load().caphn_tune(27, 0)


@pytest.mark.parametrize("name", ["gru_tiny_flickr", "gru_tiny_cc", "gru_odd_cc"])
def test_decoder_forward_backward_tiny(ops, name):
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    B, T = g["captions"].shape
    P = g["features"].shape[1]
    dd = dec_dims(dims, B, T, P)
    params = dec_params_from_oracle(p, g["theta"], dims, DEV)
    ws = ops.decoder_workspace(dd, DEV)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    logits, alphas = ops.decoder_forward(dd, params, feats, caps, ws)
    assert maxdiff(logits.cpu(), g["logits"]) < 1e-6
    assert maxdiff(alphas.cpu(), g["alphas"]) < 1e-6
    lo, dlogits = ops.cross_entropy_fwd_bwd(logits, caps, 0)
    assert abs(float(lo[0]) - float(g["loss"])) < 1e-6
    grads = {n: torch.full(s, float("nan"), device=DEV) for n, s in dd.param_shapes().items()}
    ops.decoder_backward(dd, params, feats, caps, dlogits, grads, ws)
    for n, gt in grads.items():
        if n.startswith("gru."):
            continue
        assert maxdiff(gt.cpu(), g["glit/captioner." + n]) < 1e-6, n
    dth = torch.cat([grads["gru." + n].flatten() for n, _ in dims.cell_param_shapes()])
    assert maxdiff(dth.cpu(), g["dtheta"]) < 1e-6
    # quirk: caps[:,0] / caps[:,T-1] do not influence logits
    l2, _ = ops.decoder_forward(dd, params, feats, g["captions_q0"].to(DEV), ws)
    assert torch.equal(l2, logits)
    l3, _ = ops.decoder_forward(dd, params, feats, g["captions_q1"].to(DEV), ws)
    assert maxdiff(l3.cpu(), g["logits_q1"]) < 1e-6


@pytest.mark.parametrize("name", ["gru_tiny_flickr", "gru_tiny_cc", "gru_odd_cc", "lstm_tiny"])
def test_decoder_free_running_and_scheduled_sampling(ops, name):
    """sample_prob = 1.0 (every step t >= 1 feeds back its own argmax) and a mixed 0.5 pattern, against the
    reference's outputs; token argmax bit-exact."""
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    B, T = g["captions"].shape
    P = g["features"].shape[1]
    raw = dims.cell == "lstm"
    dd = dec_dims(dims, B, T, P, raw=raw)
    params = dec_params_from_oracle(p, g["theta"], dims, DEV)
    params = {n: params[n] for n in dd.names()}
    ws = ops.decoder_workspace(dd, DEV)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    lf, af = ops.decoder_forward_sampled(dd, params, feats, caps, [True] * T, ws)
    assert maxdiff(lf.cpu(), g["logits_free"]) < 1e-6 and maxdiff(af.cpu(), g["alphas_free"]) < 1e-6
    assert torch.equal(lf.argmax(-1).cpu(), g["tokens_free"])
    pattern = [bool(v) for v in g["mixed_pattern"]]
    lm, am = ops.decoder_forward_sampled(dd, params, feats, caps, pattern, ws)
    assert maxdiff(lm.cpu(), g["logits_mixed"]) < 1e-6 and maxdiff(am.cpu(), g["alphas_mixed"]) < 1e-6
    # all-False flags == teacher forcing
    l0, _ = ops.decoder_forward_sampled(dd, params, feats, caps, [False] * T, ws)
    assert maxdiff(l0.cpu(), g["logits"]) < 1e-6


@pytest.mark.parametrize("name", ["gru_tiny_cc", "gru_odd_cc"])
def test_decoder_row_subset_equals_full(ops, name):
    """dims.rows=True (vocab GEMMs restricted to rows whose target is not <pad>): identical loss and gradients;
    logits identical on the live rows (the others are never written)."""
    import dataclasses
    dims = TINY_DIMS[name]
    g, p = load_case(name)
    B, T = g["captions"].shape
    P = g["features"].shape[1]
    dd = dec_dims(dims, B, T, P)
    dr = dataclasses.replace(dd, rows=True)
    params = dec_params_from_oracle(p, g["theta"], dims, DEV)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV).clone()
    caps[0, -2:] = 0; caps[1, -1] = 0; caps[B - 1, 3:] = 0      # ragged captions: <pad> tails of different lengths
    assert int((caps == 0).sum()) > 0
    out = {}
    for tag, d_ in (("full", dd), ("rows", dr)):
        ws = ops.decoder_workspace(d_, DEV)
        logits = torch.full((B, T, dims.V), 7.0, device=DEV)
        if d_.rows:
            ops.decoder_prepare_rows(d_, caps, 0, ws)
        ops.decoder_forward(d_, params, feats, caps, ws, logits=logits)
        lo, dl = ops.cross_entropy_fwd_bwd(logits, caps, 0)
        grads = {n: torch.full(s, float("nan"), device=DEV) for n, s in d_.param_shapes().items()}
        ops.decoder_backward(d_, params, feats, caps, dl, grads, ws)
        out[tag] = (logits, float(lo[0]), grads)
    live = (caps != 0)
    assert torch.equal(out["rows"][0][live], out["full"][0][live])
    assert float((out["rows"][0][~live] - 7.0).abs().max()) == 0.0          # untouched
    assert out["rows"][1] == out["full"][1]
    for n in out["full"][2]:
        assert maxdiff(out["rows"][2][n].cpu(), out["full"][2][n].cpu()) < 2e-7, n
    # and both agree with the oracle on these ragged captions
    xs = p["captioner.embed.weight"][int(g["style_token"])] if "style_token" in g and int(g["style_token"]) >= 0 else g["x_style"]
    _, _, _, _, gref = O.forward_backward(dims, p, None if xs.dim() and "x_style" not in g else xs, g["features"], caps.cpu(),
                                          style_token=int(g["style_token"]) if int(g["style_token"]) >= 0 else None)
    for n in out["rows"][2]:
        if not n.startswith("gru."):
            assert maxdiff(out["rows"][2][n].cpu(), gref["captioner." + n]) < 1e-6, n


def test_decoder_lstm_raw_features_tiny(ops):
    """LSTM cell, attention over the raw features (no feature_fc): the reference's AttentionLstm
    (models/decoderlstm.py:188-261) with hypernet-injected weights, golden case lstm_tiny."""
    dims = TINY_DIMS["lstm_tiny"]
    g, p = load_case("lstm_tiny")
    B, T = g["captions"].shape
    P = g["features"].shape[1]
    dd = dec_dims(dims, B, T, P, raw=True)
    params = dec_params_from_oracle(p, g["theta"], dims, DEV)
    params = {n: params[n] for n in dd.names()}
    ws = ops.decoder_workspace(dd, DEV)
    feats, caps = g["features"].to(DEV), g["captions"].to(DEV)
    logits, alphas = ops.decoder_forward(dd, params, feats, caps, ws)
    assert maxdiff(logits.cpu(), g["logits"]) < 1e-6
    assert maxdiff(alphas.cpu(), g["alphas"]) < 1e-6
    lo, dlogits = ops.cross_entropy_fwd_bwd(logits, caps, 0)
    assert abs(float(lo[0]) - float(g["loss"])) < 1e-6
    grads = {n: torch.full(s, float("nan"), device=DEV) for n, s in dd.param_shapes().items()}
    ops.decoder_backward(dd, params, feats, caps, dlogits, grads, ws)
    for n, gt in grads.items():
        if n.startswith("lstm."):
            continue
        assert maxdiff(gt.cpu(), g["glit/captioner." + n]) < 1e-6, n
    dth = torch.cat([grads["lstm." + n].flatten() for n, _ in dims.cell_param_shapes()])
    assert maxdiff(dth.cpu(), g["dtheta"]) < 1e-6


@pytest.mark.parametrize("H,P", [(12, 6), (200, 49)])
def test_decoder_lstm_with_feature_fc_vs_oracle(ops, H, P):
    """LSTM cell behind a feature_fc (the hypernet-LSTM configuration).  At H=200, P=49 the four-gate G
    slab exceeds the LDS, so three gate slabs stay resident and the fourth streams from L2."""
    dims = O.Dims(D=24, F=H, E=H - 2 if H < 100 else H, H=H, V=50, he=6, cell="lstm")
    p = O.init_params(dims, seed=5)
    batch = O.synth_batch(dims, B=4, T=7, P=P, seed=6)
    x = torch.zeros(dims.he); x[2] = 1.0
    loss, logits_ref, alphas_ref, theta, gref = O.forward_backward(dims, p, x, batch["features"], batch["captions"])
    dd = dec_dims(dims, 4, 7, P)
    params = dec_params_from_oracle(p, theta, dims, DEV)
    ws = ops.decoder_workspace(dd, DEV)
    feats, caps = batch["features"].to(DEV), batch["captions"].to(DEV)
    logits, alphas = ops.decoder_forward(dd, params, feats, caps, ws)
    assert maxdiff(logits.cpu(), logits_ref) < 3e-6
    assert maxdiff(alphas.cpu(), alphas_ref) < 1e-6
    lo, dlogits = ops.cross_entropy_fwd_bwd(logits, caps, 0)
    grads = {n: torch.full(s, float("nan"), device=DEV) for n, s in dd.param_shapes().items()}
    ops.decoder_backward(dd, params, feats, caps, dlogits, grads, ws)
    for n, gt in grads.items():
        if n.startswith("lstm."):
            continue
        assert maxdiff(gt.cpu(), gref["captioner." + n]) < 3e-6, n
    dth = torch.cat([grads["lstm." + n].flatten() for n, _ in dims.cell_param_shapes()])
    assert maxdiff(dth.cpu(), gref["dtheta"]) < 3e-6


def test_cross_entropy_matches_torch(ops):
    torch.manual_seed(1)
    rows, V = 77, 9684
    logits = torch.randn(rows, V) * 3
    tgt = torch.randint(0, V, (rows,))
    tgt[::5] = 0
    lr = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, tgt, ignore_index=0)
    ref.backward()
    out, dl = ops.cross_entropy_fwd_bwd(logits.to(DEV), tgt.to(DEV), 0)
    assert abs(float(out[0]) - float(ref)) < 1e-6
    assert int(out[1]) == int((tgt != 0).sum())
    assert maxdiff(dl.cpu(), lr.grad) < 1e-7      # values ~1e-2: fp32 rounding
    # in place
    lg = logits.to(DEV)
    out2, dl2 = ops.cross_entropy_fwd_bwd(lg, tgt.to(DEV), 0, dlogits=lg)
    assert torch.equal(dl2, dl) and float(out2[0]) == float(out[0])


def test_embedding(ops):
    torch.manual_seed(2)
    table = torch.randn(50, 13)
    idx = torch.tensor([3, -1, 7, 3, 49, -1, 0])
    out = ops.embedding_gather(table.to(DEV), idx.to(DEV))
    ref = torch.where(idx[:, None] >= 0, table[idx.clamp(min=0)], torch.zeros(1))
    assert torch.equal(out.cpu(), ref)
    g = torch.randn(7, 13)
    tg = torch.zeros(50, 13, device=DEV)
    ops.embedding_scatter_add(g.to(DEV), idx.to(DEV), tg)
    r = torch.zeros(50, 13)
    for i, j in enumerate(idx.tolist()):
        if j >= 0:
            r[j] += g[i]
    assert maxdiff(tg.cpu(), r) < 1e-6


def test_adam_and_clip(ops):
    torch.manual_seed(4)
    n = 100003
    p, g = torch.randn(n), torch.randn(n) * 0.1
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd, gd = p.to(DEV), m.to(DEV), v.to(DEV), g.to(DEV)
    part = ops.sumsq_partials(gd)
    assert abs(float(part.sum()) - float((g.double() ** 2).sum())) < 1e-6 * float((g.double() ** 2).sum())
    coef = ops.clip_coef(part, None, 5.0, 1.0)
    tot, c = O.clip_coef([g], 5.0)
    assert abs(float(coef[1]) - tot) < 1e-5 * tot and abs(float(coef[0]) - c) < 1e-6
    for step in (1, 2, 3):
        O.adam_step(p, g * c, m, v, step, 1e-3)
        ops.adam_dense(pd, md, vd, gd, coef, 1e-3, step)
    assert maxdiff(pd.cpu(), p) < 1e-6 and maxdiff(md.cpu(), m) < 1e-7 and maxdiff(vd.cpu(), v) < 1e-8


@pytest.mark.parametrize("R,rows,k", [(1, 1000, 480), (2, 333, 240), (3, 57, 5), (1, 4001, 200)])
def test_adam_rank_equals_dense(ops, R, rows, k):
    torch.manual_seed(R * 100 + k)
    W = torch.randn(rows, k)
    gf, af = torch.randn(R, rows) * 0.1, torch.randn(R, k)
    dense = torch.einsum("rm,rk->mk", gf.double(), af.double())
    acc = torch.zeros(1, dtype=torch.float64, device=DEV)
    ops.rank_sumsq(gf.to(DEV), af.to(DEV), acc)
    assert abs(float(acc) - float((dense ** 2).sum())) < 1e-9 * float((dense ** 2).sum()) + 1e-12
    coef = torch.tensor([0.37, 0.0], device=DEV)
    Wd, md, vd = W.to(DEV), torch.zeros(rows, k, device=DEV), torch.zeros(rows, k, device=DEV)
    m, v = torch.zeros(rows, k), torch.zeros(rows, k)
    for step in (1, 2):
        O.adam_step(W, (dense * 0.37).float(), m, v, step, 1e-3)
        ops.adam_rank(Wd, md, vd, gf.to(DEV), af.to(DEV), coef, 1e-3, step)
    assert maxdiff(Wd.cpu(), W) < 1e-6
    assert maxdiff(ops.outer(gf[0].to(DEV), af[0].to(DEV)).cpu(), torch.outer(gf[0], af[0])) < 1e-7



def test_adam_rank_eight_ranks_matches_dense(ops):
    """Rank-8 gradient sum_r g_r (x) a_r (eight data-parallel ranks): factors staged in LDS, row factors hoisted."""
    g = torch.Generator().manual_seed(17)
    R, rows, k = 8, 301, 96
    a = torch.randn(R, k, generator=g).to(DEV)
    gfac = (torch.randn(R, rows, generator=g) * 1e-2).to(DEV)
    W = torch.randn(rows, k, generator=g).to(DEV)
    coef = torch.tensor([0.7, 0.0], device=DEV)
    Wc, m, v = W.clone(), torch.zeros_like(W), torch.zeros_like(W)
    ops.adam_rank(Wc, m, v, gfac, a, coef, 1e-3, 1)
    grad = 0.7 * (gfac.t().double().cpu() @ a.double().cpu())
    assert maxdiff(m.cpu(), 0.1 * grad) < 1e-7                          # m after one step = (1 - beta1) * grad
    assert maxdiff(v.cpu(), 0.001 * grad * grad) < 1e-9
    ref = W.double().cpu() - 1e-3 * grad / (grad.abs() + 1e-8)           # first Adam step: lr * g / (|g| + eps)
    assert maxdiff(Wc.cpu(), ref) < 1e-6


@pytest.mark.parametrize("R,rows,k,fused", [(1, 77, 1031, True), (1, 50, 2050, False), (3, 41, 2501, True), (1, 9, 8437, True)])
def test_adam_rank_long_odd_rows(ops, R, rows, k, fused):
    """Rows longer than 2048 or of a width that is no multiple of 4 (hypernet.py's heads: 11250, 8437): wave-per-row
    dword streams, with the next step's theta = W' a' + b from the same pass."""
    g = torch.Generator().manual_seed(k)
    a = torch.randn(R, k, generator=g)
    gfac = torch.randn(R, rows, generator=g) * 1e-2
    W = torch.randn(rows, k, generator=g)
    na, nb = torch.randn(k, generator=g), torch.randn(rows, generator=g)
    coef = torch.tensor([0.7, 0.0], device=DEV)
    Wd, md, vd = W.to(DEV), torch.zeros(rows, k, device=DEV), torch.zeros(rows, k, device=DEV)
    th = torch.zeros(rows, device=DEV)
    m, v = torch.zeros(rows, k), torch.zeros(rows, k)
    dense = (0.7 * (gfac.t().double() @ a.double())).float()
    for step in (1, 2):
        O.adam_step(W, dense, m, v, step, 1e-3)
        kw = dict(next_a=na.to(DEV), next_bias=nb.to(DEV), next_theta=th) if fused else {}
        ops.adam_rank(Wd, md, vd, gfac.to(DEV), a.to(DEV), coef, 1e-3, step, **kw)
    assert maxdiff(Wd.cpu(), W) < 1e-6 and maxdiff(md.cpu(), m) < 1e-7 and maxdiff(vd.cpu(), v) < 1e-9
    if fused:
        assert maxdiff(th.cpu(), (Wd.cpu().double() @ na.double() + nb.double()).float()) < 2e-5 * (k / 1000) ** 0.5 + 1e-5


@pytest.mark.parametrize("R,fused,shapes", [(1, True, [(600, 200), (600, 200)]), (2, False, [(600, 200), (37, 132), (64, 8)]),
                                             (1, True, [(600, 200), (90, 480)]), (1, False, [(50, 6), (9, 7)])])
def test_adam_rank_multi_equals_one_by_one(ops, R, fused, shapes):
    """Several rank-R members in one launch (the hypernet's two bias heads) == caphn_adam_rank(_gemv)_f32 member by member, bit for
    bit; members of different width classes / unaligned widths take the one-by-one path inside the call."""
    g = torch.Generator().manual_seed(len(shapes) * 10 + R)
    coef = torch.tensor([0.6, 0.0], device=DEV)
    one, many = [], []
    for rows, k in shapes:
        W = torch.randn(rows, k, generator=g).to(DEV)
        gf, af = (torch.randn(R, rows, generator=g) * 0.1).to(DEV), torch.randn(R, k, generator=g).to(DEV)
        na, nb = torch.randn(k, generator=g).to(DEV), torch.randn(rows, generator=g).to(DEV)
        for lst in (one, many):
            lst.append([W.clone(), torch.zeros(rows, k, device=DEV), torch.zeros(rows, k, device=DEV), gf, af] +
                       ([na, nb, torch.zeros(rows, device=DEV)] if fused and k % 4 == 0 else []))
    for step in (1, 2):
        for j in one:
            kw = dict(next_a=j[5], next_bias=j[6], next_theta=j[7]) if len(j) > 5 else {}
            ops.adam_rank(j[0], j[1], j[2], j[3], j[4], coef, 1e-3, step, **kw)
        ops.adam_rank_multi(many, coef, 1e-3, step)
    for a, b in zip(one, many):
        for x, y in zip(a[:3] + a[7:], b[:3] + b[7:]):
            assert torch.equal(x, y)


def test_cross_entropy_in_place_reads_target_logit_before_overwrite(ops):
    """d logits written over the logits (the engine's layout): the target logit of a row must be read before any wave of
    that row stores.  A delayed load showed up as a reported loss off by ~0.2 / n on some runs (gradients unaffected);
    full-size rows, several repetitions."""
    g = torch.Generator().manual_seed(23)
    rows, V = 2560, 9684
    logits = torch.randn(rows, V, generator=g)
    tgt = torch.randint(1, V, (rows,), generator=g)
    ref = float(torch.nn.functional.cross_entropy(logits.double(), tgt))
    for _ in range(5):
        lg = logits.to(DEV).clone()
        out, _ = ops.cross_entropy_fwd_bwd(lg, tgt.to(DEV), 0, dlogits=lg)
        assert abs(float(out[0]) - ref) < 2e-5


@pytest.mark.parametrize("R", [1, 3, 8])
def test_grad_norm_coef_one_launch_matches_dense_reference(R):
    """caphn_grad_norm_coef: clip_grad_norm_'s total norm over a dense arena plus rank-R gradients given by their factors
    (|| sum_r g_r (x) a_r ||_F^2 by the Gram identity), and the clip coefficient."""
    torch.manual_seed(R)
    n = 70001
    flat = torch.randn(n, device=DEV)
    th = 5000
    gfull = torch.randn(R, th + 7, device=DEV)[:, 3:3 + th]          # strided rows, unaligned start
    afull = torch.randn(R, 96, device=DEV)
    segs = [(gfull[:, 0:3000], afull[:, 0:40]), (gfull[:, 3000:5000], afull[:, 40:96])]
    from caphn import ops as cops
    gn = cops.GradNorm(n, R, len(segs), DEV)
    out = torch.zeros(2, device=DEV)
    dense = [torch.einsum("rm,rk->mk", g.double(), a.double()) for g, a in segs]
    tot = float((flat.double() ** 2).sum() + sum((d ** 2).sum() for d in dense)) ** 0.5
    for max_norm in (1e9, 0.5, 3.0):
        gn(flat, segs, max_norm, 1.0 / R, out)
        norm = tot / R
        want = (1.0 / R) * min(1.0, max_norm / (norm + 1e-6))
        assert abs(float(out[1]) - norm) < 1e-5 * norm
        assert abs(float(out[0]) - want) < 1e-6 * want
    # bit-identical from launch to launch (fixed summation order)
    a = out.clone(); gn(flat, segs, 3.0, 1.0 / R, out)
    assert torch.equal(a, out)


def test_split3_planes_are_exact():
    """hi + mid + lo == x bit for bit (8 + 8 + 8 significand bits), row tails and appended rows are zero."""
    from caphn import ops as cops
    torch.manual_seed(0)
    x = (torch.randn(37, 83, device=DEV) * torch.logspace(-20, 20, 83, device=DEV))[:, :77]        # strided view, wide exponent range
    x[3, 5] = 0.0; x[4, 6] = -0.0; x[5, 7] = 1e-41                                                 # zeros and a subnormal
    p = cops.Planes(x, zero_rows=3)
    s = p.plane(0).float() + p.plane(1).float() + p.plane(2).float()
    # exact for every normal number; a plane keeps the upper 16 bits of its residual, so fp32 subnormals whose bits all sit
    # in the lower half (|x| < 2^-133) come out as 0 -- as in the split-on-use kernel
    assert float(s[5, 7]) == 0.0
    s[5, 7] = x[5, 7]
    assert torch.equal(s, x)
    full = p.buf.view(3, -1, p.ldp)
    assert float(full[:, :37, 77:].float().abs().max()) == 0.0 and float(full[:, 37:40].float().abs().max()) == 0.0


@pytest.mark.parametrize("ta,tb,M,N,K", [(False, True, 300, 77, 200), (False, False, 129, 200, 96), (True, False, 200, 72, 1000),
                                          (False, True, 1664, 9684, 200), (True, False, 64, 264, 6272), (False, False, 70, 200, 9684)])
def test_gemm_on_presplit_operands_equals_split_on_use(ta, tb, M, N, K):
    """The pre-split path multiplies the same bf16 planes as the split-on-use kernel: identical products, same accumulation
    order inside a tile -> results equal to rounding of the (different) K-slab schedule at most; checked against fp64 too."""
    from caphn import ops as cops
    torch.manual_seed(1)
    A = torch.randn((K, M) if ta else (M, K), device=DEV)
    B = torch.randn((N, K) if tb else (K, N), device=DEV)
    bias = torch.randn(N, device=DEV)
    kp = (K + 7) & ~7
    # K % 8 != 0: a K-contiguous operand's row tails are zero-filled by the split, a K-slow operand gets zero rows appended
    pa = cops.Planes(A, zero_rows=(kp - K) if ta else 0)
    pb = cops.Planes(B, zero_rows=(kp - K) if not tb else 0)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double()) + bias.double()
    got = cops.gemm_planes(pa, pb, ta, tb, bias=bias, kp=kp if K % 8 else 0)
    base = cops.gemm(A, B, ta, tb, bias=bias)
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) < 3e-6 * scale
    assert float((got - base).abs().max()) < 1e-6 * scale
    if ta:                                   # split-K with atomics into a zeroed C
        got2 = cops.gemm_planes(pa, pb, ta, tb, splitk=4, kp=kp if K % 8 else 0)
        assert float((got2.double() - (ref - bias.double())).abs().max()) < 3e-6 * scale
