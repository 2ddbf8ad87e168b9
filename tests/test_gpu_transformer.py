"""N4: LayerNorm and the multi-head attention core (caphn_layernorm_*, caphn_attention_*) against fp64 torch math."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from caphn import ops as o
    return o


@pytest.mark.parametrize("shape,eps", [((300, 256), 1e-5), ((7, 3, 100), 1e-12), ((33, 700), 1e-5), ((5, 64), 1e-5)])
def test_layernorm_matches_torch(ops, shape, eps):
    g = torch.Generator().manual_seed(sum(shape))
    d = shape[-1]
    x = torch.randn(shape, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(d, generator=g), torch.randn(d, generator=g)
    dy = torch.randn(shape, generator=g)
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.layer_norm(x64, (d,), g64, b64, eps)
    ref.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), eps)
    assert maxdiff(y.cpu().double(), ref.detach()) < 3e-6
    dx, dg, db = ops.layernorm_bwd(x.to(DEV), gamma.to(DEV), mean, rstd, dy.to(DEV))
    assert maxdiff(dx.cpu().double(), x64.grad) < 5e-6
    rows = x.numel() // d
    assert maxdiff(dg.cpu().double(), g64.grad) < 2e-6 * max(1.0, math.sqrt(rows))
    assert maxdiff(db.cpu().double(), b64.grad) < 2e-6 * max(1.0, math.sqrt(rows))


def _ref_attention(q, k, v, nh, attn_mask, kpm):
    tq, bs, dm = q.shape
    tk, dh = k.shape[0], dm // nh
    qh = q.reshape(tq, bs, nh, dh).permute(1, 2, 0, 3)
    kh = k.reshape(tk, bs, nh, dh).permute(1, 2, 0, 3)
    vh = v.reshape(tk, bs, nh, dh).permute(1, 2, 0, 3)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if attn_mask is not None:
        s = s + attn_mask
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :].bool(), float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ vh).permute(2, 0, 1, 3).reshape(tq, bs, dm)


@pytest.mark.parametrize("bs,nh,dh,tq,tk,causal,pad,packed", [
    (3, 8, 32, 20, 49, False, True, False), (2, 8, 32, 33, 33, True, True, True), (2, 2, 64, 17, 130, False, False, False),
    (1, 4, 20, 70, 70, True, False, True), (4, 8, 32, 128, 128, True, True, False)])
def test_attention_core_matches_torch(ops, bs, nh, dh, tq, tk, causal, pad, packed):
    g = torch.Generator().manual_seed(bs * 1000 + tq)
    dm = nh * dh
    if packed:                                   # q, k, v as slices of one packed projection, as in_proj produces them
        assert tq == tk
        qkv = torch.randn(tq, bs, 3 * dm, generator=g).to(DEV)
        q, k, v = qkv[..., :dm], qkv[..., dm:2 * dm], qkv[..., 2 * dm:]
    else:
        q, k, v = (torch.randn(t, bs, dm, generator=g).to(DEV) for t in (tq, tk, tk))
    mask = None
    if causal:
        mask = torch.full((tq, tk), float("-inf")).triu(1)
    kpm = None
    if pad:
        kpm = torch.zeros(bs, tk, dtype=torch.bool)
        for b in range(bs):
            kpm[b, tk - 1 - 2 * b:] = True         # the first key is never padded: no fully masked row
    d_o = torch.randn(tq, bs, dm, generator=g)
    q64, k64, v64 = (t.detach().cpu().double().requires_grad_(True) for t in (q, k, v))
    ref = _ref_attention(q64, k64, v64, nh, mask.double() if mask is not None else None, kpm)
    ref.backward(d_o.double())
    md = mask.to(DEV) if mask is not None else None
    kd = kpm.to(torch.uint8).to(DEV) if kpm is not None else None
    o, lse = ops.attention_fwd(q, k, v, nh, md, kd)
    assert maxdiff(o.cpu().double(), ref.detach()) < 3e-6
    dq, dk, dv = ops.attention_bwd(q, k, v, nh, md, kd, o, lse, d_o.to(DEV))
    assert maxdiff(dq.cpu().double(), q64.grad) < 5e-6
    assert maxdiff(dk.cpu().double(), k64.grad) < 5e-6
    assert maxdiff(dv.cpu().double(), v64.grad) < 5e-6


def test_attention_limits_are_reported(ops):
    from caphn._lib import CaphnError
    q = torch.randn(4, 1, 8 * 32, device=DEV)
    k = torch.randn(600, 1, 8 * 32, device=DEV)
    with pytest.raises(CaphnError):
        ops.attention_fwd(q, k, k, 8)              # 600 keys of one head do not fit LDS
    with pytest.raises(CaphnError):
        ops.attention_fwd(q.cpu(), k.cpu(), k.cpu(), 8)


def test_dropout_mask_statistics_and_backward(ops):
    """caphn_dropout_f32: keep fraction, scale, determinism in (seed, offset), and backward = the same mask."""
    n, p = 1 << 20, 0.1
    x = torch.ones(n, device=DEV)
    y = ops.dropout(x, p, seed=1234)
    kept = (y > 0).float().mean().item()
    assert abs(kept - (1 - p)) < 3e-3                                   # 3 sigma of a binomial over 2^20 is 9e-4
    assert torch.allclose(y[y > 0], torch.full((1,), 1 / (1 - p), device=DEV))
    assert torch.equal(y, ops.dropout(x, p, seed=1234)) and not torch.equal(y, ops.dropout(x, p, seed=1235))
    assert torch.equal(ops.dropout(x, p, seed=1234, offset=7)[:-7], y[7:])          # the stream is indexed by offset + i
    # neighbouring elements are uncorrelated enough: the lag-1 product of the centred mask averages to ~0
    m = (y > 0).float() - (1 - p)
    assert abs(float((m[1:] * m[:-1]).mean())) < 1e-3
    from caphn import functional as CF
    xr = torch.randn(1000, device=DEV, requires_grad=True)
    out = CF.dropout(xr, 0.3, True, seed=99)
    out.backward(torch.ones_like(out))
    assert torch.equal(xr.grad, ops.dropout(torch.ones(1000, device=DEV), 0.3, 99))
    assert CF.dropout(xr, 0.3, False) is xr and CF.dropout(xr, 0.0, True) is xr


@pytest.mark.parametrize("tq,tk,causal", [(20, 49, False), (33, 33, True)])
def test_attention_probability_dropout_matches_torch_with_the_same_mask(ops, tq, tk, causal):
    """Dropout on the attention probabilities (nn.MultiheadAttention(dropout=p), training): the kernel's mask is
    dropout(ones[bs*nh, tq, tk], p, seed); with that mask handed to an fp64 torch restatement, output and all three
    gradients agree."""
    g = torch.Generator().manual_seed(tq)
    bs, nh, dh, p, seed = 2, 4, 32, 0.2, 4242
    dm = nh * dh
    q, k, v = (torch.randn(t, bs, dm, generator=g).to(DEV) for t in (tq, tk, tk))
    mask = torch.full((tq, tk), float("-inf")).triu(1) if causal else None
    M = ops.dropout(torch.ones(bs * nh * tq * tk, device=DEV), p, seed).view(bs, nh, tq, tk).cpu().double()
    assert 0.7 < float((M > 0).double().mean()) < 0.9
    q64, k64, v64 = (t.detach().cpu().double().requires_grad_(True) for t in (q, k, v))
    qh = q64.reshape(tq, bs, nh, dh).permute(1, 2, 0, 3)
    kh = k64.reshape(tk, bs, nh, dh).permute(1, 2, 0, 3)
    vh = v64.reshape(tk, bs, nh, dh).permute(1, 2, 0, 3)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if mask is not None:
        s = s + mask.double()
    ref = ((torch.softmax(s, -1) * M) @ vh).permute(2, 0, 1, 3).reshape(tq, bs, dm)
    d_o = torch.randn(tq, bs, dm, generator=g)
    ref.backward(d_o.double())
    md = mask.to(DEV) if mask is not None else None
    o, lse = ops.attention_fwd(q, k, v, nh, md, None, dropout_p=p, seed=seed)
    assert maxdiff(o.cpu().double(), ref.detach()) < 3e-6
    dq, dk, dv = ops.attention_bwd(q, k, v, nh, md, None, o, lse, d_o.to(DEV), dropout_p=p, seed=seed)
    assert maxdiff(dq.cpu().double(), q64.grad) < 5e-6
    assert maxdiff(dk.cpu().double(), k64.grad) < 5e-6
    assert maxdiff(dv.cpu().double(), v64.grad) < 5e-6


def test_ffn_with_inner_dropout_matches_torch_with_the_same_mask(ops):
    """linear2(dropout(relu(linear1(x)))): forward and every gradient against torch with the kernel's own mask."""
    from caphn import functional as CF
    g = torch.Generator().manual_seed(3)
    rows, d, ff, p, seed = 37, 24, 56, 0.25, 777
    x = torch.randn(rows, d, generator=g)
    w1, b1, w2, b2 = torch.randn(ff, d, generator=g) * 0.3, torch.randn(ff, generator=g), torch.randn(d, ff, generator=g) * 0.3, torch.randn(d, generator=g)
    dev = [t.to(DEV).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y = CF.ffn(dev[0], dev[1], dev[2], dev[3], dev[4], dropout_p=p, seed=seed)
    M = ops.dropout(torch.ones(rows * ff, device=DEV), p, seed).view(rows, ff).cpu().double()
    r64 = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    ref = F.linear(F.relu(F.linear(r64[0], r64[1], r64[2])) * M, r64[3], r64[4])
    assert maxdiff(y.detach().cpu().double(), ref.detach()) < 3e-6
    gy = torch.randn(rows, d, generator=g)
    y.backward(gy.to(DEV)); ref.backward(gy.double())
    for a, b in zip(dev, r64):
        assert maxdiff(a.grad.cpu().double(), b.grad) < 1e-5


def test_residual_add_with_branch_dropout(ops):
    """x + dropout(branch) in one kernel: equals x + caphn_dropout(branch) for the same seed; gradients: identity for x,
    the same mask for the branch; p = 0 is the plain sum."""
    from caphn import functional as CF
    g = torch.Generator().manual_seed(12)
    x = torch.randn(7, 5, 33, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(7, 5, 33, generator=g).to(DEV).requires_grad_(True)
    y = CF.add(x, b, 0.3, True, seed=55)
    assert maxdiff(y.detach().cpu(), (x.detach() + ops.dropout(b.detach(), 0.3, 55)).cpu()) < 1e-6      # fused multiply-add: 1 ulp
    gy = torch.randn(7, 5, 33, generator=g).to(DEV)
    y.backward(gy)
    assert torch.equal(x.grad, gy) and torch.equal(b.grad, ops.dropout(gy, 0.3, 55))
    assert torch.equal(CF.add(x, b).detach(), x.detach() + b.detach())
    assert torch.equal(CF.add(x, b, 0.3, False).detach(), x.detach() + b.detach())
    assert float((y.detach() == x.detach()).float().mean()) > 0.2                  # ~30 % of the branch dropped


def test_edge_shapes_and_fully_masked_rows(ops):
    """One query / one key, heads narrower than a float4, a single LayerNorm row of width 1, and a row whose keys are all
    masked (zeros here; torch yields NaN, documented in include/caphn.h)."""
    from caphn._lib import CaphnError
    g = torch.Generator().manual_seed(77)
    # 3 heads of 5
    q, k, v = (torch.randn(t, 2, 15, generator=g).to(DEV) for t in (1, 1, 1))
    o, lse = ops.attention_fwd(q, k, v, 3)
    assert maxdiff(o.cpu(), v.cpu()) < 1e-6                      # one key: softmax = 1, o = v
    q, k, v = (torch.randn(t, 2, 15, generator=g).to(DEV) for t in (4, 6, 6))
    ref = _ref_attention(q.cpu().double(), k.cpu().double(), v.cpu().double(), 3, None, None)
    o, lse = ops.attention_fwd(q, k, v, 3)
    assert maxdiff(o.cpu().double(), ref) < 3e-6
    # every key of sample 1 padded: zeros out, zero gradients, sample 0 untouched
    kpm = torch.zeros(2, 6, dtype=torch.uint8); kpm[1] = 1
    o2, lse2 = ops.attention_fwd(q, k, v, 3, None, kpm.to(DEV))
    assert torch.equal(o2[:, 1], torch.zeros_like(o2[:, 1])) and maxdiff(o2[:, 0].cpu(), o[:, 0].cpu()) < 1e-6
    dq, dk, dv = ops.attention_bwd(q, k, v, 3, None, kpm.to(DEV), o2, lse2, torch.ones_like(o2))
    for t in (dq, dk, dv):
        assert torch.isfinite(t).all() and float(t[:, 1].abs().max()) == 0.0
    # LayerNorm: one row, width 1 (variance 0 -> y = beta) and width 2
    y, m, r = ops.layernorm_fwd(torch.tensor([[3.0]], device=DEV), torch.tensor([2.0], device=DEV), torch.tensor([0.5], device=DEV), 1e-5)
    assert abs(float(y) - 0.5) < 1e-6
    x2 = torch.tensor([[1.0, 3.0]], device=DEV)
    y2, _, _ = ops.layernorm_fwd(x2, torch.ones(2, device=DEV), torch.zeros(2, device=DEV), 1e-5)
    assert maxdiff(y2.cpu(), torch.nn.functional.layer_norm(x2.cpu(), (2,))) < 1e-6
    with pytest.raises(CaphnError):
        ops.layernorm_fwd(torch.zeros(2, 2000, device=DEV), torch.ones(2000, device=DEV), torch.zeros(2000, device=DEV), 1e-5)   # d > 1024
    with pytest.raises(CaphnError):
        ops.dropout(torch.ones(4, device=DEV), 1.0, 1)                       # p must be < 1
    assert torch.equal(ops.dropout(torch.ones(4, device=DEV), 0.0, 1), torch.ones(4, device=DEV))


def test_embedding_node_padding_row_gets_no_gradient():
    from caphn import functional as CF
    table = torch.randn(9, 6, device=DEV, requires_grad=True)
    idx = torch.tensor([[0, 3, 3], [8, 0, 1]], device=DEV)
    out = CF.embedding(idx, table, padding_idx=0)
    assert torch.equal(out.detach(), table.detach()[idx])
    out.sum().backward()
    want = torch.zeros(9, 6); want[3] = 2.0; want[8] = 1.0; want[1] = 1.0
    assert torch.equal(table.grad.cpu(), want)
