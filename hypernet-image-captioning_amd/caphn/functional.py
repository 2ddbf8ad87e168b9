"""torch.autograd.Function wrappers: the reference's module API on top of the C ABI.

Autograd only carries graph edges here; every forward/backward body is a libcaphn call.
"""
from typing import Dict, List

import torch

from . import ops
from ._lib import CaphnError

class _DecoderFn(torch.autograd.Function):
    """AttentionGru.forward / AttentionLstm.forward with sample_prob=0.0   models/decoderlstm.py:49-120, :224-261"""

    @staticmethod
    def forward(ctx, dims, features, captions, *tensors):
        if features.requires_grad:
            raise NotImplementedError("no gradient flows to the encoder's feature map: the reference freezes its encoder "
                                      "(models/encoder.py:11-13); detach the features")
        params = {n: t.detach().contiguous() for n, t in zip(dims.names(), tensors)}
        features = features.detach().contiguous()
        captions = captions.contiguous()
        ws = ops.decoder_workspace(dims, features.device)
        from . import config
        logits, alphas = ops.decoder_forward(dims, params, features, captions, ws, validate=config.VALIDATE_TOKENS)
        # saved through autograd: a parameter updated in place between forward and backward is detected (version check)
        ctx.save_for_backward(features, captions, *tensors)
        ctx.dims, ctx.ws = dims, ws
        return logits, alphas

    @staticmethod
    def backward(ctx, dlogits, dalphas):
        dims = ctx.dims
        features, captions, *tensors = ctx.saved_tensors
        params = {n: t.detach().contiguous() for n, t in zip(dims.names(), tensors)}
        dev = features.device
        if dlogits is None:
            dlogits = torch.zeros(dims.B, dims.T, dims.V, device=dev)
        grads = {n: torch.empty(s, dtype=torch.float32, device=dev) for n, s in dims.param_shapes().items()}
        ops.decoder_backward(dims, params, features, captions, dlogits.contiguous(), grads, ctx.ws,
                             dalphas.contiguous() if dalphas is not None else None)
        ctx.ws = None
        return (None, None, None) + tuple(grads[n] for n in dims.names())


class _DecoderSampledFn(torch.autograd.Function):
    """AttentionGru / AttentionLstm forward with scheduled sampling (models/decoderlstm.py:78-96, :236-251) as a trainable
    node: the argmax feedback is not differentiable, so the backward is the teacher-forced one over the token ids the
    forward actually fed (kept in the workspace)."""

    @staticmethod
    def forward(ctx, dims, flags, features, captions, *tensors):
        if features.requires_grad:
            raise NotImplementedError("no gradient flows to the encoder's feature map (frozen encoder, models/encoder.py:11-13)")
        params = {n: t.detach().contiguous() for n, t in zip(dims.names(), tensors)}
        features = features.detach().contiguous()
        captions = captions.contiguous()
        ws = ops.decoder_workspace(dims, features.device)
        logits, alphas = ops.decoder_forward_sampled(dims, params, features, captions, flags, ws, keep_state=True)
        ctx.save_for_backward(features, captions, *tensors)
        ctx.dims, ctx.ws = dims, ws
        return logits, alphas

    @staticmethod
    def backward(ctx, dlogits, dalphas):
        dims = ctx.dims
        features, captions, *tensors = ctx.saved_tensors
        params = {n: t.detach().contiguous() for n, t in zip(dims.names(), tensors)}
        dev = features.device
        if dlogits is None:
            dlogits = torch.zeros(dims.B, dims.T, dims.V, device=dev)
        grads = {n: torch.empty(s, dtype=torch.float32, device=dev) for n, s in dims.param_shapes().items()}
        ops.decoder_backward(dims, params, features, captions, dlogits.contiguous(), grads, ctx.ws,
                             dalphas.contiguous() if dalphas is not None else None)
        ctx.ws = None
        return (None, None, None, None) + tuple(grads[n] for n in dims.names())


def attention_sampled_forward(dims: ops.DecDims, flags, features: torch.Tensor, captions: torch.Tensor,
                              named: Dict[str, torch.Tensor]):
    if not features.is_cuda:
        raise CaphnError("libcaphn's HIP kernels need CUDA(HIP) tensors (there is no CPU fallback)")
    if bool((captions < 0).any()) or bool((captions >= dims.V).any()):
        raise IndexError("caption token id out of range")
    return _DecoderSampledFn.apply(dims, list(flags), features.float(), captions.long(), *[named[n] for n in dims.names()])


def attention_gru_forward(dims: ops.DecDims, features: torch.Tensor, captions: torch.Tensor,
                          named: Dict[str, torch.Tensor]):
    if not features.is_cuda:
        raise CaphnError("AttentionGru.forward runs on libcaphn's HIP kernels only: move the module and its "
                         "inputs to a CUDA(HIP) device (there is no CPU fallback)")
    return _DecoderFn.apply(dims, features.float(), captions.long(), *[named[n] for n in dims.names()])


class _BahdanauFn(torch.autograd.Function):
    """BahdanauAttention.forward   models/attention.py:21-46: every step a libcaphn call (two GEMMs + the fused score / softmax /
    context kernel forward; its gradient kernel, two fused weight-gradient GEMMs and two data-gradient GEMMs backward)."""

    @staticmethod
    def forward(ctx, features, hidden, wa_w, wa_b, ua_w, ua_b, va_w, va_b):
        f = features.detach().contiguous().float()
        h = hidden.detach().contiguous().float()
        B, P, Fd = f.shape
        waf = ops.gemm(f.view(B * P, Fd), wa_w.detach().contiguous(), False, True, bias=wa_b.detach().contiguous()).view(B, P, -1)
        uah = ops.gemm(h, ua_w.detach().contiguous(), False, True, bias=ua_b.detach().contiguous())
        v = va_w.detach().reshape(-1).contiguous()
        c, alpha = ops.bahdanau_fwd(f, waf, uah, v, va_b.detach().contiguous())
        ctx.save_for_backward(f, h, wa_w, ua_w, va_w, waf, uah, alpha)
        ctx.need = (features.requires_grad, hidden.requires_grad)
        return c, alpha

    @staticmethod
    def backward(ctx, dctx, dalpha):
        f, h, wa_w, ua_w, va_w, waf, uah, alpha = ctx.saved_tensors
        B, P, Fd = f.shape
        H = uah.shape[1]
        if dctx is None:
            dctx = torch.zeros(B, Fd, device=f.device)
        v = va_w.detach().reshape(-1).contiguous()
        dwaf, duah, part, df = ops.bahdanau_bwd(f, waf, uah, v, alpha, dctx.contiguous().float(),
                                                dalpha.contiguous().float() if dalpha is not None else None, ctx.need[0])
        dWa, dba = ops.linear_wgrad(dwaf.view(B * P, H), f.view(B * P, Fd))
        dUa, dbu = ops.linear_wgrad(duah, h)
        dv = ops.colsum(part)
        if df is not None:      # + the path through W_a
            ops.gemm(dwaf.view(B * P, H), wa_w.detach().contiguous(), out=df.view(B * P, Fd), accumulate=True)
        dh = ops.gemm(duah, ua_w.detach().contiguous()) if ctx.need[1] else None
        return df, dh, dWa, dba, dUa, dbu, dv[:H].view_as(va_w), dv[H:H + 1]


def bahdanau_attention(features, hidden, W_a, U_a, v_a):
    """W_a, U_a, v_a: the module's nn.Linear layers (v_a with one output)."""
    if not features.is_cuda:
        raise CaphnError("BahdanauAttention.forward runs on libcaphn's HIP kernels only (no CPU fallback)")
    return _BahdanauFn.apply(features, hidden, W_a.weight, W_a.bias, U_a.weight, U_a.bias, v_a.weight, v_a.bias)


class _HyperFn(torch.autograd.Function):
    """theta = cat_i head_i(hn_base(x))   hypernet_attention.py:111-118"""

    @staticmethod
    def forward(ctx, shape, x, factor_sink, *tensors):
        names = shape.param_names()
        params = {n: t.detach().contiguous() for n, t in zip(names, tensors)}
        theta, acts = ops.hyper_forward(shape, params, x.detach().float())
        ctx.shape, ctx.params, ctx.acts, ctx.xshape = shape, params, acts, x.shape
        ctx.need_x = x.requires_grad
        ctx.factor_sink = factor_sink
        return theta

    @staticmethod
    def backward(ctx, dtheta):
        shape = ctx.shape
        dev = dtheta.device
        dtheta = dtheta.contiguous()
        sink = ctx.factor_sink
        skip = set()
        if sink is not None:
            # rank-1 mode (caphn.optim.FusedAdam): dL/dW2_i = dtheta_i (x) a_i is never materialised -- the factors are handed
            # to the optimiser, which forms the product on the fly while it streams W, m, v once
            layout = ops.hyper_acts_layout(shape)
            o = 0
            for i, (k, w) in enumerate(shape.heads):
                ao, an = layout[f"a{i}"]
                # (a list: several backward passes before one optimiser step -- gradient accumulation -- make a rank-R member)
                sink.setdefault(f"hn_heads.{i}.2.weight", []).append((dtheta[o:o + w], ctx.acts[ao:ao + an]))
                skip.add(f"hn_heads.{i}.2.weight")
                o += w
        grads = {n: torch.empty(s, dtype=torch.float32, device=dev) for n, s in shape.param_shapes().items() if n not in skip}
        gx = ops.hyper_backward(shape, ctx.params, dtheta, ctx.acts, grads, want_x=ctx.need_x)
        return (None, gx.reshape(ctx.xshape) if gx is not None else None, None) + tuple(grads.get(n) for n in shape.param_names())


def hyper_forward(shape: ops.HyperShape, x: torch.Tensor, named: Dict[str, torch.Tensor], factor_sink=None) -> torch.Tensor:
    """factor_sink: a dict that receives {second-layer weight name: (dtheta_i, a_i)} in the backward INSTEAD of the dense
    [w_i, k_i] gradients (whose .grad then stays None); for optimisers that apply rank-1 gradients (caphn.optim.FusedAdam)."""
    if not x.is_cuda:
        raise CaphnError("HyperNet.forward runs on libcaphn's HIP kernels only (no CPU fallback)")
    return _HyperFn.apply(shape, x, factor_sink, *[named[n] for n in shape.param_names()])


class _PlainDecoderFn(torch.autograd.Function):
    """DecoderGRU / DecoderRNN forward with teacher forcing   later.py:394-447 / :254-317"""

    @staticmethod
    def forward(ctx, dims, opts, features, captions, h0, c0, *tensors):
        params = {n: t.detach().contiguous() for n, t in zip(dims.names(), tensors)}
        features = features.detach().contiguous()
        captions = captions.contiguous()
        ws = ops.plain_workspace(dims, features.device)
        if opts is not None and opts.get("seed") is not None:
            # teacher_forcing=False: words drawn per step; the workspace keeps their ids for the backward
            logits, opts["chosen"] = ops.plain_forward_sampled(dims, params, features, h0, c0, ws, opts["seed"])
        else:
            logits = ops.plain_forward(dims, params, features, captions, h0, c0, ws)
        ctx.dims, ctx.ws, ctx.params, ctx.features, ctx.captions, ctx.h0, ctx.c0 = dims, ws, params, features, captions, h0, c0
        ctx.need_f = True
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        dims = ctx.dims
        dev = ctx.features.device
        grads = {n: torch.empty(s, dtype=torch.float32, device=dev) for n, s in dims.param_shapes().items()}
        dfeat = torch.empty_like(ctx.features)
        ops.plain_backward(dims, ctx.params, ctx.features, ctx.captions, ctx.h0, ctx.c0, dlogits.contiguous(), grads, ctx.ws, dfeat)
        ctx.ws = None
        return (None, None, dfeat, None, None, None) + tuple(grads[n] for n in dims.names())


def plain_decoder_forward(dims: ops.PlainDims, features, captions, h0, c0, named: Dict[str, torch.Tensor], sample=None):
    """sample: {"seed": int} selects the teacher_forcing=False forward (one word drawn per caption and step from the previous
    step's softmax); the ids fed are returned in sample["chosen"] [B,T]."""
    if not features.is_cuda:
        raise CaphnError("DecoderGRU / DecoderRNN run on libcaphn's HIP kernels only: move the module and its inputs to a "
                         "CUDA(HIP) device (there is no CPU fallback)")
    return _PlainDecoderFn.apply(dims, sample, features.float(), captions.long(), h0, c0, *[named[n] for n in dims.names()])


class _MLPFn(torch.autograd.Function):
    """x -> Linear, ReLU, [Dropout,] ..., Linear over the last dimension (baseline/caption.py:34-47; with dropout the
    feed-forward block linear2(dropout(relu(linear1(x)))) of baseline/transformer.py:149,:213) as one node, so that the ReLU
    gradients are the mask epilogue of the input-gradient contractions (CAPHN_GEMM_MASK) and nothing runs outside libcaphn."""

    @staticmethod
    def forward(ctx, x, n_layers, drop_p, seed, *params):
        h = x.detach().reshape(-1, x.shape[-1]).contiguous()
        acts, ws = [h], []
        for i in range(n_layers):
            w, b = params[2 * i].detach().contiguous(), params[2 * i + 1].detach().contiguous()
            h = ops.gemm(h, w, tb=True, bias=b, relu=i < n_layers - 1)
            ws.append(w)
            if i < n_layers - 1:
                if drop_p > 0.0:
                    h = ops.dropout(h, drop_p, seed, offset=i << 40, out=h)      # zero where dropped OR clamped: still the ReLU mask
                acts.append(h)
        ctx.acts, ctx.ws, ctx.xshape, ctx.need_x, ctx.drop = acts, ws, x.shape, x.requires_grad, (drop_p, seed)
        return h.view(*x.shape[:-1], h.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        acts, ws = ctx.acts, ctx.ws
        drop_p, seed = ctx.drop
        n = len(ws)
        dz = dy.reshape(-1, ws[-1].shape[0]).contiguous()
        grads = [None] * (2 * n)
        dx = None
        for i in range(n - 1, -1, -1):
            a = acts[i]                                   # input of layer i (post-ReLU [post-dropout] output of layer i-1 for i > 0)
            grads[2 * i], grads[2 * i + 1] = ops.linear_wgrad(dz, a)
            if i > 0:
                dz = ops.gemm(dz, ws[i], mask=a)          # zero where the ReLU clamped (or the unit was dropped)
                if drop_p > 0.0:
                    dz = ops.dropout(dz, drop_p, seed, offset=(i - 1) << 40, out=dz)     # the kept units' 1 / (1 - p)
            elif ctx.need_x:
                dx = ops.gemm(dz, ws[0]).view(ctx.xshape)
        ctx.acts = ctx.ws = None
        return (dx, None, None, None) + tuple(grads)


def mlp(x, layers):
    """layers: sequence of nn.Linear; ReLU between them, none after the last."""
    if not x.is_cuda:
        raise CaphnError("caphn mlp runs on libcaphn's HIP kernels only (no CPU fallback)")
    params = []
    for l in layers:
        params += [l.weight, l.bias]
    return _MLPFn.apply(x.float(), len(layers), 0.0, 0, *params)


class _Conv1x1Fn(torch.autograd.Function):
    """nn.Conv2d(C_in, C_out, kernel_size=1) on NCHW input (baseline/caption.py:13-14, :27): one [bs*h*w, C_in] x
    [C_in, C_out] contraction over channels-last rows; the result is returned as an NCHW view of those rows."""

    @staticmethod
    def forward(ctx, src, w, b):
        bs, cin, h, wd = src.shape
        x = src.detach().permute(0, 2, 3, 1).reshape(-1, cin).contiguous()          # layout only
        wc = w.detach().contiguous()
        y = ops.gemm(x, wc, tb=True, bias=b.detach().contiguous())
        ctx.save_for_backward(x, wc)
        ctx.shape, ctx.need_x = (bs, cin, h, wd), src.requires_grad
        return y.view(bs, h, wd, -1).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        bs, cin, h, wd = ctx.shape
        dz = dy.permute(0, 2, 3, 1).reshape(-1, w.shape[0]).contiguous()
        dw, db = ops.linear_wgrad(dz, x)
        dx = None
        if ctx.need_x:
            dx = ops.gemm(dz, w).view(bs, h, wd, cin).permute(0, 3, 1, 2)
        return dx, dw, db


def conv1x1(src, w, b):
    if not src.is_cuda:
        raise CaphnError("caphn conv1x1 runs on libcaphn's HIP kernels only (no CPU fallback)")
    return _Conv1x1Fn.apply(src.float(), w, b)


def linear(x, w, b):
    """x W^T + b over the last dimension (one-layer case of mlp; w / b may be row slices of a packed projection)."""
    if not x.is_cuda:
        raise CaphnError("caphn linear runs on libcaphn's HIP kernels only (no CPU fallback)")
    return _MLPFn.apply(x.float(), 1, 0.0, 0, w, b)


def ffn(x, w1, b1, w2, b2, dropout_p: float = 0.0, seed=None):
    """linear2(dropout(relu(linear1(x))))   baseline/transformer.py:149,:213"""
    if not x.is_cuda:
        raise CaphnError("caphn ffn runs on libcaphn's HIP kernels only (no CPU fallback)")
    if dropout_p > 0.0 and seed is None:
        seed = next_seed()
    return _MLPFn.apply(x.float(), 2, float(dropout_p), int(seed or 0), w1, b1, w2, b2)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        xc, gc = x.detach().contiguous(), gamma.detach().contiguous()
        y, mean, rstd = ops.layernorm_fwd(xc, gc, beta.detach().contiguous(), eps)
        ctx.save_for_backward(xc, gc, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(x, gamma, mean, rstd, dy.contiguous())
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps: float = 1e-5):
    if not x.is_cuda:
        raise CaphnError("caphn layer_norm runs on libcaphn's HIP kernels only (no CPU fallback)")
    return _LayerNormFn.apply(x.float(), gamma, beta, eps)


_seed_state = {"n": 0}


def next_seed() -> int:
    """A fresh 64-bit dropout seed: torch's initial seed (so torch.manual_seed makes runs repeatable) + a call counter."""
    _seed_state["n"] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_state["n"] * 0xD1B54A32D192ED03) & (2 ** 64 - 1)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x.detach(), p, seed)

    @staticmethod
    def backward(ctx, g):
        return ops.dropout(g, ctx.p, ctx.seed), None, None


def dropout(x, p: float, training: bool = True, seed=None):
    """nn.Dropout: identity unless training and p > 0.  The mask comes from libcaphn's counter-based hash, not from torch's
    generator (statistically the same regulariser; element-wise different from the reference's masks)."""
    if not training or p <= 0.0:
        return x
    if not x.is_cuda:
        raise CaphnError("caphn dropout runs on libcaphn's HIP kernels only (no CPU fallback)")
    return _DropoutFn.apply(x.float(), float(p), next_seed() if seed is None else int(seed))


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, nh, attn_mask, key_padding, dropout_p, seed):
        q, k, v = q.detach(), k.detach(), v.detach()
        o, lse = ops.attention_fwd(q, k, v, nh, attn_mask, key_padding, dropout_p, seed)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.nh, ctx.attn_mask, ctx.key_padding, ctx.dropout_p, ctx.seed = nh, attn_mask, key_padding, dropout_p, seed
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = ops.attention_bwd(q, k, v, ctx.nh, ctx.attn_mask, ctx.key_padding, o, lse, d_o, ctx.dropout_p, ctx.seed)
        return dq, dk, dv, None, None, None, None, None


def attention(q, k, v, nh: int, attn_mask=None, key_padding_mask=None, dropout_p: float = 0.0, seed=None):
    """Multi-head softmax(q k^T / sqrt(dh) + masks) v on sequence-first [T, bs, d] tensors (the core of
    nn.MultiheadAttention as baseline/transformer.py:137,197-199 calls it).  attn_mask: additive float [tq, tk];
    key_padding_mask: bool [bs, tk], True = ignore; dropout_p: dropout on the probabilities (training)."""
    if not q.is_cuda:
        raise CaphnError("caphn attention runs on libcaphn's HIP kernels only (no CPU fallback)")
    am = attn_mask.to(device=q.device, dtype=torch.float32).contiguous() if attn_mask is not None else None
    kp = key_padding_mask.to(device=q.device, dtype=torch.uint8).contiguous() if key_padding_mask is not None else None
    if dropout_p > 0.0 and seed is None:
        seed = next_seed()
    return _AttentionFn.apply(q, k, v, nh, am, kp, float(dropout_p), int(seed or 0))


class _AddFn(torch.autograd.Function):
    """a + dropout(b, p) (same shapes) in one pass; p = 0 is the plain residual sum."""

    @staticmethod
    def forward(ctx, a, b, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.add_dropout(a.detach(), b.detach(), p, seed)

    @staticmethod
    def backward(ctx, g):
        return g, (ops.dropout(g, ctx.p, ctx.seed) if ctx.p > 0.0 else g), None, None


def add(a, b, dropout_p: float = 0.0, training: bool = True, seed=None):
    """a + b, or a + dropout(b) when dropout_p > 0 and training (baseline/transformer.py's  x + self.dropoutN(branch))."""
    if not a.is_cuda:
        raise CaphnError("caphn add runs on libcaphn's HIP kernels only (no CPU fallback)")
    if a.shape != b.shape:
        b = b.expand_as(a)
    p = float(dropout_p) if training else 0.0
    if p > 0.0 and seed is None:
        seed = next_seed()
    return _AddFn.apply(a.float(), b.float(), p, int(seed or 0))


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, table, padding_idx):
        flat = idx.reshape(-1).contiguous()
        out = ops.embedding_gather(table.detach().contiguous(), flat)
        ctx.save_for_backward(flat)
        ctx.shape, ctx.padding_idx = table.shape, padding_idx
        return out.view(*idx.shape, table.shape[1])

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        dt = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        ops.embedding_scatter_add(g.reshape(-1, ctx.shape[1]).contiguous(), flat, dt)
        if ctx.padding_idx is not None:
            dt[ctx.padding_idx].zero_()                  # nn.Embedding(padding_idx=...): that row receives no gradient
        return None, dt, None


def embedding(idx, table, padding_idx=None):
    if not table.is_cuda:
        raise CaphnError("caphn embedding runs on libcaphn's HIP kernels only (no CPU fallback)")
    return _EmbeddingFn.apply(idx.long(), table, padding_idx)


class _CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits.view(-1, V), targets.view(-1), ignore_index=...) (mean over the counted targets): loss and
    d logits in one pass over the logits (caphn_cross_entropy_fwd_bwd); the backward only applies the upstream scalar."""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        lg = logits.detach().contiguous()
        out, dlogits = ops.cross_entropy_fwd_bwd(lg.view(-1, lg.shape[-1]), targets.reshape(-1).contiguous(), ignore_index)
        ctx.dlogits, ctx.shape = dlogits, logits.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        d = ctx.dlogits
        ctx.dlogits = None
        return ops.scale_(d, g.detach().float()).view(ctx.shape), None, None


def cross_entropy(logits, targets, ignore_index: int = -100):
    if not logits.is_cuda:
        raise CaphnError("caphn cross_entropy runs on libcaphn's HIP kernels only (no CPU fallback)")
    return _CrossEntropyFn.apply(logits.float(), targets.long(), int(ignore_index))
