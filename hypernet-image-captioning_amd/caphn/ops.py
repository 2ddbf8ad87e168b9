"""Tensor-level wrappers over the C ABI (include/caphn.h).  PyTorch is used for device memory
and streams only; every arithmetic op below runs in libcaphn's hand-written HIP kernels."""
import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L

GEMM_BIAS, GEMM_RELU, GEMM_ACCUM, GEMM_MASK, GEMM_LRELU = 1, 2, 4, 8, 16


def _f32(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


# ------------------------------------------------------------------ dense contraction
def gemm(a: torch.Tensor, b: torch.Tensor, ta: bool = False, tb: bool = False,
         bias: Optional[torch.Tensor] = None, relu: bool = False, out: Optional[torch.Tensor] = None,
         accumulate: bool = False, mask: Optional[torch.Tensor] = None, splitk: int = 1, lrelu: bool = False) -> torch.Tensor:
    """out[M,N] = epilogue(op(a) @ op(b)) on the fp32 MFMA pipe (caphn_gemm_f32).
    a, b are 2-D row-major tensors whose last stride is 1 (leading stride = ld)."""
    lib = L.load()
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[0], b.shape[1]) if tb else (b.shape[1], b.shape[0])
    if K != Kb:
        raise L.CaphnError(f"gemm inner dims differ: {K} vs {Kb}")
    if out is None:
        out = _f32(M, N, device=a.device)
        if splitk > 1:
            out.zero_()
    assert out.shape == (M, N) and out.stride(1) == 1
    flags = (GEMM_BIAS if bias is not None else 0) | (GEMM_RELU if relu else 0) | \
            (GEMM_ACCUM if accumulate else 0) | (GEMM_MASK if mask is not None else 0) | (GEMM_LRELU if lrelu else 0)
    for t in (a, b, out, bias, mask):
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise L.CaphnError("gemm needs fp32 CUDA tensors")
    rc = lib.caphn_gemm_f32(int(ta), int(tb), M, N, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                            out.data_ptr(), out.stride(0), bias.data_ptr() if bias is not None else None,
                            mask.data_ptr() if mask is not None else None,
                            mask.stride(0) if mask is not None else 0, flags, splitk, L.stream_ptr())
    L.check(rc, "caphn_gemm_f32")
    return out


class Planes:
    """The three bf16 planes (hi, mid, lo; x == hi + mid + lo exactly) of a 2-D fp32 matrix, split once for repeated use as a
    GEMM operand (caphn_split3_bf16).  `zero_rows`: rows of zeros appended (K rounded up to 8 for a K-slow operand)."""

    def __init__(self, x: torch.Tensor, zero_rows: int = 0):
        lib = L.load()
        assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
        rows, cols = x.shape
        self.src, self.rows, self.cols = x, rows, cols
        self.ldp = (cols + 7) & ~7
        self.ps = ((rows + zero_rows + 7) & ~7) * self.ldp
        self.buf = torch.empty(3 * self.ps, dtype=torch.bfloat16, device=x.device)
        L.check(lib.caphn_split3_bf16(C.c_void_p(x.data_ptr()), rows, cols, x.stride(0), C.c_void_p(self.buf.data_ptr()), self.ldp,
                                      self.ps, zero_rows, L.stream_ptr()), "caphn_split3_bf16")

    def plane(self, p: int) -> torch.Tensor:
        return self.buf[p * self.ps:(p + 1) * self.ps].view(-1, self.ldp)[:self.rows, :self.cols]


def gemm_planes(a: Planes, b: Planes, ta: bool = False, tb: bool = False, bias: Optional[torch.Tensor] = None,
                relu: bool = False, out: Optional[torch.Tensor] = None, splitk: int = 1, kp: int = 0) -> torch.Tensor:
    """ops.gemm on pre-split operands."""
    lib = L.load()
    A, B = a.src, b.src
    M, K = (A.shape[1], A.shape[0]) if ta else (A.shape[0], A.shape[1])
    N, Kb = (B.shape[0], B.shape[1]) if tb else (B.shape[1], B.shape[0])
    if K != Kb:
        raise L.CaphnError(f"gemm inner dims differ: {K} vs {Kb}")
    if out is None:
        out = torch.zeros(M, N, dtype=torch.float32, device=A.device) if splitk > 1 else _f32(M, N, device=A.device)
    flags = (GEMM_BIAS if bias is not None else 0) | (GEMM_RELU if relu else 0)
    L.check(lib.caphn_gemm_planes_f32(int(ta), int(tb), M, N, K, A.data_ptr(), A.stride(0), a.buf.data_ptr(), a.ldp, a.ps,
                                      B.data_ptr(), B.stride(0), b.buf.data_ptr(), b.ldp, b.ps, out.data_ptr(), out.stride(0),
                                      bias.data_ptr() if bias is not None else None, None, 0, flags, splitk, kp, L.stream_ptr()),
            "caphn_gemm_planes_f32")
    return out


def zero_(t: torch.Tensor) -> torch.Tensor:
    """Zero-fill a contiguous fp32 CUDA tensor with libcaphn's dwordx4 store kernel."""
    lib = L.load()
    L.check(lib.caphn_zero_f32(L.ptr(t), t.numel(), L.stream_ptr()), "caphn_zero_f32")
    return t


def add_dropout(x: torch.Tensor, branch: torch.Tensor, p: float = 0.0, seed: int = 0) -> torch.Tensor:
    """x + dropout(branch, p) in one pass (p = 0: x + branch)."""
    lib = L.load()
    x, branch = x.contiguous(), branch.contiguous()
    assert x.shape == branch.shape
    out = torch.empty_like(x)
    L.check(lib.caphn_add_dropout_f32(x.numel(), L.ptr(x), L.ptr(branch), float(p), int(seed) & (2 ** 64 - 1), 0, L.ptr(out),
                                      L.stream_ptr()), "caphn_add_dropout_f32")
    return out


def scale_(x: torch.Tensor, scale_dev: torch.Tensor) -> torch.Tensor:
    """x *= scale_dev[0] in place (scale on the device)."""
    lib = L.load()
    L.check(lib.caphn_scale_f32(x.numel(), L.ptr(x), L.ptr(scale_dev.reshape(-1)[:1].contiguous()), L.ptr(x), L.stream_ptr()),
            "caphn_scale_f32")
    return x


def lrelu_bwd(dy: torch.Tensor, post: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.LeakyReLU() backward from the layer's output: dy * (post > 0 ? 1 : 0.01)."""
    lib = L.load()
    assert dy.numel() == post.numel() and dy.is_contiguous() and post.is_contiguous()
    if out is None:
        out = torch.empty_like(dy)
    L.check(lib.caphn_lrelu_bwd_f32(dy.numel(), L.ptr(dy), L.ptr(post), L.ptr(out), L.stream_ptr()), "caphn_lrelu_bwd_f32")
    return out


def axpy_(y: torch.Tensor, x: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """y += alpha * x (contiguous fp32, same length)."""
    lib = L.load()
    assert x.numel() == y.numel() and x.is_contiguous() and y.is_contiguous()
    L.check(lib.caphn_axpy_f32(y.numel(), float(alpha), L.ptr(x), L.ptr(y), L.stream_ptr()), "caphn_axpy_f32")
    return y


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor):
    """(dW [M,N], db [M]) of y = x W^T + b from dy [K,M] and x [K,N] in one launch (bias gradient fused into the
    weight-gradient GEMM)."""
    lib = L.load()
    K, M = dy.shape
    N = x.shape[1]
    assert x.shape[0] == K and dy.stride(1) == 1 and x.stride(1) == 1
    dw, db = _f32(M, N, device=dy.device), _f32(M, device=dy.device)
    ws = torch.empty(lib.caphn_colsum_workspace_bytes(K, M), dtype=torch.uint8, device=dy.device)
    L.check(lib.caphn_linear_wgrad_f32(M, N, K, L.ptr(dy), dy.stride(0), L.ptr(x), x.stride(0), L.ptr(dw), N, L.ptr(db),
                                       C.c_void_p(ws.data_ptr()), L.stream_ptr()), "caphn_linear_wgrad_f32")
    return dw, db


def bahdanau_fwd(f: torch.Tensor, waf: torch.Tensor, uah: torch.Tensor, v_a: torch.Tensor, b_va: torch.Tensor):
    """(ctx [B,F], alpha [B,P]) from features f [B,P,F], waf [B,P,H] = W_a f + b, uah [B,H] = U_a h + b (models/attention.py:34-46)."""
    lib = L.load()
    B, P, Fd = f.shape
    H = uah.shape[1]
    ctx, alpha = _f32(B, Fd, device=f.device), _f32(B, P, device=f.device)
    L.check(lib.caphn_bahdanau_fwd(B, P, Fd, H, L.ptr(f), L.ptr(waf), L.ptr(uah), L.ptr(v_a), L.ptr(b_va), L.ptr(ctx), L.ptr(alpha),
                                   L.stream_ptr()), "caphn_bahdanau_fwd")
    return ctx, alpha


def bahdanau_bwd(f, waf, uah, v_a, alpha, dctx, dalpha, want_df: bool):
    """-> (dWaf [B,P,H], duah [B,H], part [B,H+1] (d v_a | d b_va partials per caption), df [B,P,F] or None)."""
    lib = L.load()
    B, P, Fd = f.shape
    H = uah.shape[1]
    dwaf, duah, part = _f32(B, P, H, device=f.device), _f32(B, H, device=f.device), _f32(B, H + 1, device=f.device)
    df = _f32(B, P, Fd, device=f.device) if want_df else None
    L.check(lib.caphn_bahdanau_bwd(B, P, Fd, H, L.ptr(f), L.ptr(waf), L.ptr(uah), L.ptr(v_a), L.ptr(alpha), L.ptr(dctx),
                                   L.ptr(dalpha, allow_none=True), L.ptr(dwaf), L.ptr(duah), L.ptr(part),
                                   L.ptr(df, allow_none=True), L.stream_ptr()), "caphn_bahdanau_bwd")
    return dwaf, duah, part, df


def colsum(a: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = L.load()
    M, N = a.shape
    ws = torch.empty(lib.caphn_colsum_workspace_bytes(M, N), dtype=torch.uint8, device=a.device)
    if out is None:
        out = _f32(N, device=a.device)
    L.check(lib.caphn_colsum_f32(M, N, L.ptr(a), a.stride(0), L.ptr(out), C.c_void_p(ws.data_ptr()), L.stream_ptr()),
            "caphn_colsum_f32")
    return out


# ------------------------------------------------------------------ hypernetwork
HYPER_NAMES = ("hn_base.0.weight", "hn_base.0.bias", "hn_base.2.weight", "hn_base.2.bias")


@dataclass
class HyperShape:
    """he and the (k_i, w_i) of every head (hypernet_attention.py:68-97).  d_in / d_mid (0 = he): input and
    hidden width of hn_base when they differ from he (hypernet.py:55-60: Linear(E,4E), Linear(4E,8E))."""
    he: int
    heads: List[Tuple[int, int]]
    d_in: int = 0
    d_mid: int = 0

    @property
    def din(self) -> int:
        return self.d_in or self.he

    @property
    def dmid(self) -> int:
        return self.d_mid or self.he

    @property
    def theta_size(self) -> int:
        return sum(w for _, w in self.heads)

    def param_names(self) -> List[str]:
        n = list(HYPER_NAMES)
        for i in range(len(self.heads)):
            n += [f"hn_heads.{i}.0.weight", f"hn_heads.{i}.0.bias", f"hn_heads.{i}.2.weight", f"hn_heads.{i}.2.bias"]
        return n

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        he = self.he
        s = {"hn_base.0.weight": (self.dmid, self.din), "hn_base.0.bias": (self.dmid,),
             "hn_base.2.weight": (he, self.dmid), "hn_base.2.bias": (he,)}
        for i, (k, w) in enumerate(self.heads):
            s[f"hn_heads.{i}.0.weight"] = (k, he); s[f"hn_heads.{i}.0.bias"] = (k,)
            s[f"hn_heads.{i}.2.weight"] = (w, k); s[f"hn_heads.{i}.2.bias"] = (w,)
        return s


def _hyper_desc(shape: HyperShape, p: Dict[str, torch.Tensor]) -> L.HyperDesc:
    if p.get("__frozen__") and "__desc__" in p:
        return p["__desc__"]
    d = L.HyperDesc()
    d.he = shape.he
    d.d_in, d.d_mid = shape.d_in, shape.d_mid
    if len(shape.heads) > L.MAX_HEADS:
        raise L.CaphnError(f"{len(shape.heads)} hypernet heads; libcaphn supports {L.MAX_HEADS}")
    d.n_heads = len(shape.heads)
    exp = shape.param_shapes()
    for n, s in exp.items():
        if tuple(p[n].shape) != tuple(s):
            raise L.CaphnError(f"{n}: expected shape {s}, got {tuple(p[n].shape)}")
        L.ptr(p[n])
    d.base_w0 = p["hn_base.0.weight"].data_ptr(); d.base_b0 = p["hn_base.0.bias"].data_ptr()
    d.base_w2 = p["hn_base.2.weight"].data_ptr(); d.base_b2 = p["hn_base.2.bias"].data_ptr()
    for i, (k, w) in enumerate(shape.heads):
        d.k[i] = k; d.w[i] = w
        d.w1[i] = p[f"hn_heads.{i}.0.weight"].data_ptr(); d.b1[i] = p[f"hn_heads.{i}.0.bias"].data_ptr()
        d.w2[i] = p[f"hn_heads.{i}.2.weight"].data_ptr(); d.b2[i] = p[f"hn_heads.{i}.2.bias"].data_ptr()
    if p.get("__frozen__"):
        p["__desc__"] = d
    return d


def hyper_acts_layout(shape: HyperShape) -> Dict[str, Tuple[int, int]]:
    """(offset, length) of x, a0, base and every a_i inside the acts buffer (segments padded to 4)."""
    up4 = lambda v: (v + 3) & ~3
    out, o = {}, 0
    for name, n in [("x", shape.din), ("a0", shape.dmid), ("base", shape.he)] + \
                   [(f"a{i}", k) for i, (k, _) in enumerate(shape.heads)]:
        out[name] = (o, n)
        o += up4(n)
    out["_total"] = (0, o)
    return out


def hyper_forward(shape: HyperShape, p: Dict[str, torch.Tensor], x: torch.Tensor,
                  theta: Optional[torch.Tensor] = None, acts: Optional[torch.Tensor] = None):
    """theta = cat_i head_i(hn_base(x))   (hypernet_attention.py:111-118).  x: [he] or [1,he]."""
    lib = L.load()
    d = _hyper_desc(shape, p)
    x = x.reshape(-1)
    if x.numel() != shape.din:
        raise L.CaphnError(f"hypernet input has {x.numel()} elements, expected {shape.din}")
    dev = x.device
    n_acts = lib.caphn_hyper_acts_floats(C.byref(d))
    if theta is None:
        theta = _f32(shape.theta_size, device=dev)
    if acts is None:
        acts = torch.zeros(n_acts, dtype=torch.float32, device=dev)
    assert acts.numel() >= n_acts and theta.numel() == shape.theta_size
    L.check(lib.caphn_hyper_forward(C.byref(d), L.ptr(x.contiguous()), L.ptr(theta), L.ptr(acts), L.stream_ptr()),
            "caphn_hyper_forward")
    return theta, acts


def hyper_backward(shape: HyperShape, p: Dict[str, torch.Tensor], dtheta: torch.Tensor, acts: torch.Tensor,
                   grads: Dict[str, torch.Tensor], want_x: bool = False,
                   ws: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """VJP of hyper_forward.  `grads` maps parameter names to preallocated output tensors; missing
    names are skipped (e.g. the dense 576 MB second-layer weight gradients)."""
    lib = L.load()
    d = _hyper_desc(shape, p)
    g = L.HyperGrads()
    gp = lambda n: (L.ptr(grads[n]).value if n in grads and grads[n] is not None else None)
    g.g_base_w0 = gp("hn_base.0.weight"); g.g_base_b0 = gp("hn_base.0.bias")
    g.g_base_w2 = gp("hn_base.2.weight"); g.g_base_b2 = gp("hn_base.2.bias")
    for i in range(len(shape.heads)):
        g.g_w1[i] = gp(f"hn_heads.{i}.0.weight"); g.g_b1[i] = gp(f"hn_heads.{i}.0.bias")
        g.g_w2[i] = gp(f"hn_heads.{i}.2.weight"); g.g_b2[i] = gp(f"hn_heads.{i}.2.bias")
    gx = _f32(shape.din, device=dtheta.device) if want_x else None
    g.g_x = gx.data_ptr() if gx is not None else None
    nbytes = lib.caphn_hyper_backward_workspace_bytes(C.byref(d))
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dtheta.device)
    L.check(lib.caphn_hyper_backward(C.byref(d), L.ptr(dtheta), L.ptr(acts), C.byref(g),
                                     C.c_void_p(ws.data_ptr()), L.stream_ptr()), "caphn_hyper_backward")
    return gx


# ------------------------------------------------------------------ decoder
@dataclass(frozen=True)
class DecDims:
    B: int
    T: int
    P: int
    D: int
    F: int
    E: int
    H: int
    V: int
    cell: str = "gru"        # "gru" (AttentionGru) or "lstm" (AttentionLstm)
    raw: bool = False        # True: no feature_fc, attention over the raw D-channel features (F == D)
    rows: bool = False       # True: vocab GEMMs only touch rows with a live target (decoder_prepare_rows first)
    pre: int = 0             # bit mask (see caphn_decoder_dims.precomputed): 1 precompute, 2 G, 4 inputs
    gz: bool = False         # True: the caller zero-filled every gradient output (one zero_ over its arena)
    layers: int = 1          # AttentionGru(num_layers): extra GRUCells h = layer(h, h) per time step (GRU only)
    drop_p: float = 0.0      # dropout on h_t in training mode (caphn_decoder_dims.dropout_p); the backward needs the same seed
    seed: int = 0
    logits_ld: int = 0       # row pitch of the logits / d logits buffer (caphn_decoder_dims.logits_ld); 0 = V, contiguous [B,T,V]

    @property
    def NG(self) -> int:
        return 4 if self.cell == "lstm" else 3

    def c(self) -> L.DecoderDims:
        return L.DecoderDims(self.B, self.T, self.P, self.D, self.F, self.E, self.H, self.V,
                             1 if self.cell == "lstm" else 0, int(self.raw), int(self.rows), int(self.gz), int(self.pre),
                             int(self.layers), float(self.drop_p), int(self.seed) & (2 ** 64 - 1), int(self.logits_ld))

    def fields(self):
        """Ordered (C struct field, parameter name) pairs this configuration uses."""
        f = []
        if not self.raw:
            f += [("fc0_w", "feature_fc.0.weight"), ("fc0_b", "feature_fc.0.bias"),
                  ("fc2_w", "feature_fc.2.weight"), ("fc2_b", "feature_fc.2.bias")]
        f += [("embed_w", "embed.weight"), ("out_w", "fc.weight"), ("out_b", "fc.bias"),
              ("Wa_w", "attention.W_a.weight"), ("Wa_b", "attention.W_a.bias"),
              ("Ua_w", "attention.U_a.weight"), ("Ua_b", "attention.U_a.bias"),
              ("va_w", "attention.v_a.weight"), ("va_b", "attention.v_a.bias"),
              ("inith_w", "init_h.weight"), ("inith_b", "init_h.bias")]
        if self.cell == "lstm":
            f += [("initc_w", "init_c.weight"), ("initc_b", "init_c.bias")]
        c = self.cell
        f += [("w_ih", f"{c}.weight_ih"), ("w_hh", f"{c}.weight_hh"), ("b_ih", f"{c}.bias_ih"), ("b_hh", f"{c}.bias_hh")]
        for l in range(self.layers - 1):
            f += [(f"lw_ih{l}", f"layers.{l}.weight_ih"), (f"lw_hh{l}", f"layers.{l}.weight_hh"),
                  (f"lb_ih{l}", f"layers.{l}.bias_ih"), (f"lb_hh{l}", f"layers.{l}.bias_hh")]
        return f

    def names(self) -> List[str]:
        return [n for _, n in self.fields()]

    def cell_names(self) -> List[str]:
        return [f"{self.cell}.{n}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        D, F, E, H, V, G = self.D, self.F, self.E, self.H, self.V, self.NG * self.H
        c = self.cell
        full = {"feature_fc.0.weight": (F, D), "feature_fc.0.bias": (F,), "feature_fc.2.weight": (F, F),
                "feature_fc.2.bias": (F,), "embed.weight": (V, E), "fc.weight": (V, H), "fc.bias": (V,),
                "attention.W_a.weight": (H, F), "attention.W_a.bias": (H,), "attention.U_a.weight": (H, H),
                "attention.U_a.bias": (H,), "attention.v_a.weight": (1, H), "attention.v_a.bias": (1,),
                "init_h.weight": (H, F), "init_h.bias": (H,), "init_c.weight": (H, F), "init_c.bias": (H,),
                f"{c}.weight_ih": (G, E + F), f"{c}.weight_hh": (G, H), f"{c}.bias_ih": (G,), f"{c}.bias_hh": (G,)}
        for l in range(self.layers - 1):
            full.update({f"layers.{l}.weight_ih": (3 * H, H), f"layers.{l}.weight_hh": (3 * H, H),
                         f"layers.{l}.bias_ih": (3 * H,), f"layers.{l}.bias_hh": (3 * H,)})
        return {n: full[n] for n in self.names()}


# GRU configuration's names (kept for callers that only deal with AttentionGru)
DEC_FIELD_TO_NAME = dict(DecDims(1, 1, 1, 1, 1, 1, 1, 1).fields())


def _dec_struct(cls, dims: DecDims, t: Dict[str, torch.Tensor]):
    """ctypes view of a name -> tensor dict.  A caller that keeps the SAME dict object for the same storage (the fused
    engine does) may set t["__frozen__"] = True: the validated struct is then memoised inside the dict."""
    key = ("__struct__", cls.__name__, dims.cell, dims.raw, dims.layers)
    if t.get("__frozen__") and key in t:
        return t[key]
    s = cls()
    shapes = dims.param_shapes()
    for field, name in dims.fields():
        ten = t[name]
        if tuple(ten.shape) != shapes[name]:
            raise L.CaphnError(f"{name}: expected shape {shapes[name]}, got {tuple(ten.shape)}")
        setattr(s, field, L.ptr(ten).value)
    if t.get("__frozen__"):
        t[key] = s
    return s


def stream_copy(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst = src (contiguous fp32, same size) on libcaphn's streaming copy kernel; small tails go through the 4-element path."""
    lib = L.load()
    n = src.numel()
    assert dst.numel() == n
    if n % 4 == 0 and src.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0:
        L.check(lib.caphn_stream_copy_f32(n, L.ptr(src), L.ptr(dst), L.stream_ptr()), "caphn_stream_copy_f32")
    else:
        dst.copy_(src)
    return dst


def device_error(clear: bool = False) -> int:
    """The current device's sticky failure word (include/caphn.h caphn_device_error): 0 or CAPHN_ETIMEOUT (-4).  Kernels write it
    asynchronously: synchronise first if the question is about work already enqueued."""
    return int(L.load().caphn_device_error(1 if clear else 0))


def decoder_workspace(dims: DecDims, device) -> torch.Tensor:
    lib = L.load()
    cd = dims.c()
    n = lib.caphn_decoder_workspace_bytes(C.byref(cd))
    if n == 0:
        raise L.CaphnError("bad decoder dims")
    return torch.empty(n, dtype=torch.uint8, device=device)


_PRE_EPOCH: Dict[int, int] = {}      # device index -> number of caphn_decoder_precompute calls issued from this process


def precompute_epoch(device) -> int:
    """The library keeps the events of the LAST caphn_decoder_precompute per device; a forward that lets the composite wait on
    them (DecDims.pre bit 16) must know that its own call is still the last one."""
    return _PRE_EPOCH.get(torch.device(device).index or 0, 0)


def decoder_precompute(dims: DecDims, params: Dict[str, torch.Tensor], features: torch.Tensor, ws: torch.Tensor,
                       captions: Optional[torch.Tensor] = None) -> None:
    """feature_fc, init_hidden and W_a f of the forward for `features` (models/decoderlstm.py:61-63, attention.py:34),
    issued ahead of the forward on the current stream; follow with decoder_forward(dims with pre=1).  With captions
    (and the final generated cell weights in params) also G, the embedding lookup and the x-side gates (pre=7)."""
    lib = L.load()
    if tuple(features.shape) != (dims.B, dims.P, dims.D):
        raise L.CaphnError(f"features {tuple(features.shape)} do not match {dims}")
    if captions is not None and tuple(captions.shape) != (dims.B, dims.T):
        raise L.CaphnError(f"captions {tuple(captions.shape)} do not match {dims}")
    cd = dims.c()
    ps = _dec_struct(L.DecoderParams, dims, params)
    _PRE_EPOCH[features.device.index or 0] = _PRE_EPOCH.get(features.device.index or 0, 0) + 1
    L.check(lib.caphn_decoder_precompute(C.byref(cd), C.byref(ps), L.ptr(features),
                                         L.ptr(captions, torch.int64, allow_none=True), C.c_void_p(ws.data_ptr()),
                                         L.stream_ptr()), "caphn_decoder_precompute")


def decoder_pair_pack_desc(dims: DecDims, ws: torch.Tensor):
    """Where the pair recurrent kernels keep their packed copy of W_hh inside `ws` (caphn_decoder_pair_pack_desc): a _lib.PairPack,
    or None when these dims do not run the pair kernels."""
    lib = L.load()
    cd = dims.c()
    out = L.PairPack()
    rc = lib.caphn_decoder_pair_pack_desc(C.byref(cd), C.c_void_p(ws.data_ptr()), C.byref(out))
    return out if rc == 0 else None


def decoder_pair_prep(dims: DecDims, params: Dict[str, torch.Tensor], ws: torch.Tensor) -> None:
    """The pair recurrent kernels' prep launch ahead of the forward (exchange areas, d Hs accumulator, U_a rows of the packed weight
    copy; the W_hh rows come from adam_rank_multi(..., pack=...)): follow with decoder_forward(dims with pre | 128)."""
    lib = L.load()
    cd = dims.c()
    ps = _dec_struct(L.DecoderParams, dims, params)
    L.check(lib.caphn_decoder_pair_prep(C.byref(cd), C.byref(ps), C.c_void_p(ws.data_ptr()), L.stream_ptr()), "caphn_decoder_pair_prep")


def decoder_inputs(dims: DecDims, params: Dict[str, torch.Tensor], captions: torch.Tensor, ws: torch.Tensor) -> None:
    """Embedding lookup + x-side gate pre-activations of the forward (decoderlstm.py:62, :82-88, :100) ahead of the
    rest; follow with decoder_forward(dims with pre | 4)."""
    lib = L.load()
    if tuple(captions.shape) != (dims.B, dims.T):
        raise L.CaphnError(f"captions {tuple(captions.shape)} do not match {dims}")
    cd = dims.c()
    ps = _dec_struct(L.DecoderParams, dims, params)
    L.check(lib.caphn_decoder_inputs(C.byref(cd), C.byref(ps), L.ptr(captions, torch.int64), C.c_void_p(ws.data_ptr()),
                                     L.stream_ptr()), "caphn_decoder_inputs")


def decoder_lookup(dims: DecDims, params: Dict[str, torch.Tensor], captions: torch.Tensor, ws: torch.Tensor) -> None:
    """Only the embedding lookup of decoder_inputs (needs the captions and the table, not the generated weights): ahead of time,
    beside the optimiser; follow with decoder_inputs / decoder_forward on dims with pre | 64."""
    lib = L.load()
    if tuple(captions.shape) != (dims.B, dims.T):
        raise L.CaphnError(f"captions {tuple(captions.shape)} do not match {dims}")
    cd = dims.c()
    ps = _dec_struct(L.DecoderParams, dims, params)
    L.check(lib.caphn_decoder_lookup(C.byref(cd), C.byref(ps), L.ptr(captions, torch.int64), C.c_void_p(ws.data_ptr()),
                                     L.stream_ptr()), "caphn_decoder_lookup")


def decoder_forward(dims: DecDims, params: Dict[str, torch.Tensor], features: torch.Tensor, captions: torch.Tensor,
                    ws: torch.Tensor, logits: Optional[torch.Tensor] = None, alphas: Optional[torch.Tensor] = None,
                    validate: bool = True, want_alphas: bool = True):
    """AttentionGru.forward(features, captions, 0.0) (models/decoderlstm.py:49-120) or, with
    dims.cell == "lstm", AttentionLstm.forward(captions, features, 0.0) (:224-261).
    validate=True range-checks the token ids (one host sync); the fused engine checks its batches once."""
    lib = L.load()
    if tuple(features.shape) != (dims.B, dims.P, dims.D) or tuple(captions.shape) != (dims.B, dims.T):
        raise L.CaphnError(f"features {tuple(features.shape)} / captions {tuple(captions.shape)} do not match {dims}")
    if validate and (bool((captions < 0).any()) or bool((captions >= dims.V).any())):
        raise IndexError("caption token id out of range")       # torch's embedding raises IndexError too
    cd = dims.c()
    ps = _dec_struct(L.DecoderParams, dims, params)
    dev = features.device
    if logits is None:
        logits = _f32(dims.B, dims.T, dims.logits_ld or dims.V, device=dev)
    elif logits.numel() < dims.B * dims.T * (dims.logits_ld or dims.V):
        raise L.CaphnError(f"logits buffer of {logits.numel()} floats is too small for {dims}")
    if alphas is None and want_alphas:          # want_alphas=False: the attention maps stay in the workspace (no copy out)
        alphas = _f32(dims.B, dims.T, dims.P, device=dev)
    L.check(lib.caphn_decoder_forward(C.byref(cd), C.byref(ps), L.ptr(features), L.ptr(captions, torch.int64),
                                      L.ptr(logits), L.ptr(alphas, allow_none=True), C.c_void_p(ws.data_ptr()), L.stream_ptr()),
            "caphn_decoder_forward")
    if dims.logits_ld and dims.logits_ld != dims.V and logits.shape[-1] == dims.logits_ld:
        return logits[..., :dims.V], alphas          # (a view: rows are logits_ld floats apart)
    return logits, alphas


def decoder_prepare_rows(dims: DecDims, targets: torch.Tensor, ignore_index: int, ws: torch.Tensor) -> None:
    """Row map of the (b,t) positions whose target is not ignore_index, kept in the workspace (dims.rows=True)."""
    lib = L.load()
    cd = dims.c()
    L.check(lib.caphn_decoder_prepare_rows(C.byref(cd), L.ptr(targets.reshape(-1), torch.int64), ignore_index,
                                           C.c_void_p(ws.data_ptr()), L.stream_ptr()), "caphn_decoder_prepare_rows")


def decoder_forward_sampled(dims: DecDims, params: Dict[str, torch.Tensor], features: torch.Tensor,
                            captions: torch.Tensor, use_sampling: Sequence[bool], ws: torch.Tensor, keep_state: bool = False):
    """Free-running / scheduled-sampling forward (models/decoderlstm.py:78-96, :236-251).
    use_sampling[t] is the per-step draw ``np.random.random() < sample_prob`` (entry 0 ignored).
    keep_state=True keeps what decoder_backward needs in ws (training through scheduled sampling)."""
    lib = L.load()
    if tuple(features.shape) != (dims.B, dims.P, dims.D) or tuple(captions.shape) != (dims.B, dims.T):
        raise L.CaphnError(f"features {tuple(features.shape)} / captions {tuple(captions.shape)} do not match {dims}")
    if len(use_sampling) != dims.T:
        raise L.CaphnError("use_sampling needs one flag per timestep")
    if bool((captions < 0).any()) or bool((captions >= dims.V).any()):
        raise IndexError("caption token id out of range")
    cd = dims.c()
    ps = _dec_struct(L.DecoderParams, dims, params)
    dev = features.device
    logits = _f32(dims.B, dims.T, dims.V, device=dev)
    alphas = _f32(dims.B, dims.T, dims.P, device=dev)
    flags = bytes(1 if bool(x) else 0 for x in use_sampling)
    fn = lib.caphn_decoder_forward_sampled_train if keep_state else lib.caphn_decoder_forward_sampled
    L.check(fn(C.byref(cd), C.byref(ps), L.ptr(features), L.ptr(captions, torch.int64), flags, L.ptr(logits), L.ptr(alphas),
               C.c_void_p(ws.data_ptr()), L.stream_ptr()),
            "caphn_decoder_forward_sampled_train" if keep_state else "caphn_decoder_forward_sampled")
    return logits, alphas


def decoder_search(dims: DecDims, params: Dict[str, torch.Tensor], features: torch.Tensor, beam: int, max_steps: int,
                   end_token: int = 2, first_token: int = 0, greedy: bool = False, poll_every: int = 8,
                   want_alphas: bool = False):
    """Batched device-resident decoding.  greedy=False: beam search of HyperNet.test_step
    (hypernet_attention.py:251-306; beam=3, max_steps=51 there).  greedy=True: AttentionGru.greedy_search
    (models/decoderlstm.py:138-175; beam must be 1, max_steps = max_sentence).

    dims.B is the number of IMAGES (dims.T is ignored); features [B, P, D].  The host looks at the device only
    every `poll_every` steps (one 4-byte read) to stop once every image's beam has emptied.
    Returns (seqs [B, max_steps+1] int64 starting with first_token, lengths [B], scores [B], finished [B] bool,
    alphas [B, max_steps, P] or None)."""
    lib = L.load()
    n = dims.B
    if tuple(features.shape) != (n, dims.P, dims.D):
        raise L.CaphnError(f"features {tuple(features.shape)} do not match {dims}")
    if greedy and beam != 1:
        raise L.CaphnError("greedy search is the beam == 1 case")
    if want_alphas and beam != 1:
        raise L.CaphnError("attention maps are only tracked for beam == 1")
    rows = DecDims(n * beam, 1, dims.P, dims.D, dims.F, dims.E, dims.H, dims.V, dims.cell, dims.raw, False, layers=dims.layers)
    cd = rows.c()
    cfg = L.SearchCfg(n, beam, max_steps, 0 if greedy else 1, 1 if greedy else 0, first_token, end_token)
    ps = _dec_struct(L.DecoderParams, rows, params)
    dev = features.device
    nbytes = lib.caphn_decoder_search_workspace_bytes(C.byref(cd), C.byref(cfg))
    if nbytes == 0:
        raise L.CaphnError(f"unsupported search configuration (beam {beam}, {dims})")
    ws = decoder_workspace(rows, dev)
    sws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    seqs = torch.zeros(n, max_steps + 1, dtype=torch.int64, device=dev)
    lengths = torch.zeros(n, dtype=torch.int32, device=dev)
    scores = _f32(n, device=dev)
    finished = torch.zeros(n, dtype=torch.int32, device=dev)
    n_active = torch.zeros(1, dtype=torch.int32, device=dev)
    alphas = torch.zeros(n, max_steps, dims.P, dtype=torch.float32, device=dev) if want_alphas else None
    wsp, swp = C.c_void_p(ws.data_ptr()), C.c_void_p(sws.data_ptr())
    L.check(lib.caphn_decoder_search_begin(C.byref(cd), C.byref(ps), C.byref(cfg), L.ptr(features), wsp, swp,
                                           L.stream_ptr()), "caphn_decoder_search_begin")
    done = 0
    while done < max_steps:
        chunk = min(poll_every, max_steps - done)
        L.check(lib.caphn_decoder_search_steps(C.byref(cd), C.byref(ps), C.byref(cfg), done + 1, chunk,
                                               L.ptr(alphas, allow_none=True), wsp, swp, L.stream_ptr()),
                "caphn_decoder_search_steps")
        done += chunk
        L.check(lib.caphn_decoder_search_result(C.byref(cd), C.byref(cfg), done, wsp, swp, L.ptr(seqs, torch.int64),
                                                L.ptr(lengths, torch.int32), L.ptr(scores), L.ptr(finished, torch.int32),
                                                L.ptr(n_active, torch.int32), L.stream_ptr()), "caphn_decoder_search_result")
        if int(n_active.item()) == 0:
            break
    return seqs, lengths, scores, finished.bool(), alphas


def decoder_backward(dims: DecDims, params: Dict[str, torch.Tensor], features: torch.Tensor, captions: torch.Tensor,
                     dlogits: torch.Tensor, grads: Dict[str, torch.Tensor], ws: torch.Tensor,
                     dalphas: Optional[torch.Tensor] = None) -> None:
    """Backward of decoder_forward; `grads` holds one preallocated tensor per parameter name.
    dlogits is consumed (may be overwritten)."""
    lib = L.load()
    cd = dims.c()
    ps = _dec_struct(L.DecoderParams, dims, params)
    gs = _dec_struct(L.DecoderGrads, dims, grads)
    L.check(lib.caphn_decoder_backward(C.byref(cd), C.byref(ps), L.ptr(features), L.ptr(captions, torch.int64),
                                       L.ptr(dlogits), L.ptr(dalphas, allow_none=True), C.byref(gs),
                                       C.c_void_p(ws.data_ptr()), L.stream_ptr()), "caphn_decoder_backward")


# ------------------------------------------------------------------ loss
def cross_entropy_fwd_bwd(logits: torch.Tensor, targets: torch.Tensor, ignore_index: int = 0,
                          dlogits: Optional[torch.Tensor] = None, leave_ignored_rows: bool = False):
    """F.cross_entropy(logits.view(-1,V), targets.view(-1), ignore_index) and its gradient.
    Returns (loss_out[2] = {mean loss, n_valid}, dlogits).  dlogits may alias logits."""
    lib = L.load()
    V = logits.shape[-1]
    rows = logits.numel() // V
    if dlogits is None:
        dlogits = torch.empty_like(logits)
    ws = torch.empty(lib.caphn_ce_workspace_bytes(rows), dtype=torch.uint8, device=logits.device)
    out = _f32(2, device=logits.device)
    L.check(lib.caphn_cross_entropy_fwd_bwd(rows, V, L.ptr(logits), L.ptr(targets.reshape(-1), torch.int64),
                                            ignore_index, L.ptr(dlogits), L.ptr(out), int(leave_ignored_rows),
                                            C.c_void_p(ws.data_ptr()),
                                            L.stream_ptr()), "caphn_cross_entropy_fwd_bwd")
    return out, dlogits


def cross_entropy_rows(logits: torch.Tensor, targets: torch.Tensor, ignore_index: int, dlogits: torch.Tensor,
                       ws: torch.Tensor, leave_ignored_rows: bool = False, n_valid_ptr: Optional[int] = None, V: Optional[int] = None) -> None:
    """Stage 1 of the loss: d logits (may alias logits) and per-row losses into ws; cross_entropy_finish reduces them.
    V: the vocabulary size when the buffers' last dimension is a PADDED row pitch (DecDims.logits_ld) -- rows are then
    logits.shape[-1] floats apart and hold V logits each."""
    lib = L.load()
    ld = logits.shape[-1]
    rows = logits.numel() // ld
    V = ld if V is None else int(V)
    assert dlogits.shape[-1] == ld and V <= ld
    L.check(lib.caphn_cross_entropy_rows_ld(rows, V, ld, L.ptr(logits), L.ptr(targets.reshape(-1), torch.int64), ignore_index,
                                            L.ptr(dlogits), int(leave_ignored_rows),
                                            C.c_void_p(n_valid_ptr) if n_valid_ptr else None, C.c_void_p(ws.data_ptr()),
                                            L.stream_ptr()), "caphn_cross_entropy_rows_ld")


def cross_entropy_finish(rows: int, ws: torch.Tensor, out: torch.Tensor, n_valid_ptr: Optional[int] = None) -> None:
    lib = L.load()
    L.check(lib.caphn_cross_entropy_finish(rows, C.c_void_p(n_valid_ptr) if n_valid_ptr else None, L.ptr(out),
                                           C.c_void_p(ws.data_ptr()), L.stream_ptr()), "caphn_cross_entropy_finish")


def ce_workspace(rows: int, device) -> torch.Tensor:
    return torch.empty(L.load().caphn_ce_workspace_bytes(rows), dtype=torch.uint8, device=device)


def decoder_rowcount_ptr(dims: DecDims, ws: torch.Tensor) -> int:
    """Device address (int) of the live-row count decoder_prepare_rows leaves in the workspace."""
    cd = dims.c()
    return int(L.load().caphn_decoder_rowcount_ptr(C.byref(cd), C.c_void_p(ws.data_ptr())) or 0)


# ------------------------------------------------------------------ embedding
def embedding_gather(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    lib = L.load()
    rows, E = idx.numel(), table.shape[1]
    out = _f32(rows, E, device=table.device)
    L.check(lib.caphn_embedding_gather(rows, E, L.ptr(table), L.ptr(idx.reshape(-1), torch.int64), L.ptr(out),
                                       L.stream_ptr()), "caphn_embedding_gather")
    return out


def embedding_scatter_add(g: torch.Tensor, idx: torch.Tensor, table_grad: torch.Tensor) -> None:
    lib = L.load()
    rows, E = g.shape
    L.check(lib.caphn_embedding_scatter_add_v(rows, E, int(table_grad.shape[0]), L.ptr(g), L.ptr(idx.reshape(-1), torch.int64),
                                              L.ptr(table_grad), L.stream_ptr()), "caphn_embedding_scatter_add_v")


# ------------------------------------------------------------------ optimiser
def sumsq_partials(x: torch.Tensor, partial: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = L.load()
    nb = lib.caphn_sumsq_blocks(x.numel())
    if partial is None:
        partial = torch.empty(nb, dtype=torch.float64, device=x.device)
    L.check(lib.caphn_sumsq_f32(x.numel(), L.ptr(x), L.ptr(partial, torch.float64), L.stream_ptr()), "caphn_sumsq_f32")
    return partial[:nb]


def rank_sumsq(gfac: torch.Tensor, afac: torch.Tensor, acc: torch.Tensor) -> None:
    """acc[0] += || sum_r gfac[r] (x) afac[r] ||_F^2   (gfac [R,rows], afac [R,k], last stride 1)."""
    lib = L.load()
    R, rows = gfac.shape
    k = afac.shape[1]
    ws = torch.empty(2 * R * R, dtype=torch.float64, device=gfac.device)
    L.check(lib.caphn_rank_sumsq_f32(R, rows, k, gfac.data_ptr(), gfac.stride(0), afac.data_ptr(), afac.stride(0),
                                     L.ptr(acc, torch.float64), L.ptr(ws, torch.float64), L.stream_ptr()),
            "caphn_rank_sumsq_f32")


def rank_sumsq_multi(pairs, acc: torch.Tensor, ws: Optional[torch.Tensor] = None) -> None:
    """acc[0] += sum over (gfac [R,rows], afac [R,k]) pairs of || sum_r gfac[r] (x) afac[r] ||_F^2, one launch."""
    lib = L.load()
    n = len(pairs)
    R = pairs[0][0].shape[0]
    if ws is None:
        ws = torch.empty(2 * R * R * n, dtype=torch.float64, device=acc.device)
    rows = (C.c_int * n)(*[g.shape[1] for g, _ in pairs])
    ks = (C.c_int * n)(*[a.shape[1] for _, a in pairs])
    gp = (C.c_void_p * n)(*[g.data_ptr() for g, _ in pairs])
    ap = (C.c_void_p * n)(*[a.data_ptr() for _, a in pairs])
    ldg = (C.c_size_t * n)(*[g.stride(0) for g, _ in pairs])
    lda = (C.c_size_t * n)(*[a.stride(0) for _, a in pairs])
    L.check(lib.caphn_rank_sumsq_multi_f32(R, n, rows, ks, gp, ldg, ap, lda, L.ptr(acc, torch.float64),
                                           L.ptr(ws, torch.float64), L.stream_ptr()), "caphn_rank_sumsq_multi_f32")


class GradNorm:
    """clip_grad_norm_'s coefficient in one launch (caphn_grad_norm_coef): sum of squares of a flat gradient arena plus the
    norm of rank-R gradients given by their factors.  Holds the (zero-initialised) workspace and the argument arrays, which
    are rebuilt only when a pointer or a stride changes."""

    def __init__(self, n: int, R: int, njobs: int, device):
        lib = L.load()
        self.n, self.R, self.njobs = n, R, njobs
        self.ws = torch.zeros(lib.caphn_grad_norm_workspace_bytes(n, R, njobs), dtype=torch.uint8, device=device)
        self._key = None

    def __call__(self, flat_g: torch.Tensor, pairs, max_norm: float, scale: float, out: torch.Tensor) -> torch.Tensor:
        lib = L.load()
        n = len(pairs)
        assert flat_g.numel() == self.n and n == self.njobs and (n == 0 or pairs[0][0].shape[0] == self.R)
        key = (flat_g.data_ptr(),) + tuple((g.data_ptr(), g.stride(0), g.shape[1], a.data_ptr(), a.stride(0), a.shape[1]) for g, a in pairs)
        if key != self._key:
            self._args = ((C.c_int * n)(*[g.shape[1] for g, _ in pairs]), (C.c_int * n)(*[a.shape[1] for _, a in pairs]),
                          (C.c_void_p * n)(*[g.data_ptr() for g, _ in pairs]), (C.c_size_t * n)(*[g.stride(0) for g, _ in pairs]),
                          (C.c_void_p * n)(*[a.data_ptr() for _, a in pairs]), (C.c_size_t * n)(*[a.stride(0) for _, a in pairs]))
            for g, a in pairs:
                assert g.stride(1) == 1 and a.stride(1) == 1 and g.dtype == torch.float32 and a.dtype == torch.float32
            L.ptr(flat_g)
            self._key = key
        rows, ks, gp, ldg, ap, lda = self._args
        L.check(lib.caphn_grad_norm_coef(self.n, C.c_void_p(flat_g.data_ptr()), self.R, n, rows, ks, gp, ldg, ap, lda,
                                         float(max_norm), float(scale), L.ptr(out), C.c_void_p(self.ws.data_ptr()),
                                         L.stream_ptr()), "caphn_grad_norm_coef")
        return out


def grad_norm_multi(grads, pairs, max_norm: float, scale: float, out: torch.Tensor, ws: Optional[torch.Tensor] = None):
    """clip_grad_norm_'s coefficient over a LIST of dense gradient tensors plus rank-R members given as (gfac [R,rows],
    afac [R,k]) pairs (caphn_grad_norm_multi).  Returns (out, ws): out[0] = coefficient * scale, out[1] = the total norm."""
    lib = L.load()
    nt, nj = len(grads), len(pairs)
    R = pairs[0][0].shape[0] if nj else 1
    for g in grads:
        L.ptr(g)
    for g, a in pairs:
        assert g.stride(1) == 1 and a.stride(1) == 1 and g.dtype == torch.float32 and a.dtype == torch.float32 and g.shape[0] == R
    ns = (C.c_size_t * max(nt, 1))(*[g.numel() for g in grads])
    need = lib.caphn_grad_norm_multi_workspace_bytes(nt, ns, R, nj)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=out.device)
    gp = (C.c_void_p * max(nt, 1))(*[g.data_ptr() for g in grads])
    m = max(nj, 1)
    L.check(lib.caphn_grad_norm_multi(nt, gp, ns, R, nj, (C.c_int * m)(*[g.shape[1] for g, _ in pairs]),
                                      (C.c_int * m)(*[a.shape[1] for _, a in pairs]),
                                      (C.c_void_p * m)(*[g.data_ptr() for g, _ in pairs]), (C.c_size_t * m)(*[g.stride(0) for g, _ in pairs]),
                                      (C.c_void_p * m)(*[a.data_ptr() for _, a in pairs]), (C.c_size_t * m)(*[a.stride(0) for _, a in pairs]),
                                      float(max_norm), float(scale), L.ptr(out), C.c_void_p(ws.data_ptr()), L.stream_ptr()),
            "caphn_grad_norm_multi")
    return out, ws


def adam_multi(ps, ms, vs, gs, coef, lr, step, betas=(0.9, 0.999), eps=1e-8) -> None:
    """Adam on a list of (contiguous fp32) tensors in one launch per 48 tensors (caphn_adam_multi_f32)."""
    lib = L.load()
    n = len(ps)
    hp = _hp(lr, betas, eps, step)
    for p, m, v, g in zip(ps, ms, vs, gs):
        L.ptr(p); L.ptr(m); L.ptr(v); L.ptr(g)
        if not (m.numel() == p.numel() == v.numel() == g.numel()):
            raise L.CaphnError("adam_multi: parameter / state / gradient sizes differ")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    L.check(lib.caphn_adam_multi_f32(n, arr(ps), arr(ms), arr(vs), arr(gs), (C.c_size_t * n)(*[p.numel() for p in ps]),
                                     L.ptr(coef), C.byref(hp), L.stream_ptr()), "caphn_adam_multi_f32")


def grad_norm_adam_dense(gn: "GradNorm", p, m, v, flat_g: torch.Tensor, pairs, max_norm: float, scale: float, out: torch.Tensor,
                         lr, step, betas, eps, dev_scalars=None, ce=None) -> torch.Tensor:
    """gn(flat_g, pairs, ...) followed by adam_dense(p, m, v, flat_g, out, ...) in two launches (caphn_grad_norm_adam_dense): the
    clip coefficient is finished inside the Adam kernel.  ce = (rows, ce_ws, n_valid_ptr or None, loss_out): the launch also reduces the
    cross entropy's per-row losses (cross_entropy_rows) to loss_out -- cross_entropy_finish without a launch of its own."""
    lib = L.load()
    n = len(pairs)
    assert flat_g.numel() == gn.n and n == gn.njobs and (n == 0 or pairs[0][0].shape[0] == gn.R)
    key = (flat_g.data_ptr(),) + tuple((g.data_ptr(), g.stride(0), g.shape[1], a.data_ptr(), a.stride(0), a.shape[1]) for g, a in pairs)
    if key != gn._key:
        gn._args = ((C.c_int * n)(*[g.shape[1] for g, _ in pairs]), (C.c_int * n)(*[a.shape[1] for _, a in pairs]),
                    (C.c_void_p * n)(*[g.data_ptr() for g, _ in pairs]), (C.c_size_t * n)(*[g.stride(0) for g, _ in pairs]),
                    (C.c_void_p * n)(*[a.data_ptr() for _, a in pairs]), (C.c_size_t * n)(*[a.stride(0) for _, a in pairs]))
        for g, a in pairs:
            assert g.stride(1) == 1 and a.stride(1) == 1 and g.dtype == torch.float32 and a.dtype == torch.float32
        L.ptr(flat_g); L.ptr(p); L.ptr(m); L.ptr(v)
        gn._key = key
    rows, ks, gp, ldg, ap, lda = gn._args
    hp = _hp(lr, betas, eps, step, dev_scalars)
    ce_rows, ce_ws, ce_cnt, ce_out = (0, None, None, None) if ce is None else ce
    L.check(lib.caphn_grad_norm_adam_dense(gn.n, C.c_void_p(p.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()),
                                           C.c_void_p(flat_g.data_ptr()), gn.R, n, rows, ks, gp, ldg, ap, lda, float(max_norm),
                                           float(scale), L.ptr(out), C.c_void_p(gn.ws.data_ptr()), C.byref(hp), int(ce_rows),
                                           None if ce_ws is None else C.c_void_p(ce_ws.data_ptr()), ce_cnt,
                                           None if ce_out is None else C.c_void_p(ce_out.data_ptr()), L.stream_ptr()),
            "caphn_grad_norm_adam_dense")
    return out


def decoder_hyper_backward(dims: DecDims, params, features, captions, dlogits, grads, ws,
                           shape: HyperShape, hyper_params, acts, hyper_grads, hyper_ws, want_x: bool = False,
                           x_accum_into: Optional[torch.Tensor] = None):
    """decoder_backward + hyper_backward in one call; the hypernet VJP overlaps the decoder's tail.
    The cell gradients in `grads` must be consecutive views of one dtheta buffer (theta order)."""
    lib = L.load()
    cd = dims.c()
    ps = _dec_struct(L.DecoderParams, dims, params)
    gs = _dec_struct(L.DecoderGrads, dims, grads)
    hd = _hyper_desc(shape, hyper_params)
    if hyper_grads.get("__frozen__") and "__hgrads__" in hyper_grads:
        g = hyper_grads["__hgrads__"]
    else:
        g = L.HyperGrads()
        gp = lambda n: (L.ptr(hyper_grads[n]).value if n in hyper_grads and hyper_grads[n] is not None else None)
        g.g_base_w0 = gp("hn_base.0.weight"); g.g_base_b0 = gp("hn_base.0.bias")
        g.g_base_w2 = gp("hn_base.2.weight"); g.g_base_b2 = gp("hn_base.2.bias")
        for i in range(len(shape.heads)):
            g.g_w1[i] = gp(f"hn_heads.{i}.0.weight"); g.g_b1[i] = gp(f"hn_heads.{i}.0.bias")
            g.g_w2[i] = gp(f"hn_heads.{i}.2.weight"); g.g_b2[i] = gp(f"hn_heads.{i}.2.bias")
        if hyper_grads.get("__frozen__"):
            hyper_grads["__hgrads__"] = g
    # x_accum_into: the row's gradient is ADDED there by the VJP's last kernel (the style token's row of the embedding gradient:
    # no separate scatter-add); else want_x returns it as a fresh tensor
    if x_accum_into is not None:
        gx = None
        g.g_x = L.ptr(x_accum_into).value
        g.x_accumulate = 1
    else:
        gx = _f32(shape.he, device=dlogits.device) if want_x else None
        g.g_x = gx.data_ptr() if gx is not None else None
        g.x_accumulate = 0
    L.check(lib.caphn_decoder_hyper_backward(C.byref(cd), C.byref(ps), L.ptr(features), L.ptr(captions, torch.int64),
                                             L.ptr(dlogits), None, C.byref(gs), C.c_void_p(ws.data_ptr()),
                                             C.byref(hd), L.ptr(acts), C.byref(g), C.c_void_p(hyper_ws.data_ptr()),
                                             L.stream_ptr()), "caphn_decoder_hyper_backward")
    return gx


MS_DTHETA, MS_VOCAB, MS_EMBED, MS_HYPER = 0, 1, 2, 3


def backward_milestone_wait(which: int) -> None:
    """The CURRENT stream waits until the last decoder backward enqueued on this device has finished part `which`
    (include/caphn.h: CAPHN_MS_*), not for the rest of it."""
    L.check(L.load().caphn_decoder_backward_milestone(int(which), L.stream_ptr()), "caphn_decoder_backward_milestone")


def clip_coef(partial: torch.Tensor, extra: Optional[torch.Tensor], max_norm: float, scale: float,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = L.load()
    if out is None:
        out = _f32(2, device=partial.device)
    L.check(lib.caphn_clip_coef(partial.numel(), L.ptr(partial, torch.float64),
                                L.ptr(extra, torch.float64, allow_none=True), float(max_norm), float(scale),
                                L.ptr(out), L.stream_ptr()), "caphn_clip_coef")
    return out


def _hp(lr, betas, eps, step, dev_scalars=None, zero_gfac=False):
    return L.AdamHParams(float(lr), float(betas[0]), float(betas[1]), float(eps), int(step),
                         dev_scalars.data_ptr() if dev_scalars is not None else None, 1 if zero_gfac else 0)


def adam_scalars(lr, betas, step):
    """Host values of the two step-dependent Adam scalars {lr / (1 - b1^t), sqrt(1 - b2^t)}, computed exactly as the library
    computes them when they travel by value (make_adam in csrc/misc.hip: the betas of caphn_adam_hparams are C floats, the powers
    and the square root are taken in double) -- a captured step that reads them from device memory then updates bit-identically
    to an eagerly launched one."""
    import math
    b1, b2, lrf = float(C.c_float(betas[0]).value), float(C.c_float(betas[1]).value), float(C.c_float(lr).value)
    return float(lrf / (1.0 - math.pow(b1, float(step)))), float(math.sqrt(1.0 - math.pow(b2, float(step))))


def adam_dense(p, m, v, g, coef, lr, step, betas=(0.9, 0.999), eps=1e-8, dev_scalars=None) -> None:
    lib = L.load()
    hp = _hp(lr, betas, eps, step, dev_scalars)
    n = p.numel()
    assert m.numel() == n and v.numel() == n and g.numel() == n
    L.check(lib.caphn_adam_dense_f32(n, L.ptr(p), L.ptr(m), L.ptr(v), L.ptr(g), L.ptr(coef), C.byref(hp),
                                     L.stream_ptr()), "caphn_adam_dense_f32")


def adam_rank(W, m, v, gfac, afac, coef, lr, step, betas=(0.9, 0.999), eps=1e-8, dev_scalars=None,
              next_a=None, next_bias=None, next_theta=None, zero_gfac=False) -> None:
    """Adam on W [rows,k] with gradient coef * sum_r gfac[r,:,None] * afac[r,None,:] (never materialised).
    With next_a / next_bias / next_theta the pass also emits next_theta = W' next_a + next_bias (the next
    step's forward GEMV on the updated weights, at no extra HBM traffic).  zero_gfac (one factor only): the pass clears gfac."""
    lib = L.load()
    hp = _hp(lr, betas, eps, step, dev_scalars, zero_gfac)
    rows, k = W.shape
    R = gfac.shape[0]
    assert gfac.shape[1] == rows and afac.shape[1] == k and afac.shape[0] == R
    assert gfac.stride(1) == 1 and afac.stride(1) == 1
    if next_a is None:
        L.check(lib.caphn_adam_rank_f32(R, rows, k, L.ptr(W), L.ptr(m), L.ptr(v), gfac.data_ptr(), gfac.stride(0),
                                        afac.data_ptr(), afac.stride(0), L.ptr(coef), C.byref(hp), L.stream_ptr()),
                "caphn_adam_rank_f32")
    else:
        assert next_a.numel() == k and next_bias.numel() == rows and next_theta.numel() == rows
        L.check(lib.caphn_adam_rank_gemv_f32(R, rows, k, L.ptr(W), L.ptr(m), L.ptr(v), gfac.data_ptr(), gfac.stride(0),
                                             afac.data_ptr(), afac.stride(0), L.ptr(coef), C.byref(hp),
                                             next_a.data_ptr(), next_bias.data_ptr(), next_theta.data_ptr(),
                                             L.stream_ptr()), "caphn_adam_rank_gemv_f32")


def adam_rank_multi(members, coef, lr, step, betas=(0.9, 0.999), eps=1e-8, dev_scalars=None, zero_gfac=False, packs=None) -> None:
    """adam_rank over several members in one launch (the hypernet's small heads are launch-bound one by one).  members: tuples
    (W, m, v, gfac, afac) or (W, m, v, gfac, afac, next_a, next_bias, next_theta); all with the same number of factors R.
    packs: per member None or a _lib.PairPack (decoder_pair_pack_desc) -- next_theta is also stored in that packed W_hh copy."""
    lib = L.load()
    hp = _hp(lr, betas, eps, step, dev_scalars, zero_gfac)
    jobs = (L.RankJob * len(members))()
    R = members[0][3].shape[0]
    for j, mb in zip(jobs, members):
        W, m, v, gfac, afac = mb[:5]
        rows, k = W.shape
        assert gfac.shape == (R, rows) and afac.shape == (R, k) and gfac.stride(1) == 1 and afac.stride(1) == 1
        j.W, j.m, j.v = L.ptr(W), L.ptr(m), L.ptr(v)
        j.gfac, j.ldg, j.afac, j.lda = gfac.data_ptr(), gfac.stride(0), afac.data_ptr(), afac.stride(0)
        j.rows, j.k = rows, k
        if len(mb) > 5 and mb[5] is not None:
            na, nb, nt = mb[5:8]
            assert na.numel() == k and nb.numel() == rows and nt.numel() == rows
            j.next_a, j.next_bias, j.next_theta = na.data_ptr(), nb.data_ptr(), nt.data_ptr()
    if packs is not None:
        for j, pk in zip(jobs, packs):
            if pk is not None:
                j.next_pack, j.pack_H, j.pack_HA, j.pack_pitch, j.pack_hrows = pk.wp, pk.H, pk.HA, pk.pitch, pk.hrows
    L.check(lib.caphn_adam_rank_multi_f32(R, len(members), jobs, L.ptr(coef), C.byref(hp), L.stream_ptr()), "caphn_adam_rank_multi_f32")


def hyper_forward_acts(shape: HyperShape, p: Dict[str, torch.Tensor], x: torch.Tensor, acts: torch.Tensor) -> None:
    """hn_base and the heads' first layers only (fills acts; no pass over the big second layers)."""
    lib = L.load()
    d = _hyper_desc(shape, p)
    x = x.reshape(-1)
    L.check(lib.caphn_hyper_forward_acts(C.byref(d), L.ptr(x.contiguous()), L.ptr(acts), L.stream_ptr()),
            "caphn_hyper_forward_acts")


def outer(g: torch.Tensor, a: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = L.load()
    rows, k = g.numel(), a.numel()
    if out is None:
        out = _f32(rows, k, device=g.device)
    L.check(lib.caphn_outer_f32(rows, k, L.ptr(g), L.ptr(a), L.ptr(out), L.stream_ptr()), "caphn_outer_f32")
    return out


# ------------------------------------------------------------------ non-attention decoders (hypernet.py / later.py)
@dataclass(frozen=True)
class PlainDims:
    """DecoderGRU / DecoderRNN of later.py:362-457 / :227-330: B, T, embed, hidden, vocab, layers, cell."""
    B: int
    T: int
    E: int
    H: int
    V: int
    layers: int = 1
    cell: str = "gru"

    @property
    def NG(self) -> int:
        return 4 if self.cell == "lstm" else 3

    def c(self) -> L.PlainDims:
        return L.PlainDims(self.B, self.T, self.E, self.H, self.V, self.layers, 1 if self.cell == "lstm" else 0)

    def names(self) -> List[str]:
        """Reference state_dict names in the order named_parameters() yields them (hypernet.py:62-68 skips
        embed / fc_out when sizing heads)."""
        n = [f"lstm_cell.{q}" for q in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        for l in range(self.layers - 1):
            n += [f"layers.{l}.{q}" for q in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return n + ["fc_out.weight", "fc_out.bias", "embed.weight"]

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        G, E, H, V = self.NG * self.H, self.E, self.H, self.V
        s = {"lstm_cell.weight_ih": (G, E), "lstm_cell.weight_hh": (G, H), "lstm_cell.bias_ih": (G,), "lstm_cell.bias_hh": (G,)}
        for l in range(self.layers - 1):
            s.update({f"layers.{l}.weight_ih": (G, H), f"layers.{l}.weight_hh": (G, H), f"layers.{l}.bias_ih": (G,),
                      f"layers.{l}.bias_hh": (G,)})
        s.update({"fc_out.weight": (V, H), "fc_out.bias": (V,), "embed.weight": (V, E)})
        return s


def _plain_struct(cls, dims: PlainDims, t: Dict[str, torch.Tensor]):
    if dims.layers > L.MAX_LAYERS:
        raise L.CaphnError(f"{dims.layers} layers; libcaphn supports {L.MAX_LAYERS}")
    st = cls()
    for n, shp in dims.param_shapes().items():
        if tuple(t[n].shape) != tuple(shp):
            raise L.CaphnError(f"{n}: expected shape {shp}, got {tuple(t[n].shape)}")
        L.ptr(t[n])
    st.embed_w = t["embed.weight"].data_ptr(); st.out_w = t["fc_out.weight"].data_ptr(); st.out_b = t["fc_out.bias"].data_ptr()
    for l in range(dims.layers):
        pre = "lstm_cell." if l == 0 else f"layers.{l - 1}."
        st.w_ih[l] = t[pre + "weight_ih"].data_ptr(); st.w_hh[l] = t[pre + "weight_hh"].data_ptr()
        st.b_ih[l] = t[pre + "bias_ih"].data_ptr(); st.b_hh[l] = t[pre + "bias_hh"].data_ptr()
    return st


def plain_workspace(dims: PlainDims, device) -> torch.Tensor:
    lib = L.load()
    cd = dims.c()
    n = lib.caphn_plain_workspace_bytes(C.byref(cd))
    if n == 0:
        raise L.CaphnError(f"bad plain-decoder dims {dims}")
    return torch.empty(n, dtype=torch.uint8, device=device)


def _plain_check(dims: PlainDims, features, captions, h0, c0):
    if tuple(features.shape) != (dims.B, dims.E) or tuple(captions.shape) != (dims.B, dims.T):
        raise L.CaphnError(f"features {tuple(features.shape)} / captions {tuple(captions.shape)} do not match {dims}")
    if tuple(h0.shape) != (dims.B, dims.H) or (dims.cell == "lstm" and (c0 is None or tuple(c0.shape) != (dims.B, dims.H))):
        raise L.CaphnError("h0 / c0 must be [B, H]")


def plain_forward(dims: PlainDims, params: Dict[str, torch.Tensor], features: torch.Tensor, captions: torch.Tensor,
                  h0: torch.Tensor, c0: Optional[torch.Tensor], ws: torch.Tensor, check_ids: bool = True,
                  logits: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DecoderGRU / DecoderRNN forward with teacher forcing (later.py:394-447 / :254-317) -> logits [B,T,V]."""
    lib = L.load()
    _plain_check(dims, features, captions, h0, c0)
    if check_ids and (bool((captions < 0).any()) or bool((captions >= dims.V).any())):      # a host sync: trainers that
        raise IndexError("caption token id out of range")                                  # validated their loader skip it
    cd = dims.c()
    ps = _plain_struct(L.PlainParams, dims, params)
    if logits is None:
        logits = _f32(dims.B, dims.T, dims.V, device=features.device)
    L.check(lib.caphn_plain_forward(C.byref(cd), C.byref(ps), L.ptr(features), L.ptr(captions, torch.int64), L.ptr(h0),
                                    L.ptr(c0, allow_none=True), L.ptr(logits), C.c_void_p(ws.data_ptr()), L.stream_ptr()),
            "caphn_plain_forward")
    return logits


def plain_forward_sampled(dims: PlainDims, params: Dict[str, torch.Tensor], features: torch.Tensor, h0: torch.Tensor,
                          c0: Optional[torch.Tensor], ws: torch.Tensor, seed: int):
    """DecoderGRU / DecoderRNN forward with teacher_forcing=False (later.py:418-431 / :290-301): from step 1 on the input is the
    embedding of a word drawn from softmax(out_{t-1}).  -> (logits [B,T,V], chosen [B,T] int64: the id fed at step t, -1 at 0).
    The draws follow the kernel's counter-based hash of (seed, b, t), not torch.multinomial's Philox stream."""
    lib = L.load()
    cd = dims.c()
    ps = _plain_struct(L.PlainParams, dims, params)
    logits = _f32(dims.B, dims.T, dims.V, device=features.device)
    chosen = torch.empty(dims.B, dims.T, dtype=torch.int64, device=features.device)
    L.check(lib.caphn_plain_forward_sampled(C.byref(cd), C.byref(ps), L.ptr(features), L.ptr(h0), L.ptr(c0, allow_none=True),
                                            C.c_uint64(int(seed) & (2 ** 64 - 1)), L.ptr(logits), L.ptr(chosen, torch.int64),
                                            C.c_void_p(ws.data_ptr()), L.stream_ptr()), "caphn_plain_forward_sampled")
    return logits, chosen


def plain_backward(dims: PlainDims, params: Dict[str, torch.Tensor], features: torch.Tensor, captions: torch.Tensor,
                   h0: torch.Tensor, c0: Optional[torch.Tensor], dlogits: torch.Tensor, grads: Dict[str, torch.Tensor],
                   ws: torch.Tensor, dfeatures: Optional[torch.Tensor] = None) -> None:
    lib = L.load()
    _plain_check(dims, features, captions, h0, c0)
    cd = dims.c()
    ps = _plain_struct(L.PlainParams, dims, params)
    gs = _plain_struct(L.PlainGrads, dims, grads)
    gs.features = dfeatures.data_ptr() if dfeatures is not None else None
    L.check(lib.caphn_plain_backward(C.byref(cd), C.byref(ps), L.ptr(features), L.ptr(captions, torch.int64), L.ptr(h0),
                                     L.ptr(c0, allow_none=True), L.ptr(dlogits), C.byref(gs), C.c_void_p(ws.data_ptr()),
                                     L.stream_ptr()), "caphn_plain_backward")


# ------------------------------------------------------------------ N4: transformer pieces (baseline/transformer.py)
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    """nn.LayerNorm over the last dimension -> (y, mean[rows], rstd[rows])."""
    lib = L.load()
    d = x.shape[-1]
    x2 = x.reshape(-1, d)
    rows = x2.shape[0]
    y = torch.empty_like(x2)
    mean, rstd = _f32(rows, device=x.device), _f32(rows, device=x.device)
    L.check(lib.caphn_layernorm_fwd(rows, d, L.ptr(x2), L.ptr(gamma), L.ptr(beta), float(eps), L.ptr(y), L.ptr(mean), L.ptr(rstd),
                                    L.stream_ptr()), "caphn_layernorm_fwd")
    return y.view(x.shape), mean, rstd


def layernorm_bwd(x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, dy: torch.Tensor):
    lib = L.load()
    d = x.shape[-1]
    x2, dy2 = x.reshape(-1, d), dy.reshape(-1, d)
    rows = x2.shape[0]
    dx = torch.empty_like(x2)
    dg, db = _f32(d, device=x.device), _f32(d, device=x.device)
    ws = torch.empty(lib.caphn_layernorm_bwd_workspace_bytes(rows, d), dtype=torch.uint8, device=x.device)
    L.check(lib.caphn_layernorm_bwd(rows, d, L.ptr(x2), L.ptr(gamma), L.ptr(mean), L.ptr(rstd), L.ptr(dy2), L.ptr(dx), L.ptr(dg),
                                    L.ptr(db), C.c_void_p(ws.data_ptr()), L.stream_ptr()), "caphn_layernorm_bwd")
    return dx.view(x.shape), dg, db


def dropout(x: torch.Tensor, p: float, seed: int, offset: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x * keep / (1 - p) with the counter-based mask of (seed, offset + i); calling it on a gradient with the same
    seed is the backward."""
    lib = L.load()
    if not x.is_contiguous():
        x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    L.check(lib.caphn_dropout_f32(x.numel(), float(p), int(seed) & (2 ** 64 - 1), int(offset), L.ptr(x), L.ptr(out), L.stream_ptr()),
            "caphn_dropout_f32")
    return out


def _attn_dims(q, k, v, o, nh: int, dropout_p: float = 0.0, seed: int = 0) -> L.AttnDims:
    """q [tq, bs, d], k / v [tk, bs, d], o [tq, bs, d]: sequence-first views whose last stride is 1 (slices of a packed
    projection are fine)."""
    tq, bs, dm = q.shape
    tk = k.shape[0]
    for t in (q, k, v, o):
        if t.dim() != 3 or t.stride(2) != 1 or t.shape[1] != bs or t.shape[2] != dm or t.dtype != torch.float32 or not t.is_cuda:
            raise L.CaphnError("attention operands must be fp32 CUDA [T, bs, d] views with unit last stride")
    if v.shape[0] != tk or o.shape[0] != tq or dm % nh:
        raise L.CaphnError("attention shapes do not match")
    dh = dm // nh
    d = L.AttnDims(bs, nh, dh, tq, tk, q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                   o.stride(0), o.stride(1), 1.0 / (dh ** 0.5), float(dropout_p), int(seed) & (2 ** 64 - 1))
    if not L.load().caphn_attention_supported(C.byref(d)):
        raise L.CaphnError(f"attention shape not supported by libcaphn (heads of {dh}, {tq} x {tk} positions)")
    return d


def attention_fwd(q, k, v, nh: int, attn_mask: Optional[torch.Tensor] = None, key_padding: Optional[torch.Tensor] = None,
                  dropout_p: float = 0.0, seed: int = 0):
    """softmax(q k^T / sqrt(dh) + attn_mask + key padding) v per head -> (o [tq, bs, d], lse [bs*nh, tq]).  dropout_p > 0:
    the probabilities are dropped with the mask dropout(ones[bs*nh, tq, tk], dropout_p, seed) before they multiply v."""
    lib = L.load()
    o = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    d = _attn_dims(q, k, v, o, nh, dropout_p, seed)
    lse = _f32(d.bs * nh, d.tq, device=q.device)
    L.check(lib.caphn_attention_fwd(C.byref(d), C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()),
                                    L.ptr(attn_mask, allow_none=True), L.ptr(key_padding, torch.uint8, allow_none=True),
                                    L.ptr(o), L.ptr(lse), L.stream_ptr()), "caphn_attention_fwd")
    return o, lse


def attention_bwd(q, k, v, nh: int, attn_mask, key_padding, o, lse, d_o, dropout_p: float = 0.0, seed: int = 0):
    lib = L.load()
    d_o = d_o.contiguous()
    if d_o.stride() != o.stride():
        raise L.CaphnError("d_o must share o's layout")
    d = _attn_dims(q, k, v, o, nh, dropout_p, seed)
    dq, dk, dv = torch.empty_strided(q.shape, q.stride(), dtype=torch.float32, device=q.device), \
        torch.empty_strided(k.shape, k.stride(), dtype=torch.float32, device=q.device), \
        torch.empty_strided(v.shape, v.stride(), dtype=torch.float32, device=q.device)
    L.check(lib.caphn_attention_bwd(C.byref(d), C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()),
                                    L.ptr(attn_mask, allow_none=True), L.ptr(key_padding, torch.uint8, allow_none=True),
                                    L.ptr(o), L.ptr(lse), L.ptr(d_o), C.c_void_p(dq.data_ptr()), C.c_void_p(dk.data_ptr()),
                                    C.c_void_p(dv.data_ptr()), L.stream_ptr()), "caphn_attention_bwd")
    return dq, dk, dv
