"""ctypes binding of libcaphn.so (C ABI: include/caphn.h).

The product path has no CPU or eager-PyTorch fallback: if the HIP library is missing or a
call fails, this module raises.  Build it with ``python __graft_entry__.py build`` (or
``make -C hypernet-image-captioning_amd/csrc``).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAPHN_LIB_PATH") or os.path.join(_HERE, "libcaphn.so")      # (override: profiling builds, tools/ only)
MAX_HEADS = 8

c_fp = C.c_void_p  # device pointers travel as void*


class HyperDesc(C.Structure):
    _fields_ = [("he", C.c_int), ("n_heads", C.c_int),
                ("k", C.c_int * MAX_HEADS), ("w", C.c_int * MAX_HEADS),
                ("base_w0", c_fp), ("base_b0", c_fp), ("base_w2", c_fp), ("base_b2", c_fp),
                ("w1", c_fp * MAX_HEADS), ("b1", c_fp * MAX_HEADS),
                ("w2", c_fp * MAX_HEADS), ("b2", c_fp * MAX_HEADS),
                ("d_in", C.c_int), ("d_mid", C.c_int)]


class HyperGrads(C.Structure):
    _fields_ = [("g_base_w0", c_fp), ("g_base_b0", c_fp), ("g_base_w2", c_fp), ("g_base_b2", c_fp),
                ("g_w1", c_fp * MAX_HEADS), ("g_b1", c_fp * MAX_HEADS),
                ("g_w2", c_fp * MAX_HEADS), ("g_b2", c_fp * MAX_HEADS),
                ("g_x", c_fp), ("x_accumulate", C.c_int)]


class DecoderDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "T", "P", "D", "F", "E", "H", "V", "cell", "raw_features", "row_subset",
                                         "grads_zeroed", "precomputed", "layers")] + [("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
                                                                                       ("logits_ld", C.c_int)]


_DEC_FIELDS = ("fc0_w", "fc0_b", "fc2_w", "fc2_b", "embed_w", "out_w", "out_b", "Wa_w", "Wa_b",
               "Ua_w", "Ua_b", "va_w", "va_b", "inith_w", "inith_b", "w_ih", "w_hh", "b_ih", "b_hh",
               "initc_w", "initc_b") + tuple(f"{a}{i}" for a in ("lw_ih", "lw_hh", "lb_ih", "lb_hh") for i in range(3))


class DecoderParams(C.Structure):
    _fields_ = [(n, c_fp) for n in _DEC_FIELDS]


class DecoderGrads(C.Structure):
    _fields_ = [(n, c_fp) for n in _DEC_FIELDS]


class AdamHParams(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int), ("dev_scalars", c_fp), ("zero_gfac", C.c_int)]


class RankJob(C.Structure):
    _fields_ = [("W", c_fp), ("m", c_fp), ("v", c_fp), ("gfac", c_fp), ("ldg", C.c_size_t), ("afac", c_fp), ("lda", C.c_size_t),
                ("next_a", c_fp), ("next_bias", c_fp), ("next_theta", c_fp), ("rows", C.c_int), ("k", C.c_int),
                ("next_pack", c_fp), ("pack_H", C.c_int), ("pack_HA", C.c_int), ("pack_pitch", C.c_int), ("pack_hrows", C.c_int)]


class PairPack(C.Structure):
    _fields_ = [("wp", c_fp), ("H", C.c_int), ("HA", C.c_int), ("pitch", C.c_int), ("hrows", C.c_int)]


MAX_LAYERS = 4


class PlainDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "T", "E", "H", "V", "L", "cell")]


class AttnDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("bs", "nh", "dh", "tq", "tk", "q_ldt", "q_ldb", "k_ldt", "k_ldb", "v_ldt", "v_ldb",
                                       "o_ldt", "o_ldb")] + [("scale", C.c_float), ("dropout_p", C.c_float),
                                                             ("seed", C.c_uint64)]


class PlainParams(C.Structure):
    _fields_ = [("embed_w", c_fp), ("out_w", c_fp), ("out_b", c_fp),
                ("w_ih", c_fp * MAX_LAYERS), ("w_hh", c_fp * MAX_LAYERS), ("b_ih", c_fp * MAX_LAYERS), ("b_hh", c_fp * MAX_LAYERS)]


class PlainGrads(C.Structure):
    _fields_ = [("embed_w", c_fp), ("out_w", c_fp), ("out_b", c_fp),
                ("w_ih", c_fp * MAX_LAYERS), ("w_hh", c_fp * MAX_LAYERS), ("b_ih", c_fp * MAX_LAYERS), ("b_hh", c_fp * MAX_LAYERS),
                ("features", c_fp)]


class SearchCfg(C.Structure):
    _fields_ = [("n_images", C.c_int), ("beam", C.c_int), ("max_steps", C.c_int), ("zero_pad_rule", C.c_int),
                ("lookup_first", C.c_int), ("first_token", C.c_int64), ("end_token", C.c_int64)]


# name -> (restype, argtypes); must list every symbol include/caphn.h declares
SIGNATURES = {
    "caphn_abi_version": (C.c_int, []),
    "caphn_device_arch": (C.c_int, [C.c_char_p, C.c_int]),
    "caphn_device_error": (C.c_int, [C.c_int]),
    "caphn_gemm_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp, C.c_int,
                                 c_fp, C.c_int, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp]),
    "caphn_split3_bf16": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, C.c_size_t, C.c_int, c_fp]),
    "caphn_gemm_planes_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp, C.c_int, C.c_size_t,
                                        c_fp, C.c_int, c_fp, C.c_int, C.c_size_t, c_fp, C.c_int, c_fp, c_fp, C.c_int, C.c_int,
                                        C.c_int, C.c_int, c_fp]),
    "caphn_zero_f32": (C.c_int, [c_fp, C.c_size_t, c_fp]),
    "caphn_lrelu_bwd_f32": (C.c_int, [C.c_size_t, c_fp, c_fp, c_fp, c_fp]),
    "caphn_axpy_f32": (C.c_int, [C.c_size_t, C.c_float, c_fp, c_fp, c_fp]),
    "caphn_scale_f32": (C.c_int, [C.c_size_t, c_fp, c_fp, c_fp, c_fp]),
    "caphn_add_dropout_f32": (C.c_int, [C.c_size_t, c_fp, c_fp, C.c_float, C.c_uint64, C.c_uint64, c_fp, c_fp]),
    "caphn_dropout_f32": (C.c_int, [C.c_size_t, C.c_float, C.c_uint64, C.c_uint64, c_fp, c_fp, c_fp]),
    "caphn_colsum_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "caphn_linear_wgrad_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp, C.c_int, c_fp, C.c_int, c_fp, c_fp, c_fp]),
    "caphn_colsum_f32": (C.c_int, [C.c_int, C.c_int, c_fp, C.c_int, c_fp, c_fp, c_fp]),
    "caphn_hyper_acts_floats": (C.c_int, [C.POINTER(HyperDesc)]),
    "caphn_hyper_forward": (C.c_int, [C.POINTER(HyperDesc), c_fp, c_fp, c_fp, c_fp]),
    "caphn_hyper_backward_workspace_bytes": (C.c_size_t, [C.POINTER(HyperDesc)]),
    "caphn_hyper_backward": (C.c_int, [C.POINTER(HyperDesc), c_fp, c_fp, C.POINTER(HyperGrads), c_fp, c_fp]),
    "caphn_decoder_workspace_bytes": (C.c_size_t, [C.POINTER(DecoderDims)]),
    "caphn_decoder_forward": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), c_fp, c_fp,
                                        c_fp, c_fp, c_fp, c_fp]),
    "caphn_decoder_pair_prep": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), c_fp, c_fp]),
    "caphn_decoder_pair_pack_desc": (C.c_int, [C.POINTER(DecoderDims), c_fp, C.POINTER(PairPack)]),
    "caphn_decoder_precompute": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), c_fp, c_fp, c_fp, c_fp]),
    "caphn_decoder_inputs": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), c_fp, c_fp, c_fp]),
    "caphn_decoder_lookup": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), c_fp, c_fp, c_fp]),
    "caphn_decoder_prepare_rows": (C.c_int, [C.POINTER(DecoderDims), c_fp, C.c_int64, c_fp, c_fp]),
    "caphn_decoder_forward_sampled": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), c_fp, c_fp,
                                                C.c_char_p, c_fp, c_fp, c_fp, c_fp]),
    "caphn_decoder_forward_sampled_train": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), c_fp, c_fp,
                                                      C.c_char_p, c_fp, c_fp, c_fp, c_fp]),
    "caphn_decoder_backward": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), c_fp, c_fp,
                                         c_fp, c_fp, C.POINTER(DecoderGrads), c_fp, c_fp]),
    "caphn_decoder_hyper_backward": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), c_fp, c_fp,
                                               c_fp, c_fp, C.POINTER(DecoderGrads), c_fp,
                                               C.POINTER(HyperDesc), c_fp, C.POINTER(HyperGrads), c_fp, c_fp]),
    "caphn_decoder_backward_milestone": (C.c_int, [C.c_int, c_fp]),
    "caphn_decoder_search_workspace_bytes": (C.c_size_t, [C.POINTER(DecoderDims), C.POINTER(SearchCfg)]),
    "caphn_decoder_search_begin": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), C.POINTER(SearchCfg),
                                             c_fp, c_fp, c_fp, c_fp]),
    "caphn_decoder_search_steps": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(DecoderParams), C.POINTER(SearchCfg),
                                             C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]),
    "caphn_decoder_search_result": (C.c_int, [C.POINTER(DecoderDims), C.POINTER(SearchCfg), C.c_int, c_fp, c_fp,
                                              c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "caphn_plain_workspace_bytes": (C.c_size_t, [C.POINTER(PlainDims)]),
    "caphn_plain_forward": (C.c_int, [C.POINTER(PlainDims), C.POINTER(PlainParams), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "caphn_bahdanau_fwd": (C.c_int, [C.c_int] * 4 + [c_fp] * 8),
    "caphn_bahdanau_bwd": (C.c_int, [C.c_int] * 4 + [c_fp] * 12),
    "caphn_plain_forward_sampled": (C.c_int, [C.POINTER(PlainDims), C.POINTER(PlainParams), c_fp, c_fp, c_fp, C.c_uint64, c_fp, c_fp,
                                               c_fp, c_fp]),
    "caphn_plain_backward": (C.c_int, [C.POINTER(PlainDims), C.POINTER(PlainParams), c_fp, c_fp, c_fp, c_fp, c_fp,
                                       C.POINTER(PlainGrads), c_fp, c_fp]),
    "caphn_layernorm_fwd": (C.c_int, [C.c_int, C.c_int, c_fp, c_fp, c_fp, C.c_float, c_fp, c_fp, c_fp, c_fp]),
    "caphn_layernorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "caphn_layernorm_bwd": (C.c_int, [C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "caphn_attention_supported": (C.c_int, [C.POINTER(AttnDims)]),
    "caphn_attention_fwd": (C.c_int, [C.POINTER(AttnDims), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "caphn_attention_bwd": (C.c_int, [C.POINTER(AttnDims), c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "caphn_ce_workspace_bytes": (C.c_size_t, [C.c_int]),
    "caphn_cross_entropy_rows": (C.c_int, [C.c_int, C.c_int, c_fp, c_fp, C.c_int64, c_fp, C.c_int, c_fp, c_fp, c_fp]),
    "caphn_cross_entropy_rows_ld": (C.c_int, [C.c_int, C.c_int, C.c_int, c_fp, c_fp, C.c_int64, c_fp, C.c_int, c_fp, c_fp, c_fp]),
    "caphn_cross_entropy_finish": (C.c_int, [C.c_int, c_fp, c_fp, c_fp, c_fp]),
    "caphn_decoder_profile_ptr": (C.c_void_p, [C.POINTER(DecoderDims), c_fp]),
    "caphn_decoder_rowcount_ptr": (C.c_void_p, [C.POINTER(DecoderDims), c_fp]),
    "caphn_cross_entropy_fwd_bwd": (C.c_int, [C.c_int, C.c_int, c_fp, c_fp, C.c_int64, c_fp, c_fp, C.c_int, c_fp, c_fp]),
    "caphn_embedding_gather": (C.c_int, [C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]),
    "caphn_embedding_scatter_add": (C.c_int, [C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]),
    "caphn_embedding_scatter_add_v": (C.c_int, [C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]),
    "caphn_sumsq_blocks": (C.c_int, [C.c_size_t]),
    "caphn_sumsq_f32": (C.c_int, [C.c_size_t, c_fp, c_fp, c_fp]),
    "caphn_rank_sumsq_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, c_fp, C.c_size_t, c_fp, C.c_size_t,
                                       c_fp, c_fp, c_fp]),
    "caphn_rank_sumsq_multi_f32": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_size_t), c_fp, c_fp, c_fp]),
    "caphn_grad_norm_workspace_bytes": (C.c_size_t, [C.c_size_t, C.c_int, C.c_int]),
    "caphn_grad_norm_coef": (C.c_int, [C.c_size_t, c_fp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t), C.c_double, C.c_double, c_fp, c_fp, c_fp]),
    "caphn_clip_coef": (C.c_int, [C.c_int, c_fp, c_fp, C.c_double, C.c_double, c_fp, c_fp]),
    "caphn_adam_dense_f32": (C.c_int, [C.c_size_t, c_fp, c_fp, c_fp, c_fp, c_fp, C.POINTER(AdamHParams), c_fp]),
    "caphn_grad_norm_adam_dense": (C.c_int, [C.c_size_t, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                             C.c_double, C.c_double, c_fp, c_fp, C.POINTER(AdamHParams), C.c_int, c_fp, c_fp, c_fp, c_fp]),
    "caphn_grad_norm_multi_workspace_bytes": (C.c_size_t, [C.c_int, C.POINTER(C.c_size_t), C.c_int, C.c_int]),
    "caphn_grad_norm_multi": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_double, C.c_double, c_fp, c_fp, c_fp]),
    "caphn_adam_multi_f32": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), c_fp, C.POINTER(AdamHParams), c_fp]),
    "caphn_adam_rank_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_size_t,
                                      c_fp, C.c_size_t, c_fp, C.POINTER(AdamHParams), c_fp]),
    "caphn_adam_rank_gemv_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_size_t,
                                           c_fp, C.c_size_t, c_fp, C.POINTER(AdamHParams), c_fp, c_fp, c_fp, c_fp]),
    "caphn_adam_rank_multi_f32": (C.c_int, [C.c_int, C.c_int, C.POINTER(RankJob), c_fp, C.POINTER(AdamHParams), c_fp]),
    "caphn_hyper_forward_acts": (C.c_int, [C.POINTER(HyperDesc), c_fp, c_fp, c_fp]),
    "caphn_stream_copy_f32": (C.c_int, [C.c_size_t, c_fp, c_fp, c_fp]),
    "caphn_outer_f32": (C.c_int, [C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]),
    "caphn_tune": (C.c_int, [C.c_int, C.c_int]),
}

_ERR = {-1: "CAPHN_EINVAL (bad argument)", -2: "CAPHN_ELAUNCH (HIP launch error)",
        -3: "CAPHN_ELIMIT (problem exceeds a hardware limit, e.g. 160 KB LDS)",
        -4: "CAPHN_ETIMEOUT (a recurrent kernel's partner workgroup never answered: that step's results are NaN; "
            "sticky until caphn_device_error(1))"}

_lib = None


class CaphnError(RuntimeError):
    pass


def load():
    """Load libcaphn.so and attach signatures.  Raises if the library is missing: there is no
    fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CaphnError(f"{LIB_PATH} not found: build the HIP extension first "
                         "(python __graft_entry__.py build). There is no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise CaphnError(f"{what} failed: {_ERR.get(rc, rc)}")


def stream_ptr():
    """Raw hipStream_t of torch's current stream (the direct binding: ~0.3 us instead of ~8 us for
    torch.cuda.current_stream().cuda_stream -- sixteen of them per training step)."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def ptr(t, dtype=torch.float32, allow_none=False):
    """Device pointer of a contiguous CUDA tensor (validated: the kernels trust shapes)."""
    if t is None:
        if allow_none:
            return None
        raise CaphnError("required tensor is None")
    if not t.is_cuda:
        raise CaphnError("libcaphn needs CUDA(HIP) tensors; got a CPU tensor (no CPU fallback exists)")
    if t.dtype != dtype:
        raise CaphnError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise CaphnError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())
