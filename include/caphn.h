/*
 * caphn.h -- C ABI of libcaphn.so: the MI355X (gfx950) hot path of the
 * hypernetwork-conditioned captioning training step.
 *
 * The reference (zacharie12/Hypernet-image-captioning) has no FFI layer: its hot path is
 * Python nn.Modules.  Each entry point below replaces the arithmetic of one reference
 * function; the Python modules under hypernet-image-captioning_amd/ (same names and
 * signatures as the reference's) bind these symbols with ctypes.  File:line citations are
 * relative to the reference root.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (or int64 where stated) unless marked host;
 *    matrices are row-major with explicit leading dimensions (in elements);
 *  - every function enqueues work on `stream` and returns immediately: 0 on success,
 *    a negative CAPHN_E* code otherwise; nothing throws, nothing allocates device memory, nothing
 *    synchronises (graph-capture safe); scratch comes from the caller (`*_workspace_bytes`);
 *  - Hidden state: the decoder composites (caphn_decoder_precompute / _forward / _backward /
 *    _hyper_backward) run independent branches on side streams.  The library keeps ONE set of three
 *    non-blocking streams and their fork/join/milestone events PER DEVICE, created lazily by the first
 *    composite call made with that device current (creation is mutex-guarded; create it outside graph
 *    capture by running one eager step first) and never destroyed.  Branches are forked from and
 *    joined to `stream` with events, so the call is still "everything is ordered after `stream` so far,
 *    and `stream` is ordered after everything" -- and a capturing stream captures the branches too.
 *    USE of a device's set is not thread-safe: one host thread drives one device.  caphn_tune values
 *    are process-global.  Nothing else is kept between calls;
 *  - parameter tensors keep the reference's layout: nn.Linear weight [out,in], GRUCell
 *    weight_ih [3H,E+F] / weight_hh [3H,H], gate order r,z,n.
 */
#ifndef CAPHN_H
#define CAPHN_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* caphn_stream_t; /* hipStream_t */

#define CAPHN_OK 0
#define CAPHN_EINVAL (-1)    /* bad argument (null pointer, non-positive size, unsupported combination) */
#define CAPHN_ELAUNCH (-2)   /* HIP reported a launch error */
#define CAPHN_ELIMIT (-3)    /* problem does not fit a hardware limit (e.g. LDS) */
#define CAPHN_ETIMEOUT (-4)  /* a kernel gave up waiting for a partner workgroup (see caphn_device_error); STICKY: every later
                                call on that device returns it until caphn_device_error(1) clears the word */

/* Library / device probe.  Returns the ABI version (>0).  */
int caphn_abi_version(void);
/* Device-side failure word of the CURRENT device (one int in pinned host memory per device, written by kernels with a
 * system-scope store, read by the host without synchronising).  The only writer today: the two-workgroups-per-caption recurrent
 * kernels (teacher-forced caphn_decoder_forward / _backward at T > 1).  Their halves hand partial vectors to each other through
 * tagged 8-byte granules in the workspace and ASSUME CO-RESIDENCY: workgroups w and w ^ 8 of one launch must both be running for
 * either to finish a time step.  An in-order dispatcher with at least 9 free workgroup slots guarantees progress; a partner that
 * does not answer within the time bound (about 1 s of the 100 MHz wall clock; caphn_tune key 22 sets it in microseconds) makes
 * the waiting lane store 1 here and continue with NaN, so the loss and every gradient of that step are NaN as well -- never a
 * plausible wrong value -- and every workgroup still polling gives up within a millisecond of the first one.
 * Returns CAPHN_OK or CAPHN_ETIMEOUT; clear != 0 resets the word after reading it.  Every libcaphn call that launches work
 * also returns CAPHN_ETIMEOUT while the word is set (the launch itself is not skipped), so a host that never synchronises
 * still learns of the failure at its next call after the kernel has written it. */
int caphn_device_error(int clear);
/* Writes the gfx arch name of the current device into buf (host); 0 on success. */
int caphn_device_arch(char* buf, int buflen);

/* ---------------------------------------------------------------------------------------
 * Dense contraction on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32: exact fp32 products,
 * fp32 accumulate).  C[M,N] = epilogue( op(A) . op(B) ).
 *   ta == 0: A is [M,K] (lda)   ta == 1: A is stored [K,M] (lda) and used transposed
 *   tb == 0: B is [K,N] (ldb)   tb == 1: B is stored [N,K] (ldb) and used transposed
 * so nn.Linear forward  y = x W^T + b  is (ta=0,tb=1)            [models/decoderlstm.py:22-26,38,105]
 *    its input gradient dx = dy W      is (ta=0,tb=0)
 *    its weight gradient dW = dy^T x   is (ta=1,tb=0).
 * flags: CAPHN_GEMM_*.  bias is [N] or NULL.  mask (CAPHN_GEMM_MASK) is [M,N] with ldmask: the
 * result is zeroed where mask <= 0 (ReLU backward).  splitk > 1 partitions K over grid.z and
 * accumulates into C with fp32 atomics: C must be pre-initialised (zero, or the value to add to)
 * and only CAPHN_GEMM_BIAS is honoured (added by slice 0).
 */
#define CAPHN_GEMM_BIAS 1
#define CAPHN_GEMM_RELU 2
#define CAPHN_GEMM_ACCUM 4   /* C += result (read-modify-write, no atomics) */
#define CAPHN_GEMM_MASK 8
#define CAPHN_GEMM_LRELU 16  /* nn.LeakyReLU() (slope 0.01) on the result: the domain front-ends of cc_train_hypernet.py:96-106 */
int caphn_gemm_f32(int ta, int tb, int M, int N, int K,
                   const float* A, int lda, const float* B, int ldb,
                   float* C, int ldc, const float* bias,
                   const float* mask, int ldmask, int flags, int splitk,
                   caphn_stream_t stream);

/* Pre-split operands.  The split-bf16 back end multiplies every fp32 operand as three bf16 planes x = hi + mid + lo (exact:
 * 8 + 8 + 8 significand bits, bf16 has fp32's exponent range).  caphn_gemm_f32 makes the planes tile by tile, on every use
 * of a tile; an operand that is used more than once (a weight matrix, an activation that feeds the forward and two
 * gradient contractions) can be split ONCE instead:
 *   caphn_split3_bf16: planes[p][r][c], p = 0..2 (hi, mid, lo), as bf16 matrices with leading dimension ldp (elements,
 *     multiple of 8, >= cols rounded up to 8: row tails are zero-filled) and plane_stride elements between planes (multiple
 *     of 8, >= (rows + zero_rows) * ldp); zero_rows rows of zeros are appended (a K extent rounded up to 8).  planes must be
 *     16-byte aligned.
 *   caphn_gemm_planes_f32: caphn_gemm_f32 with the planes of A and B beside the fp32 matrices (same logical layouts).  The
 *     planes are used when both are given, 16-byte aligned, and K % 8 == 0 -- or Kp = K rounded up to 8 is passed and both
 *     operands' planes hold zeros for k in [K, Kp); otherwise the fp32 matrices are (always a correct result). */
int caphn_split3_bf16(const float* src, int rows, int cols, int ld, void* planes, int ldp, size_t plane_stride, int zero_rows,
                      caphn_stream_t stream);
int caphn_gemm_planes_f32(int ta, int tb, int M, int N, int K,
                          const float* A, int lda, const void* Ap, int ldap, size_t psa,
                          const float* B, int ldb, const void* Bp, int ldbp, size_t psb,
                          float* C, int ldc, const float* bias, const float* mask, int ldmask, int flags, int splitk,
                          int Kp, caphn_stream_t stream);

/* p[0..n) = 0 with dwordx4 stores (hipMemsetAsync's fill kernel is ~8x slower on large buffers). */
int caphn_zero_f32(float* p, size_t n, caphn_stream_t stream);
/* nn.Dropout in training mode: out[i] = in[i] * keep_i / (1 - p), keep_i decided by a counter-based hash of (seed, offset + i)
   (splitmix64; NOT torch's Philox stream, so masks differ from the reference's -- statistically equivalent, parity unpinned).
   The mask is never stored: the same call on the gradient is the backward.  in may equal out.  0 <= p < 1. */
int caphn_dropout_f32(size_t n, float p, unsigned long long seed, unsigned long long offset, const float* in, float* out,
                      caphn_stream_t stream);
/* out[i] = x[i] + branch[i] * keep_i / (1 - p): residual connection with dropout on the branch (baseline/transformer.py:140,
   :161 ...: src + self.dropout1(src2)); the mask is caphn_dropout_f32's for the same (seed, offset); p = 0 is a plain sum. */
int caphn_add_dropout_f32(size_t n, const float* x, const float* branch, float p, unsigned long long seed,
                          unsigned long long offset, float* out, caphn_stream_t stream);
/* out[i] = x[i] * scale_dev[0]: the chain rule through a scalar loss whose upstream gradient lives on the device (x may equal out). */
int caphn_scale_f32(size_t n, const float* x, const float* scale_dev, float* out, caphn_stream_t stream);
/* nn.LeakyReLU() backward from the layer's OUTPUT: dx[i] = dy[i] * (post[i] > 0 ? 1 : 0.01)   (dx may equal dy). */
int caphn_lrelu_bwd_f32(size_t n, const float* dy, const float* post, float* dx, caphn_stream_t stream);
/* y[0..n) += alpha * x[0..n): sums the gradients of parameters that are views of one theta range (utils.py:62-68: every
   child module restarts at offset 0, so hypernet.py's extra layers alias the first cell's slices). */
int caphn_axpy_f32(size_t n, float alpha, const float* x, float* y, caphn_stream_t stream);
/* nn.Linear's parameter gradients in one launch: dW[M,N] = dY^T X and db[M] = column sums of dY, with dY [K,M] (ldy) and
   X [K,N] (ldx).  ws: caphn_colsum_workspace_bytes(K, M).  Split-K is chosen from the shape. */
int caphn_linear_wgrad_f32(int M, int N, int K, const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw,
                           float* db, void* ws, caphn_stream_t stream);
/* out[n] = sum_m A[m,n]  (bias gradients).  ws: caphn_colsum_workspace_bytes(M,N). */
size_t caphn_colsum_workspace_bytes(int M, int N);
int caphn_colsum_f32(int M, int N, const float* A, int lda, float* out, void* ws, caphn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Hypernetwork: hn_base (2 x Linear+LeakyReLU) and per-parameter heads
 * Linear(he,k_i)+LeakyReLU+Linear(k_i,w_i), input is ONE row (M = 1): every product is a
 * GEMV and the second head layers (w_i x k_i, 576 MB fp32 at the canonical size) are streamed
 * from HBM exactly once per call.        [hypernet_attention.py:55-99 (shapes), :111-118 (forward)]
 */
#define CAPHN_MAX_HEADS 8
typedef struct caphn_hyper_desc {
    int he;                       /* width of the base output = input width of every head (also of x and of the base's
                                     hidden layer unless d_in / d_mid say otherwise)                                     */
    int n_heads;                  /* 4 for GRUCell / LSTMCell, 8 for hypernet.py's two-layer decoder                     */
    int k[CAPHN_MAX_HEADS];       /* hidden width of head i                      */
    int w[CAPHN_MAX_HEADS];       /* output size of head i (= numel of the generated parameter) */
    const float* base_w0; const float* base_b0;   /* hn_base.0  [he,he],[he] */
    const float* base_w2; const float* base_b2;   /* hn_base.2  [he,he],[he] */
    const float* w1[CAPHN_MAX_HEADS]; const float* b1[CAPHN_MAX_HEADS]; /* hn_heads.i.0 [k_i,he],[k_i] */
    const float* w2[CAPHN_MAX_HEADS]; const float* b2[CAPHN_MAX_HEADS]; /* hn_heads.i.2 [w_i,k_i],[w_i] */
    int d_in, d_mid;              /* 0 = he.  hypernet.py:55-60 builds hn_base = Linear(E,4E), LeakyReLU, Linear(4E,8E),
                                     LeakyReLU: d_in = E, d_mid = 4E, he = 8E; base_w0 [d_mid,d_in], base_w2 [he,d_mid]  */
} caphn_hyper_desc;

/* acts layout (floats), every segment padded to a multiple of 4 floats:
 * [x (d_in) | a0 (d_mid) | base (he) | a_0 (k_0) | ... | a_{n-1}] ; size = caphn_hyper_acts_floats; acts must be 16-byte aligned */
int caphn_hyper_acts_floats(const caphn_hyper_desc* d);
/* theta[sum w_i] = cat_i head_i(hn_base(x)); acts receives the post-LeakyReLU activations
 * needed by the backward (and, data-parallel, exchanged as rank-1 factors). */
int caphn_hyper_forward(const caphn_hyper_desc* d, const float* x, float* theta, float* acts,
                        caphn_stream_t stream);

/* Only hn_base and the heads' first layers: fills acts (x, a0, base, a_i) without streaming the second layers.
 * Used with caphn_adam_rank_gemv_f32 to produce the NEXT step's theta during the optimiser pass. */
int caphn_hyper_forward_acts(const caphn_hyper_desc* d, const float* x, float* acts, caphn_stream_t stream);

/* Gradient sinks of caphn_hyper_backward.  Any pointer may be NULL (that gradient is skipped),
 * except that the chain needs what lies downstream of a requested gradient.  The second-layer
 * weight gradient dW2_i = dtheta_i (x) a_i is rank-1 and is only materialised when g_w2[i] != NULL
 * (the module API / torch optimisers need it dense; the fused optimiser does not). */
typedef struct caphn_hyper_grads {
    float* g_base_w0; float* g_base_b0; float* g_base_w2; float* g_base_b2;
    float* g_w1[CAPHN_MAX_HEADS]; float* g_b1[CAPHN_MAX_HEADS];
    float* g_w2[CAPHN_MAX_HEADS]; float* g_b2[CAPHN_MAX_HEADS];
    float* g_x;                   /* [d_in] gradient w.r.t. the style/domain embedding row */
    int x_accumulate;             /* 1: g_x is ADDED to (fp32 atomics) instead of written -- point g_x at the style token's row of the
                                     embedding gradient (hypernet_attention.py:139-142: x = captioner.embed(style)) and the row's VJP
                                     needs no scatter-add launch of its own; the row must hold its other contributions or zero */
} caphn_hyper_grads;
size_t caphn_hyper_backward_workspace_bytes(const caphn_hyper_desc* d);
/* VJP of caphn_hyper_forward with dtheta (what autograd would give the reference had utils.py:57
 * not detached theta; SURVEY.md 8a H3 "intended" gradients). */
int caphn_hyper_backward(const caphn_hyper_desc* d, const float* dtheta, const float* acts,
                         const caphn_hyper_grads* g, void* ws, caphn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Decoder: AttentionGru.forward with teacher forcing            [models/decoderlstm.py:49-120]
 *          AttentionLstm.forward (same loop, LSTMCell, init_c)  [models/decoderlstm.py:224-261]
 * + BahdanauAttention.forward                                   [models/attention.py:21-46]
 * + nn.GRUCell / nn.LSTMCell arithmetic                         [decoderlstm.py:32,100 / :209,243]
 */
#define CAPHN_CELL_GRU 0     /* nn.GRUCell,  gates r,z,n   (AttentionGru,  models/decoderlstm.py:32)  */
#define CAPHN_CELL_LSTM 1    /* nn.LSTMCell, gates i,f,g,o (AttentionLstm, models/decoderlstm.py:209) */
typedef struct caphn_decoder_dims {
    int B, T, P;        /* batch, caption length, attention positions (49) */
    int D, F, E, H, V;  /* encoder channels, feature_out, embedding_dim, hidden_dim, vocab */
    int cell;           /* CAPHN_CELL_* ; NG = 3 (GRU) or 4 (LSTM) gate blocks of H rows */
    int raw_features;   /* 1: no feature_fc, attention runs directly over the D-channel features
                           (reference AttentionLstm, decoderlstm.py:242); then F must equal D */
    int row_subset;     /* 1: the caller has run caphn_decoder_prepare_rows on this workspace: the vocab projection and
                           its two backward GEMMs touch only rows whose target is not ignored by the loss (logits rows
                           of ignored targets are left unwritten, their d logits must be zero).  For fused training;
                           the module API, which must return every logits row, uses 0. */
    int grads_zeroed;   /* 1: every gradient output of caphn_decoder_backward is already zero on entry (a trainer that keeps all
                           gradients in one arena clears it with one caphn_zero_f32): the composite then skips its ~16
                           per-tensor zero fills (split-K / atomic accumulation targets) */
    int precomputed;    /* bit mask of what already sits in this workspace for this minibatch, so caphn_decoder_forward
                           skips it: 1 = feature_fc / init_hidden / W_a f (caphn_decoder_precompute), 2 = G (the same call
                           given captions, i.e. after the generated W_ih is final), 4 = embedding lookup + x-side gate
                           pre-activations (caphn_decoder_inputs, or that same call).  7: the forward starts at the
                           recurrent kernel.  Bit 8, given to BOTH caphn_decoder_forward and the backward of the same step on the
                           same workspace: the forward leaves the backward's d Hs accumulator zero-filled (inside a kernel it
                           launches anyway) and the backward skips its own zero fill.  Bit 16 (with bit 1): the
                           caphn_decoder_precompute call ran on another stream and the caller did NOT wait for it: the forward
                           waits for it itself -- for the feature_fc output before its G GEMM, for the rest before the
                           recurrent kernel (not inside a stream capture).  Bit 32, given to BOTH the forward and the backward of a
                           training step: the forward also leaves the context vectors ctx_t = sum_p alpha_tp f_p (the operand of
                           dW_ih it never forms itself) in the workspace, and the backward skips that kernel on its chain to d theta.
                           Bit 64: the embedding lookup (caphn_decoder_lookup) is already in the workspace.  Bit 128 (pair recurrent
                           kernels only): caphn_decoder_pair_prep has run on this workspace since the last backward AND the packed
                           copy of W_hh is current (caphn_rank_job::next_pack): the forward skips its prep launch */
    int layers;         /* num_layers of AttentionGru (models/decoderlstm.py:34-36): 1 (or 0) = the cell alone; L > 1 adds L - 1 GRUCells
                           applied as h = layer(h, h) after the attention cell at every time step (:101-103).  Then the time loop
                           runs one launch window per step (the extra cells are small batched GEMMs + a gate kernel between the
                           windows).  GRU only, L <= CAPHN_MAX_DEC_LAYERS. */
    float dropout_p;    /* h = self.drop(h) (models/decoderlstm.py:44,104; AttentionLstm :254) in TRAINING mode: every h_t is
                           multiplied by keep / (1 - p) before it feeds fc, the next step's cell and the next step's attention.
                           0 = off (eval mode, or p = 0 as the hypernet path constructs its decoder).  The keep decision of
                           element (b, t, k) is caphn_dropout_f32's counter-based hash of (dropout_seed, (b T + t) H + k) -- NOT
                           torch's Philox stream: masks differ from the reference's, parity is pinned with dropout off and,
                           with it on, against the oracle given this mask.  The backward needs the same p and seed. */
    unsigned long long dropout_seed;
    int logits_ld;      /* row pitch (floats) of the logits / d logits buffer given to caphn_decoder_forward / _backward / _hyper_backward;
                           0 = V (contiguous [B,T,V], what the module API returns).  A trainer that owns the buffer pads the pitch to a
                           multiple of 32 floats: the three vocabulary GEMMs then read and write whole 128-byte lines (V = 9684: 38 736
                           bytes a row, every 128-byte store straddles two lines; measured on the canonical shapes: logits -5 us,
                           dHs -12 us, dW_fc -5 us).  >= V.  The free-running / sampled / search entry points need 0. */
} caphn_decoder_dims;

#define CAPHN_MAX_DEC_LAYERS 4
typedef struct caphn_decoder_params {   /* reference state_dict names in comments */
    const float* fc0_w; const float* fc0_b;   /* captioner.feature_fc.0  [F,D],[F] */
    const float* fc2_w; const float* fc2_b;   /* captioner.feature_fc.2  [F,F],[F] */
    const float* embed_w;                     /* captioner.embed.weight  [V,E]     */
    const float* out_w; const float* out_b;   /* captioner.fc            [V,H],[V] */
    const float* Wa_w; const float* Wa_b;     /* captioner.attention.W_a [H,F],[H] */
    const float* Ua_w; const float* Ua_b;     /* captioner.attention.U_a [H,H],[H] */
    const float* va_w; const float* va_b;     /* captioner.attention.v_a [1,H],[1] */
    const float* inith_w; const float* inith_b; /* captioner.init_h      [H,F],[H] */
    const float* w_ih; const float* w_hh;     /* cell weight_ih [NG*H,E+F], weight_hh [NG*H,H] (slices of theta) */
    const float* b_ih; const float* b_hh;     /* cell bias_ih [NG*H], bias_hh [NG*H] */
    const float* initc_w; const float* initc_b; /* captioner.init_c    [H,F],[H]  (LSTM only, else NULL) */
    /* captioner.layers.l (dims.layers > 1 only): GRUCell(H, H) l = 0 .. layers-2 -- weight_ih [3H,H], weight_hh [3H,H], biases [3H] */
    const float* lw_ih[CAPHN_MAX_DEC_LAYERS - 1]; const float* lw_hh[CAPHN_MAX_DEC_LAYERS - 1];
    const float* lb_ih[CAPHN_MAX_DEC_LAYERS - 1]; const float* lb_hh[CAPHN_MAX_DEC_LAYERS - 1];
} caphn_decoder_params;

typedef struct caphn_decoder_grads {    /* same shapes as the parameters; all required (feature_fc ones unless raw_features, init_c ones if LSTM) */
    float* fc0_w; float* fc0_b; float* fc2_w; float* fc2_b;
    float* embed_w;                     /* fully overwritten (zero + scatter-add) */
    float* out_w; float* out_b;
    float* Wa_w; float* Wa_b; float* Ua_w; float* Ua_b; float* va_w; float* va_b;
    float* inith_w; float* inith_b;
    float* w_ih; float* w_hh; float* b_ih; float* b_hh;   /* = dtheta, in theta order when contiguous */
    float* initc_w; float* initc_b;     /* LSTM only */
    float* lw_ih[CAPHN_MAX_DEC_LAYERS - 1]; float* lw_hh[CAPHN_MAX_DEC_LAYERS - 1];
    float* lb_ih[CAPHN_MAX_DEC_LAYERS - 1]; float* lb_hh[CAPHN_MAX_DEC_LAYERS - 1];
} caphn_decoder_grads;

/* Saved-activation workspace shared by forward and backward (one training step). */
size_t caphn_decoder_workspace_bytes(const caphn_decoder_dims* d);
/* features [B,P,D] fp32, captions [B,T] int64 -> logits [B,T,V], alphas [B,T,P].
 * Implements the reference's input quirk: x_0 = x_1 = 0, x_t = embed[caps[:,t-1]] for t >= 2
 * (decoderlstm.py:82-88, in-place zero of a view). */
int caphn_decoder_forward(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                          const float* features, const int64_t* captions,
                          float* logits, float* alphas, void* ws, caphn_stream_t stream);
/* The part of the forward that depends neither on the captions nor on the generated cell weights: feature_fc,
 * init_hidden (init_c) and the hoisted W_a f.  A trainer that knows the next minibatch's features can issue it on
 * another stream while the optimiser streams the hypernet (then set dims.precomputed = 1 for that forward).  With
 * captions it also produces G = f W_ih[:,E:]^T, the embedding lookup and the x-side gate pre-activations, which need
 * the generated W_ih / b_ih of THAT forward (dims.precomputed = 2).  dims.precomputed bit 4 given to THIS call: the x side is
 * left out (the caller issues caphn_decoder_inputs itself, e.g. on a third stream once b_ih exists), G is still produced; bit 1
 * given to THIS call: the theta-independent part is in the workspace already (an earlier call, before W_ih existed), only G is added. */
int caphn_decoder_precompute(const caphn_decoder_dims* d, const caphn_decoder_params* p, const float* features,
                             const int64_t* captions /* optional, see dims.precomputed */, void* ws, caphn_stream_t stream);
/* Embedding lookup (with the reference's zeroed first two inputs) and x-side gate pre-activations for all T; needs the
 * captions and the generated W_ih / b_ih but not the features: a trainer waiting for a side-stream precompute can issue
 * it first (dims.precomputed |= 4). */
int caphn_decoder_inputs(const caphn_decoder_dims* d, const caphn_decoder_params* p, const int64_t* captions, void* ws,
                         caphn_stream_t stream);
/* Only the embedding lookup of caphn_decoder_inputs (token ids + input rows, decoderlstm.py:62, :82-88): needs the captions and the
 * embedding table but NOT the generated cell weights, so a trainer that knows the next minibatch's captions can issue it beside
 * the optimiser; caphn_decoder_inputs / _forward with dims.precomputed bit 64 then skip the lookup and run the gate GEMM only. */
int caphn_decoder_lookup(const caphn_decoder_dims* d, const caphn_decoder_params* p, const int64_t* captions, void* ws,
                         caphn_stream_t stream);
/* The pair recurrent kernels' prep launch, issued ahead of the forward (e.g. beside the optimiser's rank-1 passes, on a side stream;
 * never while a forward / backward on this workspace is running): clears the exchange areas and the backward's d Hs accumulator and
 * packs U_a (attention.U_a, a dense parameter) into the per-half weight copy; the W_hh rows of that copy are left to the caller --
 * caphn_adam_rank_multi_f32 with caphn_rank_job::next_pack writes them while it produces the next theta.  CAPHN_EINVAL when these
 * dims do not run the pair kernels.  caphn_decoder_pair_pack_desc: where and how (device address inside ws, H, HA = rows of the
 * first half, row pitch, rows reserved per half). */
typedef struct caphn_pair_pack { float* wp; int H, HA, pitch, hrows; } caphn_pair_pack;
int caphn_decoder_pair_prep(const caphn_decoder_dims* d, const caphn_decoder_params* p, void* ws, caphn_stream_t stream);
int caphn_decoder_pair_pack_desc(const caphn_decoder_dims* d, void* ws, caphn_pair_pack* out);
/* Compacts the (b,t) rows whose target differs from ignore_index into a row map kept in the workspace (count stays on
 * the device: no host synchronisation).  Call before caphn_decoder_forward / _backward with dims.row_subset = 1. */
/* Device address of the live-row count that caphn_decoder_prepare_rows leaves in the workspace. */
const int* caphn_decoder_rowcount_ptr(const caphn_decoder_dims* d, void* ws);
int caphn_decoder_prepare_rows(const caphn_decoder_dims* d, const int64_t* targets, int64_t ignore_index,
                               void* ws, caphn_stream_t stream);
/* Tuning aid: device address of 16 64-bit counters in the workspace -- per-phase shader-clock sums of workgroup 0 of the
 * recurrent forward ([0..8)) and backward ([8..16)) kernels, written only by a library built with -DCAPHN_REC_PROFILE. */
unsigned long long* caphn_decoder_profile_ptr(const caphn_decoder_dims* d, void* ws);
/* Free-running / scheduled-sampling forward (validation, inference; keeps no backward state).
 * use_sampling is a HOST array of T flags: the reference's per-step draw np.random.random() < sample_prob
 * (decoderlstm.py:79-80); entry 0 is ignored (step 0 never samples).  A sampling step feeds back
 * embed[argmax logits] following the GRU rule (:89-96) or the LSTM's lagging rule (:236-251). */
int caphn_decoder_forward_sampled(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                  const float* features, const int64_t* captions,
                                  const unsigned char* use_sampling,
                                  float* logits, float* alphas, void* ws, caphn_stream_t stream);
/* The same forward KEEPING the state caphn_decoder_backward needs: training through scheduled sampling (train_gru.py:84 calls
 * the captioner with sample_prob 1.0 inside training_step).  The argmax feedback is not differentiable, so the gradient is
 * the teacher-forced one with the sampled token ids in place of the caption's (their embedding rows receive d x_t);
 * caphn_decoder_backward on the same dims / workspace follows.  row_subset and precomputed must be 0. */
int caphn_decoder_forward_sampled_train(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                        const float* features, const int64_t* captions,
                                        const unsigned char* use_sampling,
                                        float* logits, float* alphas, void* ws, caphn_stream_t stream);
/* dlogits [B,T,V] (may be overwritten) -> parameter gradients.  ws must be the workspace the
 * matching forward filled.  dalphas (gradient w.r.t. the returned attention weights) may be NULL.
 * The backward may be repeated on the same workspace (it only reads the forward's saved state; the hand-off granules of the
 * pair recurrent kernels carry a per-launch epoch, so a second backward never accepts what the first one left behind) as long as
 * dlogits is supplied again -- the first call may have overwritten it -- and, with dims.precomputed bit 8, d Hs is zero again. */
int caphn_decoder_backward(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                           const float* features, const int64_t* captions,
                           float* dlogits, const float* dalphas,
                           const caphn_decoder_grads* g, void* ws, caphn_stream_t stream);

/* caphn_decoder_backward followed by caphn_hyper_backward, with the hypernet VJP started on a side stream as
 * soon as dL/dtheta (g->w_ih, w_hh, b_ih, b_hh -- which must be contiguous in theta order, i.e. g->w_ih is
 * dtheta) is complete, so its 576 MB transposed GEMV overlaps the attention / feature_fc backward chain. */
int caphn_decoder_hyper_backward(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                 const float* features, const int64_t* captions,
                                 float* dlogits, const float* dalphas,
                                 const caphn_decoder_grads* g, void* ws,
                                 const caphn_hyper_desc* hd, const float* acts, const caphn_hyper_grads* hg, void* hyper_ws,
                                 caphn_stream_t stream);

/* Milestones inside the LAST caphn_decoder_backward / caphn_decoder_hyper_backward enqueued on the current device: makes
 * `waiter` wait (hipStreamWaitEvent) until that part of the backward has run, without waiting for the rest.  A
 * data-parallel trainer starts its gradient exchange from these while the tail of the backward is still executing
 * (SURVEY.md 8e "launched as soon as BPTT finishes each tensor"); `stream` of the composite itself is ordered after all
 * of them.  Call from the thread that enqueued the composite, after it returned.
 *   CAPHN_MS_DTHETA  g->w_ih, w_hh, b_ih, b_hh (= dL/dtheta, the row factors of the rank-1 second-layer gradients)
 *   CAPHN_MS_VOCAB   g->out_w, out_b
 *   CAPHN_MS_EMBED   g->embed_w (the caption tokens' rows)
 *   CAPHN_MS_HYPER   everything caphn_hyper_backward writes (hg->*, g_x); equals CAPHN_MS_DTHETA without a hypernet hook */
#define CAPHN_MS_DTHETA 0
#define CAPHN_MS_VOCAB 1
#define CAPHN_MS_EMBED 2
#define CAPHN_MS_HYPER 3
#define CAPHN_MS_COUNT 4
int caphn_decoder_backward_milestone(int which, caphn_stream_t waiter);

/* ---------------------------------------------------------------------------------------
 * Decoding (inference): beam search of HyperNet.test_step [hypernet_attention.py:251-306] and
 * AttentionGru.greedy_search [models/decoderlstm.py:138-175] for a batch of images, with the whole decode
 * state (beams, scores, sequences, completed lists) resident on the device.
 *
 * dims: B = n_images * beam rows, T = 1, cell = GRU; features [n_images, P, D] (raw_features = 1 with D = F when
 * they are already feature_fc outputs, as greedy_search receives them).  Beams of one image share its
 * attention slabs; h is re-ordered on the device.  Protocol: _begin once, _steps in chunks (steps are numbered
 * from 1; step0 + nsteps - 1 <= max_steps), _result whenever the host wants to look (n_active[0] = images whose
 * beam is not empty yet); nothing here synchronises.
 *
 * beam search : beam = k, first_token = 0, lookup_first = 0, zero_pad_rule = 1 (a <pad> at the head of the beam
 *               zeroes every input embedding, :263-264), max_steps = 51 (the reference breaks after step 51, :300).
 * greedy      : beam = 1, first_token = 0, lookup_first = 1 (embed(0) is fed, :150,156), zero_pad_rule = 0,
 *               max_steps = max_sentence.
 * Result per image: finished = 1 and the best completed sequence (first maximum of the cumulative
 * log-probability, :311), or finished = 0 and the head of the beam as it stands.  Sequences start with
 * first_token and include end_token; positions >= length are 0. */
typedef struct {
    int n_images, beam, max_steps;
    int zero_pad_rule, lookup_first;
    int64_t first_token, end_token;
} caphn_search_cfg;
size_t caphn_decoder_search_workspace_bytes(const caphn_decoder_dims* d, const caphn_search_cfg* c);
int caphn_decoder_search_begin(const caphn_decoder_dims* d, const caphn_decoder_params* p, const caphn_search_cfg* c,
                               const float* features, void* ws, void* search_ws, caphn_stream_t stream);
/* alphas (optional, beam == 1 only): [n_images, max_steps, P] attention maps of the steps taken. */
int caphn_decoder_search_steps(const caphn_decoder_dims* d, const caphn_decoder_params* p, const caphn_search_cfg* c,
                               int step0, int nsteps, float* alphas, void* ws, void* search_ws, caphn_stream_t stream);
/* seqs [n_images, max_steps+1] int64, lengths/finished [n_images] int32, scores [n_images] f32, n_active [1] int32. */
int caphn_decoder_search_result(const caphn_decoder_dims* d, const caphn_search_cfg* c, int steps_done, void* ws, void* search_ws,
                                int64_t* seqs, int* lengths, float* scores, int* finished, int* n_active,
                                caphn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Non-attention decoders of the older hypernet path: DecoderGRU [later.py:362-457] and DecoderRNN
 * [later.py:227-330] as constructed by hypernet.py:50-53, teacher forced:
 *   x_0 = features[B,E] (image embedding), x_t = embed[caps[:,t-1]];  h = cell(x_t, h);
 *   for each extra layer: h = layer(h, h)  (LSTM: (h,c) = layer(h,(h,c)));  logits[:,t] = fc_out(h).
 * h0 (and c0 for the LSTM) come from the caller: DecoderGRU draws torch.rand (later.py:397), DecoderRNN zeros (:259).
 * Layer 0 has w_ih [NG*H, E]; layers >= 1 have w_ih [NG*H, H]; w_hh [NG*H, H]; gate order as torch's cells.
 * Every pointer may alias another layer's (hypernet.py injects overlapping views of theta, utils.py:68).
 */
#define CAPHN_MAX_LAYERS 4
typedef struct caphn_plain_dims { int B, T, E, H, V, L, cell; } caphn_plain_dims;
typedef struct caphn_plain_params {
    const float* embed_w; const float* out_w; const float* out_b;      /* embed.weight [V,E], fc_out.weight [V,H], .bias [V] */
    const float* w_ih[CAPHN_MAX_LAYERS]; const float* w_hh[CAPHN_MAX_LAYERS];
    const float* b_ih[CAPHN_MAX_LAYERS]; const float* b_hh[CAPHN_MAX_LAYERS];
} caphn_plain_params;
typedef struct caphn_plain_grads {
    float* embed_w; float* out_w; float* out_b;
    float* w_ih[CAPHN_MAX_LAYERS]; float* w_hh[CAPHN_MAX_LAYERS]; float* b_ih[CAPHN_MAX_LAYERS]; float* b_hh[CAPHN_MAX_LAYERS];
    float* features;               /* [B,E] gradient w.r.t. the image embedding (optional) */
} caphn_plain_grads;
size_t caphn_plain_workspace_bytes(const caphn_plain_dims* d);
int caphn_plain_forward(const caphn_plain_dims* d, const caphn_plain_params* p, const float* features,
                        const int64_t* captions, const float* h0, const float* c0, float* logits, void* ws,
                        caphn_stream_t stream);
/* The forward with teacher_forcing = False (later.py:418-431 / :290-301): from step 1 on the input is the embedding of a word
 * drawn from softmax(out_{t-1}) (torch.multinomial(pred, 1)), not of the caption's.  The draw of (b, t) is the inverse CDF at
 * the uniform number the counter-based hash of (seed, b T + t) gives -- the same distribution as torch.multinomial, NOT torch's
 * Philox stream, so the reference's draws cannot be reproduced (parity: given the ids drawn here, everything equals the
 * teacher-forced forward over them).  chosen [B,T] (optional): chosen[b,t] = the id fed at step t (t >= 1; -1 at t = 0).
 * The workspace is left as caphn_plain_forward leaves it with those ids in place of the caption's: caphn_plain_backward
 * applies unchanged (no gradient flows through the draw, as in the reference's autograd graph). */
int caphn_plain_forward_sampled(const caphn_plain_dims* d, const caphn_plain_params* p, const float* features,
                                const float* h0, const float* c0, unsigned long long seed, float* logits,
                                int64_t* chosen, void* ws, caphn_stream_t stream);
/* Needs the workspace as the forward left it.  Gradient buffers are distinct (non-aliasing) arrays. */
int caphn_plain_backward(const caphn_plain_dims* d, const caphn_plain_params* p, const float* features,
                         const int64_t* captions, const float* h0, const float* c0, const float* dlogits,
                         const caphn_plain_grads* g, void* ws, caphn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Stand-alone BahdanauAttention.forward (models/attention.py:21-46) behind its two Linear layers (callers that step a decoder
 * by hand; AttentionGru's own loops have it fused in the recurrent kernels):
 *   Waf [B,P,H] = W_a f + b, uah [B,H] = U_a h + b   ->   alpha [B,P] = softmax_p(v_a . tanh(Waf_p + uah) + b_va),
 *   ctx [B,F] = sum_p alpha_p f_p.
 * Backward: dctx [B,F], dalpha [B,P] (or NULL) -> dWaf [B,P,H], duah [B,H], part [B,H+1] (per-caption partials of d v_a and, in
 * column H, d b_va: the caller sums them over B), df [B,P,F] (or NULL) = alpha_p dctx, the direct path into the features. */
int caphn_bahdanau_fwd(int B, int P, int F, int H, const float* f, const float* Waf, const float* uah, const float* v_a,
                       const float* b_va, float* ctx, float* alpha, caphn_stream_t stream);
int caphn_bahdanau_bwd(int B, int P, int F, int H, const float* f, const float* Waf, const float* uah, const float* v_a,
                       const float* alpha, const float* dctx, const float* dalpha, float* dWaf, float* duah, float* part,
                       float* df, caphn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Loss: F.cross_entropy(logits.view(-1,V), caps.view(-1), ignore_index)   [hypernet_attention.py:183,
 * cc_train_hypernet.py:153].  Writes the mean loss to loss_out[0], the number of non-ignored
 * targets to loss_out[1], and d loss / d logits to dlogits (may alias logits).
 * leave_ignored_rows = 1: d logits rows of ignored targets are left unwritten instead of zero-filled -- for callers
 * whose backward only visits live rows (dims.row_subset); their logits rows may be uninitialised, too.
 * ws: caphn_ce_workspace_bytes(rows).
 */
size_t caphn_ce_workspace_bytes(int rows);
/* The same in two stages, for callers that take the loss value off their critical path: _rows writes d logits (and the
 * per-row losses into ws), _finish reduces them to loss_out.  n_valid_dev (optional, device int): the number of
 * non-ignored targets if the caller has it already (caphn_decoder_rowcount_ptr) -- saves the counting kernel. */
int caphn_cross_entropy_rows(int rows, int V, const float* logits, const int64_t* targets, int64_t ignore_index,
                             float* dlogits, int leave_ignored_rows, const int* n_valid_dev, void* ws, caphn_stream_t stream);
int caphn_cross_entropy_finish(int rows, const int* n_valid_dev, float* loss_out, void* ws, caphn_stream_t stream);
/* caphn_cross_entropy_rows over rows that are `ld` floats apart (ld >= V; caphn_decoder_dims.logits_ld) in logits AND d logits. */
int caphn_cross_entropy_rows_ld(int rows, int V, int ld, const float* logits, const int64_t* targets, int64_t ignore_index,
                                float* dlogits, int leave_ignored_rows, const int* n_valid_dev, void* ws, caphn_stream_t stream);
int caphn_cross_entropy_fwd_bwd(int rows, int V, const float* logits, const int64_t* targets,
                                int64_t ignore_index, float* dlogits, float* loss_out,
                                int leave_ignored_rows, void* ws, caphn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Embedding                                                     [decoderlstm.py:28,62]
 * gather: out[r,:] = (idx[r] < 0) ? 0 : table[idx[r],:]      scatter_add: table_grad[idx[r],:] += g[r,:]
 */
int caphn_embedding_gather(int rows, int E, const float* table, const int64_t* idx, float* out,
                           caphn_stream_t stream);
int caphn_embedding_scatter_add(int rows, int E, const float* g, const int64_t* idx, float* table_grad,
                                caphn_stream_t stream);
/* The same with the table's row count V: what the DETERMINISTIC mode (caphn_tune key 13) scans.  The form without V falls back to
 * the single value given with caphn_tune(13, V), which cannot fit two tables of different sizes in one process. */
int caphn_embedding_scatter_add_v(int rows, int E, int V, const float* g, const int64_t* idx, float* table_grad,
                                  caphn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Optimiser: clip_grad_norm_(5.0) + torch.optim.Adam            [cc_train_hypernet.py:120,405]
 */
/* partial[0..caphn_sumsq_blocks(n)) = block partial sums of squares (double). */
int caphn_sumsq_blocks(size_t n);
int caphn_sumsq_f32(size_t n, const float* x, double* partial, caphn_stream_t stream);
/* || sum_r g_r (x) a_r ||_F^2 = sum_{r,s} (g_r.g_s)(a_r.a_s) for R rank-1 terms; gfac [R,rows] (ldg),
 * afac [R,k] (lda).  Adds the value (double) to acc[0].  ws: R*R*2 doubles. */
int caphn_rank_sumsq_f32(int R, int rows, int k, const float* gfac, size_t ldg, const float* afac, size_t lda,
                         double* acc, double* ws, caphn_stream_t stream);
/* The same for n (gfac, afac) pairs -- one per hypernet head -- in a single launch; host arrays of length n.
 * acc[0] += sum over pairs.  ws: n*R*R*2 doubles. */
int caphn_rank_sumsq_multi_f32(int R, int n, const int* rows, const int* k, const float* const* gfac, const size_t* ldg,
                               const float* const* afac, const size_t* lda, double* acc, double* ws, caphn_stream_t stream);
/* coef_out[0] = scale * min(1, max_norm / (scale * sqrt(sum partial + extra[0]) + 1e-6));
 * coef_out[1] = scale * sqrt(...) (the total norm).  `scale` = 1/world_size. */
int caphn_clip_coef(int nparts, const double* partial, const double* extra, double max_norm, double scale,
                    float* coef_out, caphn_stream_t stream);

/* All of the above in two back-to-back launches (the fused trainers' path): coef_out[0] / [1] as caphn_clip_coef would give
 * for sum(x[0..n)^2) + sum over the njobs (gfac, afac) pairs of || sum_r g_r (x) a_r ||_F^2   (host arrays of length njobs;
 * njobs may be 0).  Fixed summation order (no atomics).  ws: caphn_grad_norm_workspace_bytes(n, R, njobs) bytes. */
size_t caphn_grad_norm_workspace_bytes(size_t n, int R, int njobs);
int caphn_grad_norm_coef(size_t n, const float* x, int R, int njobs, const int* rows, const int* k,
                         const float* const* gfac, const size_t* ldg, const float* const* afac, const size_t* lda,
                         double max_norm, double scale, float* coef_out, void* ws, caphn_stream_t stream);

typedef struct caphn_adam_hparams {
    float lr, beta1, beta2, eps;
    int step;                          /* 1-based, bias corrections computed from it */
    const float* dev_scalars;          /* optional DEVICE pointer to {lr / (1 - beta1^step), sqrt(1 - beta2^step)}:
                                          when non-NULL it overrides lr/step, so a captured hipGraph can be
                                          replayed for every step (the caller refreshes the two floats) */
    int zero_gfac;                     /* rank-1 passes only (caphn_adam_rank*_f32, R == 1): 1 = the pass also CLEARS gfac[0..rows) --
                                          it is the last reader of that row factor; a trainer whose factor is d theta inside its
                                          gradient arena saves the fill in front of the next backward.  0 elsewhere. */
} caphn_adam_hparams;
/* p,m,v,g flat [n]; g is multiplied by coef[0] (device) first. */
int caphn_adam_dense_f32(size_t n, float* p, float* m, float* v, const float* g, const float* coef,
                         const caphn_adam_hparams* hp, caphn_stream_t stream);
/* caphn_grad_norm_coef followed by caphn_adam_dense_f32 on the same flat gradient x = g, with the coefficient finished INSIDE the
 * Adam kernel (two launches instead of three; coef_out is still written for the rank-1 passes behind it).  When ce_rows > 0 the
 * launch also reduces the per-row losses caphn_cross_entropy_rows left in ce_ws to loss_out = {mean loss, n_valid}
 * (= caphn_cross_entropy_finish, which then need not sit between the loss and the backward). */
int caphn_grad_norm_adam_dense(size_t n, float* p, float* m, float* v, const float* g, int R, int njobs, const int* rows,
                               const int* k, const float* const* gfac, const size_t* ldg, const float* const* afac,
                               const size_t* lda, double max_norm, double scale, float* coef_out, void* ws,
                               const caphn_adam_hparams* hp, int ce_rows, const void* ce_ws, const int* ce_n_valid_dev,
                               float* loss_out, caphn_stream_t stream);
/* The same two operations for parameters that live in SEPARATE allocations (a torch.optim.Optimizer over a module's
 * parameter list: cc_train_hypernet.py:110-120 builds Adam over ~30 tensors, Lightning clips their global norm to 5.0, :405).
 * All arrays are HOST arrays of ntensors entries holding device pointers / element counts (they travel as kernel arguments;
 * gradient addresses change every step).
 * caphn_grad_norm_multi: clip_grad_norm_'s coefficient over ntensors dense gradients plus njobs rank-R members given by their
 * factors (arguments as caphn_grad_norm_coef), two launches per 48 tensors; ws: caphn_grad_norm_multi_workspace_bytes.
 * caphn_adam_multi_f32: Adam on every tensor, g multiplied by coef[0] (device) first, one launch per 48 tensors. */
size_t caphn_grad_norm_multi_workspace_bytes(int ntensors, const size_t* n, int R, int njobs);
int caphn_grad_norm_multi(int ntensors, const float* const* g, const size_t* n, int R, int njobs, const int* rows, const int* k,
                          const float* const* gfac, const size_t* ldg, const float* const* afac, const size_t* lda,
                          double max_norm, double scale, float* coef_out, void* ws, caphn_stream_t stream);
int caphn_adam_multi_f32(int ntensors, float* const* p, float* const* m, float* const* v, const float* const* g, const size_t* n,
                         const float* coef, const caphn_adam_hparams* hp, caphn_stream_t stream);
/* W,m,v [rows,k]; gradient = coef[0] * sum_r gfac[r,row] * afac[r,col], never materialised. */
int caphn_adam_rank_f32(int R, int rows, int k, float* W, float* m, float* v,
                        const float* gfac, size_t ldg, const float* afac, size_t lda,
                        const float* coef, const caphn_adam_hparams* hp, caphn_stream_t stream);
/* Same update, and while the updated row W'[row,:] is still in registers also the next forward GEMV
 * next_theta[row] = W'[row,:] . next_a + next_bias[row]   (next_bias must already hold its updated value):
 * the following step's caphn_hyper_forward for this head needs no HBM pass of its own. */
int caphn_adam_rank_gemv_f32(int R, int rows, int k, float* W, float* m, float* v,
                             const float* gfac, size_t ldg, const float* afac, size_t lda,
                             const float* coef, const caphn_adam_hparams* hp,
                             const float* next_a, const float* next_bias, float* next_theta, caphn_stream_t stream);
/* Several rank-R members in ONE launch (the hypernet's small heads -- the two [3H, k] bias heads -- are launch-bound on the optimiser's
 * chain): same update as caphn_adam_rank_f32 / _gemv_f32 per member (next_* all NULL or all set).  Members that do not fit the
 * shared fast path (k % 4, alignment, different width classes) are launched one by one; the result is the same either way. */
typedef struct caphn_rank_job {
    float* W; float* m; float* v;             /* [rows, k] */
    const float* gfac; size_t ldg;            /* [R, rows] row factors, leading dimension */
    const float* afac; size_t lda;            /* [R, k] column factors */
    const float* next_a; const float* next_bias; float* next_theta;
    int rows, k;
    /* optional (with next_*): the member is the generated W_hh [NG H, H] of the decoder and next_theta[row] is ALSO stored at its place
     * in the pair recurrent kernels' packed per-half copy -- address and layout from caphn_decoder_pair_pack_desc -- so that the next
     * forward needs no packing launch between this pass and its recurrent kernel (dims.precomputed bit 128).  NULL = off. */
    float* next_pack; int pack_H, pack_HA, pack_pitch, pack_hrows;
} caphn_rank_job;
int caphn_adam_rank_multi_f32(int R, int njobs, const caphn_rank_job* jobs, const float* coef, const caphn_adam_hparams* hp,
                              caphn_stream_t stream);
/* dst[0..n) = src[0..n) (n % 4 == 0, 16-byte aligned) with the access pattern of the streaming kernels (non-temporal dwordx4): the
 * copy microbenchmark behind bench.py's roofline.copy_ceiling_gbps. */
int caphn_stream_copy_f32(size_t n, const float* src, float* dst, caphn_stream_t stream);
/* dense outer product out[rows,k] = g[rows] (x) a[k]  (module API: torch optimisers want dW2 dense) */
int caphn_outer_f32(int rows, int k, const float* g, const float* a, float* out, caphn_stream_t stream);

/* Tuning knob used by tools/microbench_stream.py to A/B kernel variants in one process
 * (key 0: forward-GEMV variant, key 1: rank-Adam variant, key 2: GEMM back end -- 0 fp32 MFMA, 1 split-bf16 MFMA,
 * key 3: row rotation in the recurrent kernels, key 4: side-stream forking of the decoder composites, key 6: XCD-aware
 * GEMM tile order, key 7: branch-free GEMM loads, key 8: pre-split GEMM operands -- 0 off (every tile split on use), 1 on, key 9: recurrent kernels with two workgroups
 * per caption -- 0 off, 1 on, key 10: timing experiments only, key 11: REDUCED-PRECISION side mode -- 1 = every dense
 * contraction as ONE bf16 product (operands rounded to bf16 at staging, fp32 accumulate; recurrent kernels, softmax, loss,
 * Adam and the master weights stay fp32), 2 = "bf16x2": operands as TWO bf16 planes (hi + mid = 16 significand bits), three products
 * (logits within 1e-4 of the fp32 path at the canonical size), 0 = the fp32-class six-product default, key 12: forced GEMM tile (experiments), key 13: DETERMINISTIC gradients -- value V > 0 (the
 * vocabulary size) turns split-K off in the decoder composites (its partial products are summed with fp32 atomics) and computes
 * the embedding gradient by a destination-major scan of the V table rows instead of atomic scatter-adds: gradients are then
 * bit-identical from run to run, at a cost in speed; 0 = off), key 14: workgroup cap of one rank-1 Adam launch (64..65535, default
 * 4096), key 15: 1 = caphn_decoder_hyper_backward runs the chain to the hypernet VJP on the caller's stream and the
 * attention / feature_fc chain on a side stream, 0 (default) = the other way round; key 4 values: 0 one stream, 1 vocabulary weight
 * gradient after BPTT, 2 beside BPTT, 3 big leaves held back, 4 (default) 2 with the pair recurrent kernels else 1; key 16:
 * 1 (default) = the pair recurrent kernels keep part of [U_a; W_hh] on chip (registers + spare LDS) for the whole kernel, 0 = all
 * rows streamed from L2 every time step; key 17: workgroups a split-K GEMM of the composites aims at (default 1280; 256 / 512 /
 * 1024 / 2048 measured slower); key 18: tile walk of the split-bf16 GEMM inside an XCD (0 n fastest, 1 (default) m fastest when
 * B outgrows the L2 and A is the smaller operand, 2 m fastest always); key 20: bit mask of the composites' side branches in
 * use (default 7 = all three; a cleared branch runs on the caller's stream -- every smaller set measured 10-70 us slower); key 21:
 * with the vocabulary weight gradient beside BPTT (key 4 = 2), 1 (default) starts it after the dHs GEMM, 0 beside it.
 * key 16 also takes 2 (default): the pair FORWARD kernel keeps ALL of a half's [U_a; W_hh] on chip when it fits; key 22: hand-off time
 * bound of the pair kernels in microseconds (default 1 000 000); key 23: bit mask of GEMM layouts (1 NT, 2 NN, 4 TN) that run the
 * 64x64 tile with ping-pong LDS images (default 0: measured slower or equal on every shape of the step); key 24: bit 0 switches
 * the same-XCD hand-off form of the pair kernels off (A/B); key 25: GEMM compiled for 5 / 6 waves per SIMD (0 default: measured
 * slower); key 26: 1 = the composites' cross-stream dependencies as stream memory operations instead of events (0 default:
 * measured slower); key 27: 1 = the tail of caphn_hyper_backward behind the transposed GEMV (reduce, three small
 * transposed GEMVs, the small layers' rank-1 gradients) as ONE launch with counter barriers, 0 (default) = five launches (measured
 * equal in the step); key 28: 1 (default) = hn_base and the heads' first layers of caphn_hyper_forward / _forward_acts in ONE launch
 * (bit-identical results; -10 us per step), 0 = three launches; key 30: workgroups per head of the hypernet VJP's transposed GEMV
 * (64..4096, default 512: no effect measured); key 31: workgroups of the dense arena's Adam launch (default 2048: no effect measured).
 * key 14's default is 16384 since round 3 (measured -18..-23 us per step against 4096); keys 33 / 34 (experiments): split-K of the
 * backward's two live-row vocabulary GEMMs (dHs, dW_fc), 0 (default) = automatic -- every other value measured equal or slower;
 * key 35: 1 = the backward's side branches end into one another (0 default: measured slower); key 36: 1 = the K = 200 NT products
 * with N >= 1024 (the vocabulary logits) through the K-resident kernel (0 default: 1.09x alone, +12 us in the step).
 * Defaults are the measured-fastest. */
int caphn_tune(int key, int value);

/* ---------------------------------------------------------------------------------------
 * N4 (CATR): the non-GEMM pieces of baseline/transformer.py's layers.  Parity: pinned by vectors generated from the
 * reference's own Transformer (tests/golden/catr_*.npz, tools/make_golden.py --only-catr).
 *
 * nn.LayerNorm over the last dimension (baseline/transformer.py:19,27,138-139,199-201,281; biased variance), d <= 1024.
 * mean / rstd [rows] are saved for the backward.  ws: caphn_layernorm_bwd_workspace_bytes(rows, d).
 */
int caphn_layernorm_fwd(int rows, int d, const float* x, const float* gamma, const float* beta, float eps, float* y,
                        float* mean, float* rstd, caphn_stream_t stream);
size_t caphn_layernorm_bwd_workspace_bytes(int rows, int d);
int caphn_layernorm_bwd(int rows, int d, const float* x, const float* gamma, const float* mean, const float* rstd,
                        const float* dy, float* dx, float* dgamma, float* dbeta, void* ws, caphn_stream_t stream);

/* The core of nn.MultiheadAttention (baseline/transformer.py:137,197-199 -> F.multi_head_attention_forward):
 *   o[t,b,h,:] = sum_j softmax_j( q[t,b,h,:].k[j,b,h,:] * scale + attn_mask[t,j] + (key_padding[b,j] ? -inf : 0) ) v[j,b,h,:]
 * for nh heads of width dh (<= 64).  Tensors are addressed as  base + t*ldt + b*ldb + h*dh  (floats), so the sequence-first
 * [T, bs, nh*dh] tensors of the reference and slices of a packed in-projection are used in place; o / d_o share o's strides,
 * dq / dk / dv those of q / k / v.  attn_mask: additive [tq, tk] (-inf = masked) or NULL; key_padding: [bs, tk] bytes,
 * non-zero = ignore, or NULL.  lse [bs*nh, tq] (log-sum-exp per row) links forward and backward.  A row with every key masked
 * yields zeros (torch yields NaN).  Limits: caphn_attention_supported() (the K/V or Q/dO side of one (batch, head) must fit LDS:
 * about 440 positions at dh <= 32, 200 at dh <= 64).  */
typedef struct {
    int bs, nh, dh, tq, tk;
    int q_ldt, q_ldb, k_ldt, k_ldb, v_ldt, v_ldb, o_ldt, o_ldb;
    float scale;
    float dropout_p;               /* dropout on the attention probabilities (nn.MultiheadAttention(dropout=p) in training mode); 0 = off */
    unsigned long long seed;       /* mask of probability (b*nh + h, t, j): caphn_dropout_f32's hash at index ((b*nh + h)*tq + t)*tk + j */
} caphn_attn_dims;
int caphn_attention_supported(const caphn_attn_dims* d);
int caphn_attention_fwd(const caphn_attn_dims* d, const float* q, const float* k, const float* v, const float* attn_mask,
                        const unsigned char* key_padding, float* o, float* lse, caphn_stream_t stream);
int caphn_attention_bwd(const caphn_attn_dims* d, const float* q, const float* k, const float* v, const float* attn_mask,
                        const unsigned char* key_padding, const float* o, const float* lse, const float* d_o,
                        float* dq, float* dk, float* dv, caphn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CAPHN_H */
