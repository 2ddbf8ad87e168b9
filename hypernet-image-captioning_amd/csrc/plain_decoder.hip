// Non-attention multi-layer decoders of the reference's older hypernet path: DecoderGRU (later.py:362-457) and
// DecoderRNN (later.py:227-330) as used by hypernet.py:50-53, teacher forced.
//
//   x_0 = features (the image embedding), x_t = embed[caps[:, t-1]] for t >= 1      later.py:411-420 / :279-288
//   h = cell(x_t, h);  for layer in layers: h = layer(h, h)   (LSTM: (h, c) = layer(h, (h, c)))
//   out_t = fc_out(h)                                                                  :441 / :311
//
// In that configuration the hypernet is 2.8 G parameters (11 GB fp32) and the decoder is a rounding error next to
// streaming it, so the recurrence is NOT a persistent kernel here: every step is a small MFMA GEMM (h W^T for the
// whole batch) plus a fused gate kernel, all on one stream (~5 launches per step and direction); the input-side
// GEMM, the vocabulary projection and every weight gradient are batched over all T steps.
// Internal arrays are time-major [T][B][.] so that a step's rows are contiguous.
#include "common.h"
#include "gemm_internal.h"
#include <algorithm>

#define RUN(x) do { int _rc = (x); if (_rc != CAPHN_OK) return _rc; } while (0)

namespace {

struct PWs {    // float offsets
    size_t X, Xg, idx, Hbt, dHbt, dX, carry, dcar, tmp, colws;
    size_t gi[CAPHN_MAX_LAYERS], gh[CAPHN_MAX_LAYERS], gates[CAPHN_MAX_LAYERS], hn[CAPHN_MAX_LAYERS], Hl[CAPHN_MAX_LAYERS];
    size_t Cl[CAPHN_MAX_LAYERS], dgi[CAPHN_MAX_LAYERS], dgh[CAPHN_MAX_LAYERS];
    size_t total;
    int NG;
};
inline size_t up4(size_t v) { return (v + 3) & ~(size_t)3; }
inline bool pdims_ok(const caphn_plain_dims* d) {
    return d && d->B > 0 && d->T > 0 && d->E > 0 && d->H > 0 && d->V > 0 && d->L >= 1 && d->L <= CAPHN_MAX_LAYERS &&
           (d->cell == CAPHN_CELL_GRU || d->cell == CAPHN_CELL_LSTM);
}
inline PWs playout(const caphn_plain_dims* d) {
    PWs w;
    const size_t B = d->B, T = d->T, E = d->E, H = d->H, V = d->V;
    const bool lstm = d->cell == CAPHN_CELL_LSTM;
    const size_t NG = lstm ? 4 : 3, GH = NG * H;
    w.NG = (int)NG;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += up4(n); return r; };
    w.X = take(T * B * E); w.Xg = take(T * B * GH); w.idx = take(2 * T * B); w.Hbt = take(B * T * H); w.dHbt = take(B * T * H);
    w.dX = take(T * B * E); w.carry = take(B * H); w.dcar = take(lstm ? B * H : 0); w.tmp = take(B * H);
    for (int l = 0; l < d->L; ++l) {
        w.gi[l] = l == 0 ? w.Xg : take(T * B * GH);
        w.gh[l] = take(lstm ? 0 : T * B * GH);
        w.gates[l] = take(T * B * GH); w.hn[l] = take(lstm ? 0 : T * B * H); w.Hl[l] = take(T * B * H);
        w.Cl[l] = take(lstm ? T * B * H : 0);
        w.dgi[l] = take(T * B * GH); w.dgh[l] = lstm ? w.dgi[l] : take(T * B * GH);
    }
    size_t cs = std::max(caphn_colsum_workspace_bytes((int)(B * T), (int)V), caphn_colsum_workspace_bytes((int)(B * T), (int)GH)) / sizeof(float);
    w.colws = take(cs);
    w.total = o;
    return w;
}

// x rows in time-major order: t = 0 the image embedding, t >= 1 the previous caption token's embedding
__global__ __launch_bounds__(256) void plain_inputs_kernel(int B, int T, int E, const float* __restrict__ feats, const float* __restrict__ table,
                                                          const int64_t* __restrict__ caps, float* __restrict__ X, int64_t* __restrict__ idx) {
    const int row = blockIdx.x, t = row / B, b = row - t * B;
    const int64_t id = t == 0 ? (int64_t)-1 : caps[(size_t)b * T + t - 1];
    if (threadIdx.x == 0) idx[row] = id;
    const float* src = t == 0 ? feats + (size_t)b * E : table + (size_t)id * E;
    for (int e = threadIdx.x; e < E; e += 256) X[(size_t)row * E + e] = src[e];
}

// nn.GRUCell gate arithmetic for one step of one layer; h also lands in the batch-major copy when hbt != null
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(int B, int H, int T, int t, const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ hprev, float* __restrict__ h, float* __restrict__ gates,
                                                           float* __restrict__ hn, float* __restrict__ hbt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const size_t g0 = (size_t)b * 3 * H + j;
    const float r = caphn_sigmoid(gi[g0] + gh[g0]);
    const float z = caphn_sigmoid(gi[g0 + H] + gh[g0 + H]);
    const float hnv = gh[g0 + 2 * H];
    const float n = caphn_tanh(gi[g0 + 2 * H] + r * hnv);
    const float hp = hprev[i];
    const float hv = (1.f - z) * n + z * hp;
    h[i] = hv; gates[g0] = r; gates[g0 + H] = z; gates[g0 + 2 * H] = n; hn[i] = hnv;
    if (hbt) hbt[((size_t)b * T + t) * H + j] = hv;
}
// dh = dh_a[b, t] (batch-major, optional) + dh_b (optional); outputs dgi, dgh and the direct path dh * z
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(int B, int H, int T, int t, const float* __restrict__ dh_a, const float* __restrict__ dh_b,
                                                           const float* __restrict__ gates, const float* __restrict__ hn, const float* __restrict__ hprev,
                                                           float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dh_direct) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const size_t g0 = (size_t)b * 3 * H + j;
    float dh = 0.f;
    if (dh_a) dh += dh_a[((size_t)b * T + t) * H + j];
    if (dh_b) dh += dh_b[i];
    const float r = gates[g0], z = gates[g0 + H], n = gates[g0 + 2 * H];
    const float dn = dh * (1.f - z) * (1.f - n * n);
    const float dz = dh * (hprev[i] - n) * z * (1.f - z);
    const float dr = dn * hn[i] * r * (1.f - r);
    dgi[g0] = dr; dgi[g0 + H] = dz; dgi[g0 + 2 * H] = dn;
    dgh[g0] = dr; dgh[g0 + H] = dz; dgh[g0 + 2 * H] = dn * r;
    dh_direct[i] = dh * z;
}
// nn.LSTMCell: pre = x side + h side pre-activations (both biases included), gate order i,f,g,o
__global__ __launch_bounds__(256) void lstm_gates_fwd_kernel(int B, int H, int T, int t, const float* __restrict__ pre,
                                                            const float* __restrict__ cprev, float* __restrict__ h, float* __restrict__ c,
                                                            float* __restrict__ gates, float* __restrict__ hbt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const size_t g0 = (size_t)b * 4 * H + j;
    const float ig = caphn_sigmoid(pre[g0]);
    const float fg = caphn_sigmoid(pre[g0 + H]);
    const float gg = caphn_tanh(pre[g0 + 2 * H]);
    const float og = caphn_sigmoid(pre[g0 + 3 * H]);
    const float cv = fg * cprev[i] + ig * gg;
    const float hv = og * caphn_tanh(cv);
    c[i] = cv; h[i] = hv;
    gates[g0] = ig; gates[g0 + H] = fg; gates[g0 + 2 * H] = gg; gates[g0 + 3 * H] = og;
    if (hbt) hbt[((size_t)b * T + t) * H + j] = hv;
}
// dc (in/out) carries the cell-state gradient backwards through the (t, layer) evaluation order
__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(int B, int H, int T, int t, const float* __restrict__ dh_a, const float* __restrict__ dh_b,
                                                            const float* __restrict__ gates, const float* __restrict__ c, const float* __restrict__ cprev,
                                                            float* __restrict__ dgates, float* __restrict__ dc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const size_t g0 = (size_t)b * 4 * H + j;
    float dh = 0.f;
    if (dh_a) dh += dh_a[((size_t)b * T + t) * H + j];
    if (dh_b) dh += dh_b[i];
    const float ig = gates[g0], fg = gates[g0 + H], gg = gates[g0 + 2 * H], og = gates[g0 + 3 * H];
    const float tc = caphn_tanh(c[i]);
    const float dcv = dc[i] + dh * og * (1.f - tc * tc);
    dgates[g0] = dcv * gg * ig * (1.f - ig);
    dgates[g0 + H] = dcv * cprev[i] * fg * (1.f - fg);
    dgates[g0 + 2 * H] = dcv * ig * (1.f - gg * gg);
    dgates[g0 + 3 * H] = dh * tc * og * (1.f - og);
    dc[i] = dcv * fg;
}

// Step t >= 1 of the sampled forward: one workgroup per caption draws id ~ softmax(logits[b, t-1, :]) by inverse CDF and copies
// that word's embedding into the step's x row.  Thread i owns the contiguous chunk [i c, (i + 1) c) of the vocabulary.
__global__ __launch_bounds__(256) void sample_word_kernel(int B, int T, int V, int E, int t, const float* __restrict__ logits,
                                                         unsigned long long seed, const float* __restrict__ table,
                                                         float* __restrict__ X, int64_t* __restrict__ idx, int64_t* __restrict__ chosen) {
    __shared__ float red[256];
    __shared__ float pre[257];
    __shared__ int pick;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + ((size_t)b * T + (t - 1)) * V;
    const int c = (V + 255) / 256, v0 = min(tid * c, V), v1 = min(v0 + c, V);
    float mx = -INFINITY;
    for (int v = v0; v < v1; ++v) mx = fmaxf(mx, row[v]);
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sl = 0.f;
    for (int v = v0; v < v1; ++v) sl += caphn_exp(row[v] - mx);
    red[tid] = sl;
    if (tid == 0) pick = -1;
    __syncthreads();
    if (tid == 0) {                 // exclusive prefix over the 256 chunk sums, in order (deterministic)
        float acc = 0.f;
        for (int i = 0; i < 256; ++i) { pre[i] = acc; acc += red[i]; }
        pre[256] = acc;
    }
    __syncthreads();
    // u in [0, 1): the 24 high bits of splitmix64(seed, b T + t)
    unsigned long long z = seed + ((unsigned long long)b * T + t) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float target = (float)(z >> 40) * (1.0f / 16777216.0f) * pre[256];
    if (v0 < v1 && pre[tid] <= target && target < pre[tid + 1]) {
        float acc = pre[tid];
        int sel = v1 - 1;
        for (int v = v0; v < v1; ++v) { acc += caphn_exp(row[v] - mx); if (acc > target) { sel = v; break; } }
        pick = sel;
    }
    __syncthreads();
    int id = pick;
    if (id < 0) {                   // target landed on the total by rounding: the last word with a non-zero weight
        id = V - 1;
    }
    if (tid == 0) {
        idx[(size_t)t * B + b] = id;
        if (chosen) chosen[(size_t)b * T + t] = id;
    }
    const float* src = table + (size_t)id * E;
    float* dst = X + ((size_t)t * B + b) * E;
    for (int e = tid; e < E; e += 256) dst[e] = src[e];
}
__global__ __launch_bounds__(256) void plain_first_input_kernel(int B, int T, int E, const float* __restrict__ feats, float* __restrict__ X,
                                                               int64_t* __restrict__ idx, int64_t* __restrict__ chosen) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { idx[b] = -1; if (chosen) chosen[(size_t)b * T] = -1; }
    for (int e = threadIdx.x; e < E; e += 256) X[(size_t)b * E + e] = feats[(size_t)b * E + e];
}

inline int pick_sk(int M, int N, int K) {
    const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    long s = (512 + t64 - 1) / t64;
    const int nslab = (K + 31) / 32;
    if (s > nslab / 4) s = nslab / 4;
    if (s < 1) s = 1;
    if (s > 32) s = 32;
    return (int)s;
}
// C (+)= A^T B over a long K with split-K (fp32 atomics): zero-fill first unless accumulating
inline int wgrad(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, bool accum, hipStream_t s) {
    const int sk = pick_sk(M, N, K);
    if (sk > 1) {       // split-K adds atomically onto C: accumulating = not clearing it first
        if (!accum && hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * N, s) != hipSuccess) return CAPHN_ELAUNCH;
        return caphn_gemm_f32(1, 0, M, N, K, A, lda, B, ldb, C, N, nullptr, nullptr, 0, 0, sk, s);
    }
    return caphn_gemm_f32(1, 0, M, N, K, A, lda, B, ldb, C, N, nullptr, nullptr, 0, accum ? CAPHN_GEMM_ACCUM : 0, 1, s);
}

}  // namespace

static int plain_cells_step(const caphn_plain_dims* d, const caphn_plain_params* p, const PWs& w, float* ws, int t,
                            const float* h0, const float* c0, hipStream_t s);

extern "C" size_t caphn_plain_workspace_bytes(const caphn_plain_dims* d) {
    if (!pdims_ok(d)) return 0;
    return playout(d).total * sizeof(float);
}

extern "C" int caphn_plain_forward(const caphn_plain_dims* d, const caphn_plain_params* p, const float* features,
                                   const int64_t* captions, const float* h0, const float* c0, float* logits, void* ws_,
                                   caphn_stream_t stream) {
    if (!pdims_ok(d) || !p || !features || !captions || !h0 || !logits || !ws_) return CAPHN_EINVAL;
    const bool lstm = d->cell == CAPHN_CELL_LSTM;
    if (lstm && !c0) return CAPHN_EINVAL;
    if (!p->embed_w || !p->out_w || !p->out_b) return CAPHN_EINVAL;
    for (int l = 0; l < d->L; ++l) if (!p->w_ih[l] || !p->w_hh[l] || !p->b_ih[l] || !p->b_hh[l]) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const PWs w = playout(d);
    float* ws = static_cast<float*>(ws_);
    const int B = d->B, T = d->T, E = d->E, H = d->H, V = d->V, L = d->L, GH = w.NG * H, TB = T * B;
    const int nb = (B * H + 255) / 256;
    hipLaunchKernelGGL(plain_inputs_kernel, dim3(TB), dim3(256), 0, s, B, T, E, features, p->embed_w, captions, ws + w.X,
                       reinterpret_cast<int64_t*>(ws + w.idx));
    // x side of layer 0 for every step at once
    RUN(caphn_gemm_f32(0, 1, TB, GH, E, ws + w.X, E, p->w_ih[0], E, ws + w.Xg, GH, p->b_ih[0], nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
    for (int t = 0; t < T; ++t) RUN(plain_cells_step(d, p, w, ws, t, h0, c0, s));
    // vocabulary projection for all (b, t)          later.py:441 / :311
    RUN(caphn_gemm_f32(0, 1, B * T, V, H, ws + w.Hbt, H, p->out_w, H, logits, V, p->out_b, nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
    (void)nb; (void)L;
    return caphn_launch_status();
}

extern "C" int caphn_plain_forward_sampled(const caphn_plain_dims* d, const caphn_plain_params* p, const float* features,
                                           const float* h0, const float* c0, unsigned long long seed, float* logits,
                                           int64_t* chosen, void* ws_, caphn_stream_t stream) {
    if (!pdims_ok(d) || !p || !features || !h0 || !logits || !ws_) return CAPHN_EINVAL;
    const bool lstm = d->cell == CAPHN_CELL_LSTM;
    if (lstm && !c0) return CAPHN_EINVAL;
    if (!p->embed_w || !p->out_w || !p->out_b) return CAPHN_EINVAL;
    for (int l = 0; l < d->L; ++l) if (!p->w_ih[l] || !p->w_hh[l] || !p->b_ih[l] || !p->b_hh[l]) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const PWs w = playout(d);
    float* ws = static_cast<float*>(ws_);
    const int B = d->B, T = d->T, E = d->E, H = d->H, V = d->V, L = d->L, GH = w.NG * H;
    int64_t* idx = reinterpret_cast<int64_t*>(ws + w.idx);
    for (int t = 0; t < T; ++t) {
        // x_t: the image embedding, then the embedding of a word drawn from softmax(out_{t-1})          later.py:418-426
        if (t == 0) hipLaunchKernelGGL(plain_first_input_kernel, dim3(B), dim3(256), 0, s, B, T, E, features, ws + w.X, idx, chosen);
        else hipLaunchKernelGGL(sample_word_kernel, dim3(B), dim3(256), 0, s, B, T, V, E, t, logits, seed, p->embed_w, ws + w.X, idx, chosen);
        RUN(caphn_gemm_f32(0, 1, B, GH, E, ws + w.X + (size_t)t * B * E, E, p->w_ih[0], E, ws + w.Xg + (size_t)t * B * GH, GH, p->b_ih[0],
                           nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
        RUN(plain_cells_step(d, p, w, ws, t, h0, c0, s));
        // out_t = fc_out(h_t): rows (b, t) of the [B,T,V] logits, read by the next step's draw
        RUN(caphn_gemm_f32(0, 1, B, V, H, ws + w.Hl[L - 1] + (size_t)t * B * H, H, p->out_w, H, logits + (size_t)t * V, T * V, p->out_b,
                           nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
    }
    return caphn_launch_status();
}

// every cell of step t (layer 0 from the x-side pre-activations in Xg, the extra layers on their own input)
static int plain_cells_step(const caphn_plain_dims* d, const caphn_plain_params* p, const PWs& w, float* ws, int t,
                            const float* h0, const float* c0, hipStream_t s) {
    const bool lstm = d->cell == CAPHN_CELL_LSTM;
    const int B = d->B, T = d->T, H = d->H, L = d->L, GH = w.NG * H;
    const int nb = (B * H + 255) / 256;
    {
        const size_t oG = (size_t)t * B * GH, oH = (size_t)t * B * H;
        for (int l = 0; l < L; ++l) {
            // layer 0: h = cell(x_t, h_prev); layer l >= 1: h = layer(h, h)          later.py:413-416
            const float* hin = l > 0 ? ws + w.Hl[l - 1] + oH : (t == 0 ? h0 : ws + w.Hl[L - 1] + oH - (size_t)B * H);
            const float* cin = !lstm ? nullptr : (l > 0 ? ws + w.Cl[l - 1] + oH : (t == 0 ? c0 : ws + w.Cl[L - 1] + oH - (size_t)B * H));
            if (l > 0)
                RUN(caphn_gemm_f32(0, 1, B, GH, H, hin, H, p->w_ih[l], H, ws + w.gi[l] + oG, GH, p->b_ih[l], nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
            float* hbt = l == L - 1 ? ws + w.Hbt : nullptr;
            if (!lstm) {
                RUN(caphn_gemm_f32(0, 1, B, GH, H, hin, H, p->w_hh[l], H, ws + w.gh[l] + oG, GH, p->b_hh[l], nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
                hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(nb), dim3(256), 0, s, B, H, T, t, ws + w.gi[l] + oG, ws + w.gh[l] + oG, hin,
                                   ws + w.Hl[l] + oH, ws + w.gates[l] + oG, ws + w.hn[l] + oH, hbt);
            } else {
                // pre-activations = gi + gh: accumulate the h side onto the x side in place
                RUN(caphn_gemm_f32(0, 1, B, GH, H, hin, H, p->w_hh[l], H, ws + w.gi[l] + oG, GH, p->b_hh[l], nullptr, 0,
                                   CAPHN_GEMM_BIAS | CAPHN_GEMM_ACCUM, 1, s));
                hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3(nb), dim3(256), 0, s, B, H, T, t, ws + w.gi[l] + oG, cin,
                                   ws + w.Hl[l] + oH, ws + w.Cl[l] + oH, ws + w.gates[l] + oG, hbt);
            }
        }
    }
    return CAPHN_OK;
}

extern "C" int caphn_plain_backward(const caphn_plain_dims* d, const caphn_plain_params* p, const float* features,
                                    const int64_t* captions, const float* h0, const float* c0, const float* dlogits,
                                    const caphn_plain_grads* g, void* ws_, caphn_stream_t stream) {
    if (!pdims_ok(d) || !p || !features || !captions || !h0 || !dlogits || !g || !ws_) return CAPHN_EINVAL;
    const bool lstm = d->cell == CAPHN_CELL_LSTM;
    if (lstm && !c0) return CAPHN_EINVAL;
    if (!g->out_w || !g->out_b || !g->embed_w) return CAPHN_EINVAL;
    for (int l = 0; l < d->L; ++l) if (!g->w_ih[l] || !g->w_hh[l] || !g->b_ih[l] || !g->b_hh[l]) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const PWs w = playout(d);
    float* ws = static_cast<float*>(ws_);
    const int B = d->B, T = d->T, E = d->E, H = d->H, V = d->V, L = d->L, GH = w.NG * H, TB = T * B, BT = B * T;
    const int nb = (B * H + 255) / 256;
    void* cws = ws + w.colws;

    // vocabulary projection: dW_fc = dlogits^T H, db_fc = colsum, dH = dlogits W_fc   (rows batch-major)
    RUN(wgrad(V, H, BT, dlogits, V, ws + w.Hbt, H, g->out_w, false, s));
    RUN(caphn_colsum_f32(BT, V, dlogits, V, g->out_b, cws, s));
    RUN(caphn_gemm_f32(0, 0, BT, H, V, dlogits, V, p->out_w, H, ws + w.dHbt, H, nullptr, nullptr, 0, 0, 1, s));

    // BPTT over (t, layer) in reverse evaluation order; carry = gradient reaching h_{t-1, last layer}
    if (hipMemsetAsync(ws + w.carry, 0, sizeof(float) * (size_t)B * H, s) != hipSuccess) return CAPHN_ELAUNCH;
    if (lstm && hipMemsetAsync(ws + w.dcar, 0, sizeof(float) * (size_t)B * H, s) != hipSuccess) return CAPHN_ELAUNCH;
    // dst += dgates W: [B, GH] x [GH, H] is a handful of tiles with a 15-50 slab K loop on the BPTT critical path (2 per
    // step and layer); split-K spreads the loop over more workgroups and its atomics are exactly the accumulation
    const int bptt_sk = std::max(1, std::min(8, ((GH + 31) / 32) / 3));
    auto accum_gemm = [&](const float* dg, const float* W, float* dst) {
        return caphn_gemm_f32(0, 0, B, H, GH, dg, GH, W, H, dst, H, nullptr, nullptr, 0, bptt_sk > 1 ? 0 : CAPHN_GEMM_ACCUM, bptt_sk, s);
    };
    for (int t = T - 1; t >= 0; --t) {
        const size_t oG = (size_t)t * B * GH, oH = (size_t)t * B * H;
        for (int l = L - 1; l >= 0; --l) {
            const float* hin = l > 0 ? ws + w.Hl[l - 1] + oH : (t == 0 ? h0 : ws + w.Hl[L - 1] + oH - (size_t)B * H);
            const float* cin = !lstm ? nullptr : (l > 0 ? ws + w.Cl[l - 1] + oH : (t == 0 ? c0 : ws + w.Cl[L - 1] + oH - (size_t)B * H));
            // gradient arriving at this cell's output: the top layer gets the projection's plus next step's carry,
            // lower layers get what the layer above sent down (in tmp)
            const float* dh_a = l == L - 1 ? ws + w.dHbt : nullptr;
            const float* dh_b = l == L - 1 ? ws + w.carry : ws + w.tmp;
            float* dst = l == 0 ? ws + w.carry : ws + w.tmp;      // gradient for the cell evaluated just before this one
            if (!lstm) {
                // dst may alias dh_b: the kernel reads dh_b[i] before writing dst[i] (same thread, same element)
                hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(nb), dim3(256), 0, s, B, H, T, t, dh_a, dh_b, ws + w.gates[l] + oG,
                                   ws + w.hn[l] + oH, hin, ws + w.dgi[l] + oG, ws + w.dgh[l] + oG, dst);
                RUN(accum_gemm(ws + w.dgh[l] + oG, p->w_hh[l], dst));
                if (l > 0)     // the layer's input is the same h: add the x-side path
                    RUN(accum_gemm(ws + w.dgi[l] + oG, p->w_ih[l], dst));
            } else {
                hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(nb), dim3(256), 0, s, B, H, T, t, dh_a, dh_b, ws + w.gates[l] + oG,
                                   ws + w.Cl[l] + oH, cin, ws + w.dgi[l] + oG, ws + w.dcar);
                RUN(caphn_gemm_f32(0, 0, B, H, GH, ws + w.dgi[l] + oG, GH, p->w_hh[l], H, dst, H, nullptr, nullptr, 0, 0, 1, s));
                if (l > 0)
                    RUN(accum_gemm(ws + w.dgi[l] + oG, p->w_ih[l], dst));
            }
        }
    }
    // weight gradients, batched over all steps
    for (int l = 0; l < L; ++l) {
        const float* dgi = ws + w.dgi[l];
        const float* dgh = ws + w.dgh[l];
        RUN(caphn_colsum_f32(TB, GH, dgi, GH, g->b_ih[l], cws, s));
        RUN(caphn_colsum_f32(TB, GH, dgh, GH, g->b_hh[l], cws, s));
        if (l == 0) {
            RUN(wgrad(GH, E, TB, dgi, GH, ws + w.X, E, g->w_ih[0], false, s));
            // h input of layer 0: h0 at t = 0, the last layer's output of step t-1 afterwards
            RUN(wgrad(GH, H, B, dgh, GH, h0, H, g->w_hh[0], false, s));
            if (T > 1)
                RUN(wgrad(GH, H, (T - 1) * B, dgh + (size_t)B * GH, GH, ws + w.Hl[L - 1], H, g->w_hh[0], true, s));
        } else {
            RUN(wgrad(GH, H, TB, dgi, GH, ws + w.Hl[l - 1], H, g->w_ih[l], false, s));
            RUN(wgrad(GH, H, TB, dgh, GH, ws + w.Hl[l - 1], H, g->w_hh[l], false, s));
        }
    }
    // inputs: dX = dXg W_ih ; rows of step 0 are the image embedding's gradient, the rest go to the embedding table
    RUN(caphn_gemm_f32(0, 0, TB, E, GH, ws + w.dgi[0], GH, p->w_ih[0], E, ws + w.dX, E, nullptr, nullptr, 0, 0, 1, s));
    if (g->features)
        if (hipMemcpyAsync(g->features, ws + w.dX, sizeof(float) * (size_t)B * E, hipMemcpyDeviceToDevice, s) != hipSuccess) return CAPHN_ELAUNCH;
    if (hipMemsetAsync(g->embed_w, 0, sizeof(float) * (size_t)V * E, s) != hipSuccess) return CAPHN_ELAUNCH;
    RUN(caphn_embedding_scatter_add_v(TB, E, d->V, ws + w.dX, reinterpret_cast<const int64_t*>(ws + w.idx), g->embed_w, s));
    return caphn_launch_status();
}
