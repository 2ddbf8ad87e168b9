// Loss, bias-gradient column sums, embedding gather/scatter, gradient-norm clipping and Adam.
// All of these are HBM-bound streams: 16-byte coalesced accesses, grid-stride, no MFMA.
#include "common.h"
#include "gemm_internal.h"
#include <math.h>

namespace {
constexpr int RMAX = 8;      // data-parallel ranks a rank-R gradient may have (8 GPUs per node)


// ------------------------------------------------------------------ cross entropy (fwd + bwd)
// F.cross_entropy(logits.view(-1,V), caps.view(-1), ignore_index)   hypernet_attention.py:183
// One workgroup per row: max, sum-exp, then d logits = (softmax - onehot) / n_valid.
__global__ __launch_bounds__(256) void ce_count_kernel(int rows, const int64_t* __restrict__ tgt, int64_t ignore, float* nvalid) {
    __shared__ int cnt[4];
    int c = 0;
    for (int i = threadIdx.x; i < rows; i += 256) c += (tgt[i] != ignore);
    float cf = wave_sum((float)c);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = (int)cf;
    __syncthreads();
    if (threadIdx.x == 0) nvalid[0] = (float)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
}

__global__ __launch_bounds__(256) void ce_row_kernel(int V, int ld, const float* logits, const int64_t* __restrict__ tgt,
                                                     int64_t ignore, const float* __restrict__ nvalid, const int* __restrict__ nvi,
                                                     float* dlogits, float* __restrict__ row_loss, int vec) {
    __shared__ float red[4];
    __shared__ float bc[2];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (size_t)row * ld;
    float* dx = dlogits + (size_t)row * ld;
    const int64_t t = tgt[row];
    if (t == ignore) {                       // ignored rows contribute neither loss nor gradient
        for (int i = tid; i < V; i += 256) dx[i] = 0.f;
        if (tid == 0) row_loss[row] = 0.f;
        return;
    }
    __shared__ float xt_s;                   // read before anybody overwrites (dlogits may alias logits): see ce_row_reg_kernel
    if (tid == 0) xt_s = x[t];
    float m = -INFINITY;
    if (vec) { const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        for (int i = tid; i < (V >> 2); i += 256) { f32x4 v = x4[i]; m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3])); } }
    else for (int i = tid; i < V; i += 256) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) bc[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    m = bc[0];
    float s = 0.f;
    if (vec) { const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        for (int i = tid; i < (V >> 2); i += 256) { f32x4 v = x4[i];
            s += caphn_exp(v[0] - m) + caphn_exp(v[1] - m) + caphn_exp(v[2] - m) + caphn_exp(v[3] - m); } }
    else for (int i = tid; i < V; i += 256) s += caphn_exp(x[i] - m);
    s = wave_sum(s);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) bc[1] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    s = bc[1];
    const float inv = 1.0f / s, scale = 1.0f / (nvi ? (float)nvi[0] : nvalid[0]);
    if (vec) { const f32x4* x4 = reinterpret_cast<const f32x4*>(x); f32x4* d4 = reinterpret_cast<f32x4*>(dx);
        for (int i = tid; i < (V >> 2); i += 256) { f32x4 v = x4[i], o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (caphn_exp(v[e] - m) * inv - ((int64_t)(4 * i + e) == t ? 1.f : 0.f)) * scale;
            d4[i] = o; } }
    else for (int i = tid; i < V; i += 256) dx[i] = (caphn_exp(x[i] - m) * inv - ((int64_t)i == t ? 1.f : 0.f)) * scale;
    if (tid == 0) row_loss[row] = (logf(s) + m - xt_s);
}

// One-pass variant for V % 4 == 0, V <= 4 * 256 * NV: the row lives in registers (NV dwordx4 per thread), so it is read
// once and every exponential is evaluated once (the three-pass kernel above re-reads the 38 KB row from L2 twice and
// evaluates 2 V exponentials).  skip_ignored: with the live-row map nobody reads the d logits rows of ignored targets.
template <int NV>
__global__ __launch_bounds__(256) void ce_row_reg_kernel(int V, int ld, const float* logits, const int64_t* __restrict__ tgt,
                                                         int64_t ignore, const float* __restrict__ nvalid, const int* __restrict__ nvi,
                                                         float* dlogits, float* __restrict__ row_loss, int skip_ignored) {
    __shared__ float red[4];
    __shared__ float bc[2];
    const int row = blockIdx.x, tid = threadIdx.x, V4 = V >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(logits + (size_t)row * ld);
    f32x4* d4 = reinterpret_cast<f32x4*>(dlogits + (size_t)row * ld);
    const int64_t t = tgt[row];
    if (t == ignore) {
        if (!skip_ignored) for (int i = tid; i < V4; i += 256) d4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid == 0) row_loss[row] = 0.f;
        return;
    }
    // d logits may alias logits: the target's logit must be IN HAND before any thread of this row writes.  Parking it in
    // LDS in front of the first barrier forces the load to complete there; as a plain register value the compiler may
    // delay the load to its use after the row has been overwritten by other waves (seen: the reported loss off by 0.2 / n)
    __shared__ float xt_s;
    if (tid == 0) xt_s = logits[(size_t)row * ld + t];
    f32x4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = tid + 256 * j;
        v[j] = i < V4 ? x4[i] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        m = fmaxf(fmaxf(m, fmaxf(v[j][0], v[j][1])), fmaxf(v[j][2], v[j][3]));
    }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) bc[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    m = bc[0];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = caphn_exp(v[j][e] - m);        // exp(-inf) = 0 for the padding lanes
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    s = wave_sum(s);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) bc[1] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    s = bc[1];
    const float inv = 1.0f / s, scale = 1.0f / (nvi ? (float)nvi[0] : nvalid[0]);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = tid + 256 * j;
        if (i < V4) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[j][e] * inv - ((int64_t)(4 * i + e) == t ? 1.f : 0.f)) * scale;
            d4[i] = o;
        }
    }
    if (tid == 0) row_loss[row] = (logf(s) + m - xt_s);
}

__global__ __launch_bounds__(256) void ce_finish_kernel(int rows, const float* __restrict__ row_loss, const float* nvalid,
                                                        const int* __restrict__ nvi, float* out) {
    __shared__ double red[4];
    double s = 0.0;
    // all of a thread's loads are requested before the first is added (one latency instead of rows / 256 of them: the kernel
    // sits on the training step's critical path)
    for (int i0 = threadIdx.x; i0 < rows; i0 += 256 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = i0 + 256 * u; v[u] = i < rows ? row_loss[i] : 0.f; }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = nvi ? (float)nvi[0] : nvalid[0];
        out[0] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n); out[1] = n;
    }
}

// ------------------------------------------------------------------ zero fill
// hipMemsetAsync's fill kernel moves ~370 GB/s (20 us for a 7.7 MB gradient); this one is a plain dwordx4 store stream
__global__ __launch_bounds__(256) void zero_kernel(float* __restrict__ p, size_t n, int vec) {
    const size_t stride = (size_t)gridDim.x * 256;
    if (vec) {
        f32x4* p4 = reinterpret_cast<f32x4*>(p);
        const size_t n4 = n >> 2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
    }
}

// out = in * keep / (1 - p); the same launch on a gradient is the backward (nn.Dropout in training mode)
__global__ __launch_bounds__(256) void dropout_kernel(size_t n, float p, float inv_keep, unsigned long long seed, unsigned long long offset,
                                                      const float* in, float* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i] * caphn_keep_scale(seed, offset + i, p, inv_keep);
}

// out = x + branch * keep / (1 - p): a residual connection with dropout on the branch in one pass (p == 0: plain sum)
__global__ __launch_bounds__(256) void add_dropout_kernel(size_t n, float p, float inv_keep, unsigned long long seed, unsigned long long offset,
                                                          const float* __restrict__ x, const float* __restrict__ branch, float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = x[i] + branch[i] * (p > 0.f ? caphn_keep_scale(seed, offset + i, p, inv_keep) : 1.f);
}

// out = x * scale[0] (scale on the device: the upstream gradient of a scalar loss)
__global__ __launch_bounds__(256) void scale_kernel(size_t n, const float* __restrict__ x, const float* __restrict__ scale, float* __restrict__ out) {
    const float c = scale[0];
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = x[i] * c;
}

// y += alpha * x  (gradients of parameters that alias the same theta range: set_all_parameters' child-offset restart)
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(size_t n, const float* __restrict__ dy, const float* __restrict__ post, float* __restrict__ dx) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dx[i] = dy[i] * (post[i] > 0.f ? 1.f : 0.01f);
}
__global__ __launch_bounds__(256) void axpy_kernel(size_t n, float alpha, const float* __restrict__ x, float* __restrict__ y) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = fmaf(alpha, x[i], y[i]);
}

// ------------------------------------------------------------------ column sums
// 256-thread blocks = 64 columns x 4 row lanes; grid.y slices the rows.  Stage 1 writes
// part[slice][n], stage 2 is the same kernel over part (one slice).
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const float* __restrict__ A, int lda,
                                                     float* __restrict__ out, int rows_per) {
    __shared__ float red[4][65];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s0 = 0.f, s1 = 0.f;
    if (n < N) {
        int m = m0 + rl;
        for (; m + 4 < m1; m += 8) { s0 += A[(size_t)m * lda + n]; s1 += A[(size_t)(m + 4) * lda + n]; }
        if (m < m1) s0 += A[(size_t)m * lda + n];
    }
    red[rl][c] = s0 + s1;
    __syncthreads();
    if (rl == 0 && n < N) out[(size_t)blockIdx.y * N + n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
inline int colsum_slices(int M, int N) {
    const int colblocks = (N + 63) / 64;
    int S = 2048 / colblocks;            // aim at ~2048 workgroups
    if (S > (M + 31) / 32) S = (M + 31) / 32;
    if (S > 256) S = 256;
    if (S < 1) S = 1;
    return S;
}

// ------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void embed_gather_kernel(int rows, int E, const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out) {
    const int r = blockIdx.x;
    const int64_t id = idx[r];
    for (int e = threadIdx.x; e < E; e += 256) out[(size_t)r * E + e] = id < 0 ? 0.f : table[(size_t)id * E + e];
}
// out[r * ostride ...] = table[idx[r * istride]] (one time step's column of a [B, T] layout)
__global__ __launch_bounds__(256) void embed_gather_strided_kernel(int rows, int E, const float* __restrict__ table, const int64_t* __restrict__ idx,
                                                                   int istride, float* __restrict__ out, int ostride) {
    const int r = blockIdx.x;
    const int64_t id = idx[(size_t)r * istride];
    for (int e = threadIdx.x; e < E; e += 256) out[(size_t)r * ostride + e] = id < 0 ? 0.f : table[(size_t)id * E + e];
}
// deterministic variant: wave w of the grid owns table row v = w; it walks idx[0..rows) in order, adding every matching source row
__global__ __launch_bounds__(256) void embed_scatter_det_kernel(int rows, int E, int V, const float* __restrict__ g, const int64_t* __restrict__ idx,
                                                                float* __restrict__ tg) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= V) return;
    bool any = false;
    for (int r0 = 0; r0 < rows; r0 += 64) {
        const int r = r0 + lane;
        const bool hit = r < rows && idx[r] == (int64_t)v;
        unsigned long long ball = __ballot(hit);
        while (ball) {
            const int b = __ffsll((long long)ball) - 1;
            ball &= ball - 1;
            const float* src = g + (size_t)(r0 + b) * E;
            for (int e = lane; e < E; e += 64) tg[(size_t)v * E + e] += src[e];
            any = true;
        }
    }
    (void)any;
}
__global__ __launch_bounds__(256) void embed_scatter_kernel(int rows, int E, const float* __restrict__ g, const int64_t* __restrict__ idx, float* __restrict__ tg) {
    const int r = blockIdx.x;
    const int64_t id = idx[r];
    if (id < 0) return;
    for (int e = threadIdx.x; e < E; e += 256) atomicAdd(&tg[(size_t)id * E + e], g[(size_t)r * E + e]);
}

// ------------------------------------------------------------------ gradient norm
constexpr int SUMSQ_CHUNK = 256 * 4 * 8;   // elements per workgroup
__global__ __launch_bounds__(256) void sumsq_kernel(size_t n, const float* __restrict__ x, double* __restrict__ partial, int vec) {
    __shared__ double red[4];
    const size_t base = (size_t)blockIdx.x * SUMSQ_CHUNK;
    float s = 0.f;
    if (vec && base + SUMSQ_CHUNK <= n) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x + base);
#pragma unroll
        for (int i = 0; i < 8; ++i) { f32x4 v = x4[threadIdx.x + 256 * i]; s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
    } else {
        for (size_t i = base + threadIdx.x; i < n && i < base + SUMSQ_CHUNK; i += 256) s += x[i] * x[i];
    }
    double d = wave_sum_d((double)s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Gram matrices of R rank-1 factors, several (gfac, afac) pairs per launch:
// ws[job][which][r*R+s] = g_r . g_s (which = 0) or a_r . a_s (which = 1).
// grid (R*R, 2*njobs, chunks): every block reduces one chunk and adds it with one fp64 atomic (ws zeroed first)
constexpr int GRAM_CHUNKS = 64;
struct GramJob { const float* gfac; size_t ldg; const float* afac; size_t lda; int rows, k; };
struct GramJobs { GramJob j[CAPHN_MAX_HEADS]; int n; };
__global__ __launch_bounds__(256) void rank_gram_kernel(int R, GramJobs jobs, double* __restrict__ ws) {
    __shared__ double red[4];
    const int pair = blockIdx.x, job = blockIdx.y >> 1, which = blockIdx.y & 1;
    const GramJob& J = jobs.j[job];
    const int r = pair / R, s = pair % R;
    const float* u = which ? J.afac + (size_t)r * J.lda : J.gfac + (size_t)r * J.ldg;
    const float* v = which ? J.afac + (size_t)s * J.lda : J.gfac + (size_t)s * J.ldg;
    const int n = which ? J.k : J.rows;
    const int per = (n + GRAM_CHUNKS - 1) / GRAM_CHUNKS;
    const int i0 = blockIdx.z * per, i1 = min(n, i0 + per);
    if (i0 >= i1) return;
    double acc = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) acc += (double)u[i] * (double)v[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&ws[((size_t)job * 2 + which) * R * R + pair], red[0] + red[1] + red[2] + red[3]);
}
__global__ void rank_gram_finish_kernel(int R, int njobs, const double* __restrict__ ws, double* acc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int j = 0; j < njobs; ++j)
            for (int i = 0; i < R * R; ++i) t += ws[((size_t)j * 2) * R * R + i] * ws[((size_t)j * 2 + 1) * R * R + i];
        acc[0] += t;
    }
}
__global__ __launch_bounds__(256) void clip_coef_kernel(int nparts, const double* __restrict__ partial, const double* extra,
                                                        double max_norm, double scale, float* coef_out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = red[0] + red[1] + red[2] + red[3] + (extra ? extra[0] : 0.0);
        double norm = scale * sqrt(tot);
        double c = max_norm / (norm + 1e-6);          // torch.nn.utils.clip_grad_norm_
        if (c > 1.0) c = 1.0;
        coef_out[0] = (float)(scale * c);
        coef_out[1] = (float)norm;
    }
}

// The whole clip coefficient in two launches: (1) partial sums -- sum of squares of the dense gradient arena and the dot
// products behind the Gram-matrix norm of the rank-R second-layer gradients -- one slot per block, no atomics, fixed
// summation order; (2) one block reduces the slots and writes clip_grad_norm_'s coefficient.  Was six launches (sumsq,
// two fills, gram, gram finish, clip_coef) with 10-40 us of idle chip between them in front of the optimiser.  (A
// single launch whose last-arriving block finishes the job -- ticket counter + agent-scope release / acquire -- measured
// 33 us: 754 fences and tickets cost more than the kernel boundary they replace.)
//   blocks [0, nb_s): one SUMSQ_CHUNK of x each                      -> part[blk]
//   then per (job, which, pair, chunk): one chunk of one dot product -> gram slot
constexpr int NORM_CHUNKS = 64;       // (16: a chunk of the 240000-long d theta dot product was 59 dependent fp64 FMAs per thread -- the
                                      //  longest block of the launch; 64 chunks x eight loads in flight: 18.6 -> see DESIGN.md section 6)
__global__ __launch_bounds__(256) void grad_norm_partials_kernel(size_t n, const float* __restrict__ x, int vec, int nb_s, int R,
                                                                 GramJobs jobs, double* __restrict__ part) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    double mine = 0.0;
    if (bid < nb_s) {
        const size_t base = (size_t)bid * SUMSQ_CHUNK;
        float s = 0.f;
        if (vec && base + SUMSQ_CHUNK <= n) {
            const f32x4* x4 = reinterpret_cast<const f32x4*>(x + base);
#pragma unroll
            for (int i = 0; i < 8; ++i) { f32x4 v = x4[tid + 256 * i]; s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
        } else {
            for (size_t i = base + tid; i < n && i < base + SUMSQ_CHUNK; i += 256) s += x[i] * x[i];
        }
        mine = (double)s;
    } else {
        int q = bid - nb_s;
        const int chunk = q % NORM_CHUNKS; q /= NORM_CHUNKS;
        const int pair = q % (R * R); q /= (R * R);
        const int which = q & 1, job = q >> 1;
        const GramJob& J = jobs.j[job];
        const int r = pair / R, sidx = pair % R;
        const float* u = which ? J.afac + (size_t)r * J.lda : J.gfac + (size_t)r * J.ldg;
        const float* v = which ? J.afac + (size_t)sidx * J.lda : J.gfac + (size_t)sidx * J.ldg;
        const int len = which ? J.k : J.rows;
        const int per = (len + NORM_CHUNKS - 1) / NORM_CHUNKS;
        const int i0 = chunk * per, i1 = min(len, i0 + per);
        int i = i0 + tid;
        for (; i + 7 * 256 < i1; i += 8 * 256) {
            float uu[8], vv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { uu[e] = u[i + 256 * e]; vv[e] = v[i + 256 * e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) mine += (double)uu[e] * (double)vv[e];
        }
        for (; i < i1; i += 256) mine += (double)u[i] * (double)v[i];
    }
    mine = wave_sum_d(mine);
    if ((tid & 63) == 0) red[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) part[bid] = red[0] + red[1] + red[2] + red[3];
}
// the reduction of grad_norm_partials_kernel's slots to clip_grad_norm_'s coefficient, by one workgroup of 256 threads (fixed order).
// Returns (in every thread) scale * min(1, max_norm / (norm + 1e-6)); *norm_out = the total norm.
__device__ __forceinline__ float norm_finish(int nb_s, int R, int njobs, const double* __restrict__ part, double max_norm, double scale,
                                             double* red /* [5] LDS */, float* norm_out) {
    const int tid = threadIdx.x;
    const double* gram = part + nb_s;            // [job][which][pair][chunk]
    double tot = 0.0;
    for (int i = tid; i < nb_s; i += 256) tot += part[i];
    const int npair = njobs * R * R;
    for (int i = tid; i < npair; i += 256) {
        const int job = i / (R * R), pair = i % (R * R);
        const double* gg = gram + (((size_t)job * 2 + 0) * R * R + pair) * NORM_CHUNKS;
        const double* aa = gram + (((size_t)job * 2 + 1) * R * R + pair) * NORM_CHUNKS;
        double sg = 0.0, sa = 0.0;
        for (int c = 0; c < NORM_CHUNKS; ++c) { sg += gg[c]; sa += aa[c]; }
        tot += sg * sa;
    }
    tot = wave_sum_d(tot);
    if ((tid & 63) == 0) red[tid >> 6] = tot;
    __syncthreads();
    const double t = red[0] + red[1] + red[2] + red[3];
    const double norm = scale * sqrt(t);
    double c = max_norm / (norm + 1e-6);          // torch.nn.utils.clip_grad_norm_
    if (c > 1.0) c = 1.0;
    *norm_out = (float)norm;
    return (float)(scale * c);
}
__global__ __launch_bounds__(256) void grad_norm_finish_kernel(int nb_s, int R, int njobs, const double* __restrict__ part,
                                                               double max_norm, double scale, float* __restrict__ coef_out) {
    __shared__ double red[5];
    float norm;
    const float c = norm_finish(nb_s, R, njobs, part, max_norm, scale, red, &norm);
    if (threadIdx.x == 0) { coef_out[0] = c; coef_out[1] = norm; }
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam, single-tensor form)
struct AdamK { float lr_bc1, b1, b2, eps, sqrt_bc2; const float* dev; int zg = 0; };    // zg: rank-1 passes clear their row factor (R == 1)
__device__ __forceinline__ float adam_elem(float p, float g, float& m, float& v, const AdamK& k) {
    m = m + (g - m) * (1.0f - k.b1);                       // exp_avg.lerp_(grad, 1 - beta1)
    v = v * k.b2 + (1.0f - k.b2) * g * g;                  // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(v) / k.sqrt_bc2 + k.eps;     // (sqrt(v) / sqrt(bc2)).add_(eps)
    return p - k.lr_bc1 * (m / denom);                     // addcdiv_(m, denom, -lr / bc1)
}
__global__ __launch_bounds__(256) void adam_dense_kernel(size_t n, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ g, const float* __restrict__ coef, AdamK k, int vec) {
    const float c = coef[0];
    if (k.dev) { k.lr_bc1 = k.dev[0]; k.sqrt_bc2 = k.dev[1]; }
    const size_t stride = (size_t)gridDim.x * 256;
    if (vec) {
        f32x4* p4 = reinterpret_cast<f32x4*>(p); f32x4* m4 = reinterpret_cast<f32x4*>(m); f32x4* v4 = reinterpret_cast<f32x4*>(v);
        const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
        const size_t n4 = n >> 2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            f32x4 pp = p4[i], mm = m4[i], vv = v4[i], gg = g4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float me = mm[e], ve = vv[e]; pp[e] = adam_elem(pp[e], gg[e] * c, me, ve, k); mm[e] = me; vv[e] = ve; }
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
        }
        for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            float me = m[i], ve = v[i]; p[i] = adam_elem(p[i], g[i] * c, me, ve, k); m[i] = me; v[i] = ve;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            float me = m[i], ve = v[i]; p[i] = adam_elem(p[i], g[i] * c, me, ve, k); m[i] = me; v[i] = ve;
        }
    }
}

// adam_dense_kernel that finishes the clip coefficient itself: every workgroup reduces grad_norm_partials_kernel's slots (a few KB
// from L2, the same fixed order in each), workgroup 0 also publishes coef / norm for the kernels behind it and -- when asked --
// reduces the cross entropy's per-row losses to the reported scalar (that one-workgroup kernel used to sit on the chain between
// the loss and the first backward GEMM; nothing on the device needs its result).  One launch instead of three.
struct NormFin { const double* part; int nb_s, R, njobs; double max_norm, scale; float* coef_out;
                 int ce_rows; const float* ce_rowloss; const float* ce_nvalid; const int* ce_nvi; float* loss_out; };
__global__ __launch_bounds__(256) void adam_dense_clip_kernel(size_t n, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                              const float* __restrict__ g, NormFin nf, AdamK k, int vec) {
    __shared__ double red[5];
    float norm;
    const float c = norm_finish(nf.nb_s, nf.R, nf.njobs, nf.part, nf.max_norm, nf.scale, red, &norm);
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) { nf.coef_out[0] = c; nf.coef_out[1] = norm; }
        if (nf.ce_rows > 0) {
            __syncthreads();
            double s = 0.0;
            for (int i = threadIdx.x; i < nf.ce_rows; i += 256) s += (double)nf.ce_rowloss[i];
            s = wave_sum_d(s);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) {
                const float nn = nf.ce_nvi ? (float)nf.ce_nvi[0] : nf.ce_nvalid[0];
                nf.loss_out[0] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)nn); nf.loss_out[1] = nn;
            }
        }
    }
    if (k.dev) { k.lr_bc1 = k.dev[0]; k.sqrt_bc2 = k.dev[1]; }
    const size_t stride = (size_t)gridDim.x * 256;
    if (vec) {
        f32x4* p4 = reinterpret_cast<f32x4*>(p); f32x4* m4 = reinterpret_cast<f32x4*>(m); f32x4* v4 = reinterpret_cast<f32x4*>(v);
        const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
        const size_t n4 = n >> 2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            f32x4 pp = p4[i], mm = m4[i], vv = v4[i], gg = g4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float me = mm[e], ve = vv[e]; pp[e] = adam_elem(pp[e], gg[e] * c, me, ve, k); mm[e] = me; vv[e] = ve; }
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
        }
        for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            float me = m[i], ve = v[i]; p[i] = adam_elem(p[i], g[i] * c, me, ve, k); m[i] = me; v[i] = ve;
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            float me = m[i], ve = v[i]; p[i] = adam_elem(p[i], g[i] * c, me, ve, k); m[i] = me; v[i] = ve;
        }
    }
}

// W,m,v [rows,k]; grad[row][col] = coef * sum_r gfac[r][row] * afac[r][col].  One wave per row
// (row-contiguous dwordx4 streams of W, m and v: read 12 B, write 12 B per element).
// RB rows per wave iteration (3*RB*QMAX independent dwordx4 loads in flight), NTL / NTS: non-temporal
// loads / stores.  Variant chosen by caphn_tune(1, v) -- measured A/B.
struct NextGemv {                      // theta[row] = W'[row,:] . a + bias[row]
    const float* a; const float* bias; float* theta;
    // optional second copy of theta[row] in the pair recurrent kernels' packed per-half layout (caphn_rank_job::next_pack): row is an
    // element index of W_hh [NG H, H]; element (R, c) of gate block q = R / H, j = R % H goes to half hh = (j >= HA), local row
    // (q + 1) nk_hh + (j - k0_hh), column c
    float* pack = nullptr; int pH = 0, pHA = 0, ppitch = 0, phrows = 0;
};
__device__ __forceinline__ void next_store(const NextGemv& nx, int row, float v) {
    nx.theta[row] = v;
    if (nx.pack) {
        const int R = row / nx.pH, c = row - R * nx.pH, q = R / nx.pH, j = R - q * nx.pH;
        const int hh = j >= nx.pHA ? 1 : 0, kk = j - (hh ? nx.pHA : 0), nk = hh ? nx.pH - nx.pHA : nx.pHA;
        nx.pack[((size_t)hh * nx.phrows + (size_t)(q + 1) * nk + kk) * nx.ppitch + c] = v;
    }
}
template <int QMAX, int RB, bool NTL, bool NTS, bool MULTI>
__device__ __forceinline__ void adam_rank_rows(int R, int rows, int k, float* W, float* m, float* v,
                                               const float* gfac, size_t ldg, const float* afac, size_t lda,
                                               float c, const AdamK& K, int wave_g, int nwaves, int lane, NextGemv nx,
                                               const f32x4* a_s /* LDS copy of afac [R][k/4] (R > 1), or null */) {
    const int k4 = k >> 2;
    f32x4 an[QMAX];            // next step's head activations (fused forward GEMV on the updated weights)
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        const int cidx = lane + 64 * q;
        an[q] = (nx.a && cidx < k4) ? reinterpret_cast<const f32x4*>(nx.a)[cidx] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 a1[QMAX];            // column factors stay in registers for every row (R == 1 fast path)
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        const int cidx = lane + 64 * q;
        a1[q] = (!MULTI && cidx < k4) ? reinterpret_cast<const f32x4*>(afac)[cidx] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row0 = wave_g * RB; row0 < rows; row0 += nwaves * RB) {
        f32x4 pp[RB][QMAX], mm[RB][QMAX], vv[RB][QMAX];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const size_t base = (size_t)(row0 + i) * k;
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                const int cidx = lane + 64 * q;
                if (row0 + i < rows && cidx < k4) {
                    const f32x4* pw = reinterpret_cast<const f32x4*>(W + base) + cidx;
                    const f32x4* pm = reinterpret_cast<const f32x4*>(m + base) + cidx;
                    const f32x4* pv = reinterpret_cast<const f32x4*>(v + base) + cidx;
                    pp[i][q] = NTL ? __builtin_nontemporal_load(pw) : *pw;
                    mm[i][q] = NTL ? __builtin_nontemporal_load(pm) : *pm;
                    vv[i][q] = NTL ? __builtin_nontemporal_load(pv) : *pv;
                }
            }
        }
        // the R row factors of these RB rows: requested together with the streams above (they were loaded one row at a
        // time in front of each row's arithmetic: R dependent cache-line fetches per row, +54 % at R = 8)
        // (MULTI is a template parameter: with the R == 1 case in the same code the extra registers cost the
        // single-GPU pass 12 %)
        float grs[RB][MULTI ? RMAX : 1];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int r = 0; r < (MULTI ? RMAX : 1); ++r)
                grs[i][r] = (r < R && row0 + i < rows) ? gfac[(size_t)r * ldg + row0 + i] : 0.f;
        if constexpr (!MULTI) {
            // caphn_adam_hparams::zero_gfac: this wave is the LAST reader of its rows' factor (d theta in the trainer's gradient
            // arena, which the next backward accumulates into): clear it here instead of with a launch in front of the next forward
            if (K.zg && lane == 0) {
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    if (row0 + i < rows) const_cast<float*>(gfac)[row0 + i] = 0.f;
            }
        }
        float dots[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) dots[i] = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (row0 + i >= rows) continue;
            const size_t base = (size_t)(row0 + i) * k;
            float dot = 0.f;
            float gr[MULTI ? RMAX : 1];
#pragma unroll
            for (int r = 0; r < (MULTI ? RMAX : 1); ++r) gr[r] = grs[i][r] * c;
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                const int cidx = lane + 64 * q;
                if (cidx < k4) {
                    f32x4 g;
                    if constexpr (!MULTI) g = a1[q] * gr[0];
                    else {
                        g = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int r = 0; r < RMAX; ++r)
                            if (r < R)
                                g += (a_s ? a_s[r * k4 + cidx] : reinterpret_cast<const f32x4*>(afac + (size_t)r * lda)[cidx]) * gr[r];
                    }
                    f32x4 po = pp[i][q], mo = mm[i][q], vo = vv[i][q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float me = mo[e], ve = vo[e]; po[e] = adam_elem(po[e], g[e], me, ve, K); mo[e] = me; vo[e] = ve; }
                    dot += po[0] * an[q][0] + po[1] * an[q][1] + po[2] * an[q][2] + po[3] * an[q][3];
                    f32x4* qw = reinterpret_cast<f32x4*>(W + base) + cidx;
                    f32x4* qm = reinterpret_cast<f32x4*>(m + base) + cidx;
                    f32x4* qv = reinterpret_cast<f32x4*>(v + base) + cidx;
                    if (NTS) { __builtin_nontemporal_store(po, qw); __builtin_nontemporal_store(mo, qm); __builtin_nontemporal_store(vo, qv); }
                    else { *qw = po; *qm = mo; *qv = vo; }
                }
            }
            if (nx.a) dots[i] = wave_sum(dot);          // (wave-uniform)
        }
        if (nx.a && lane == 0) {
            // the RB rows' results leave in ONE store each for theta and for the packed copy when the block is whole and aligned
            // (scalar write-through stores are one fabric write each: 120 000 + 120 000 of them cost the W_hh pass ~10 us)
            bool vec_ok = row0 + RB <= rows && ((reinterpret_cast<uintptr_t>(nx.theta + row0) & (sizeof(float) * RB - 1)) == 0) &&
                          (RB == 2 || RB == 4);
            if (vec_ok && nx.pack) vec_ok = (nx.pH % RB) == 0 && (nx.ppitch % RB) == 0 && (reinterpret_cast<uintptr_t>(nx.pack) & 15) == 0;
            if (vec_ok) {
                float vals[RB];
#pragma unroll
                for (int i = 0; i < RB; ++i) vals[i] = dots[i] + nx.bias[row0 + i];
                float* pd = nullptr;
                if (nx.pack) {
                    const int R = row0 / nx.pH, c = row0 - R * nx.pH, q = R / nx.pH, j = R - q * nx.pH;
                    const int hh = j >= nx.pHA ? 1 : 0, kk = j - (hh ? nx.pHA : 0), nk = hh ? nx.pH - nx.pHA : nx.pHA;
                    pd = nx.pack + ((size_t)hh * nx.phrows + (size_t)(q + 1) * nk + kk) * nx.ppitch + c;
                }
                if constexpr (RB == 4) {
                    const f32x4 v4 = {vals[0], vals[1], vals[2], vals[3]};
                    *reinterpret_cast<f32x4*>(nx.theta + row0) = v4;
                    if (pd) *reinterpret_cast<f32x4*>(pd) = v4;
                } else if constexpr (RB == 2) {
                    typedef float f32x2v __attribute__((ext_vector_type(2)));
                    const f32x2v v2 = {vals[0], vals[1]};
                    *reinterpret_cast<f32x2v*>(nx.theta + row0) = v2;
                    if (pd) *reinterpret_cast<f32x2v*>(pd) = v2;
                }
            } else {
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    if (row0 + i < rows) next_store(nx, row0 + i, dots[i] + nx.bias[row0 + i]);
            }
        }
    }
}
// QMAX is a kernel template parameter chosen on the host (see gemv_fwd_kernel in hyper.hip: with every width
// inlined the kernel allocated 255 VGPRs and ran at 1-2 waves/SIMD).
template <int RB, bool NTL, bool NTS, int QMAX, bool MULTI>
__global__ __launch_bounds__(256) void adam_rank_kernel(int R, int rows, int k, float* W, float* m, float* v,
                                                        const float* gfac, size_t ldg, const float* afac, size_t lda,
                                                        const float* coef, AdamK K, int vec, NextGemv nx, int use_lds) {
    extern __shared__ __attribute__((aligned(16))) f32x4 a_lds[];     // R > 1: the ranks' column factors, [R][k/4]
    const float c = coef[0];
    if (K.dev) { K.lr_bc1 = K.dev[0]; K.sqrt_bc2 = K.dev[1]; }
    if (vec && k <= 256 * QMAX) {
        // data parallel: the gradient is sum_r g_r (x) a_r.  Re-reading a_r from global memory for every row made the
        // pass 54 % slower at R = 8 (735 vs 477 us on 240000 x 480); the factors (R*k floats, 15 KB) live in LDS instead
        const f32x4* a_s = nullptr;
        if (R > 1 && use_lds) {
            const int k4 = k >> 2;
            for (int i = threadIdx.x; i < R * k4; i += 256) {
                const int r = i / k4, cc = i - r * k4;
                a_lds[i] = reinterpret_cast<const f32x4*>(afac + (size_t)r * lda)[cc];
            }
            __syncthreads();
            a_s = a_lds;
        }
        const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4, lane = threadIdx.x & 63;
        adam_rank_rows<QMAX, RB, NTL, NTS, MULTI>(R, rows, k, W, m, v, gfac, ldg, afac, lda, c, K, wave_g, nwaves, lane, nx, a_s);
    } else {
        const size_t n = (size_t)rows * k, stride = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const size_t row = i / k, col = i % k;
            float g = 0.f;
            for (int r = 0; r < R; ++r) g += gfac[(size_t)r * ldg + row] * afac[(size_t)r * lda + col];
            float me = m[i], ve = v[i]; W[i] = adam_elem(W[i], g * c, me, ve, K); m[i] = me; v[i] = ve;
        }
    }
}
// Several SMALL rank-R members in one launch (blockIdx.y = member): the bias heads of the hypernet are two [600, k] matrices, 11 us
// of launch-bound kernel each on the optimiser's chain.  Same row code; the members share k's width class and R.
struct RankJob { float* W; float* m; float* v; const float* gfac; size_t ldg; const float* afac; size_t lda; NextGemv nx; int rows, k; };
struct RankJobs { RankJob j[4]; };
template <int RB, bool NTL, bool NTS, int QMAX, bool MULTI>
__global__ __launch_bounds__(256) void adam_rank_jobs_kernel(int R, RankJobs jobs, const float* coef, AdamK K, int use_lds) {
    extern __shared__ __attribute__((aligned(16))) f32x4 a_lds[];
    const RankJob& J = jobs.j[blockIdx.y];
    if ((long)blockIdx.x * 4 * RB >= J.rows) return;          // block-uniform: the grid is sized for the largest member
    const float c = coef[0];
    if (K.dev) { K.lr_bc1 = K.dev[0]; K.sqrt_bc2 = K.dev[1]; }
    const f32x4* a_s = nullptr;
    if (R > 1 && use_lds) {
        const int k4 = J.k >> 2;
        for (int i = threadIdx.x; i < R * k4; i += 256) {
            const int r = i / k4, cc = i - r * k4;
            a_lds[i] = reinterpret_cast<const f32x4*>(J.afac + (size_t)r * J.lda)[cc];
        }
        __syncthreads();
        a_s = a_lds;
    }
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4, lane = threadIdx.x & 63;
    adam_rank_rows<QMAX, RB, NTL, NTS, MULTI>(R, J.rows, J.k, J.W, J.m, J.v, J.gfac, J.ldg, J.afac, J.lda, c, K, wave_g, nwaves, lane, J.nx, a_s);
}
// Long rows of any width / alignment (hypernet.py's heads: k = 11250, 8437 -- rows that are only 8 or 4 byte aligned):
// one wave per row, lane-contiguous dword streams (256 B per wave instruction whatever the alignment), U columns per
// lane in flight for each of W, m, v.  The element-linear fallback below (a 64-bit divide per element) moved 1.4 TB/s
// on those shapes.  The next step's theta row is the wave's dot product of the updated row, as in adam_rank_rows.
template <bool NT, bool NTS>
__global__ __launch_bounds__(256) void adam_rank_long_kernel(int R, int rows, int k, float* __restrict__ W, float* __restrict__ m,
                                                             float* __restrict__ v, const float* __restrict__ gfac, size_t ldg,
                                                             const float* __restrict__ afac, size_t lda,
                                                             const float* __restrict__ coef, AdamK K, NextGemv nx) {
    const float c = coef[0];
    if (K.dev) { K.lr_bc1 = K.dev[0]; K.sqrt_bc2 = K.dev[1]; }
    constexpr int U = 8;
    const int lane = threadIdx.x & 63, wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const float* __restrict__ na = nx.a;
    for (int r = wave_g; r < rows; r += nwaves) {
        const size_t base = (size_t)r * k;
        float gr[RMAX];
#pragma unroll
        for (int q = 0; q < RMAX; ++q) gr[q] = q < R ? gfac[(size_t)q * ldg + r] : 0.f;
        float dot = 0.f;
        int col = lane;
        for (; col + 64 * (U - 1) < k; col += 64 * U) {
            float w[U], mm[U], vv[U], a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t i = base + col + 64 * u;
                if (NT) { w[u] = __builtin_nontemporal_load(W + i); mm[u] = __builtin_nontemporal_load(m + i); vv[u] = __builtin_nontemporal_load(v + i); }
                else { w[u] = W[i]; mm[u] = m[i]; vv[u] = v[i]; }
                a[u] = afac[col + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float g = gr[0] * a[u];
                for (int q = 1; q < R; ++q) g += gr[q] * afac[(size_t)q * lda + col + 64 * u];
                w[u] = adam_elem(w[u], g * c, mm[u], vv[u], K);
                if (na) dot += w[u] * na[col + 64 * u];
                const size_t i = base + col + 64 * u;
                if (NTS) { __builtin_nontemporal_store(w[u], W + i); __builtin_nontemporal_store(mm[u], m + i); __builtin_nontemporal_store(vv[u], v + i); }
                else { W[i] = w[u]; m[i] = mm[u]; v[i] = vv[u]; }
            }
        }
        for (; col < k; col += 64) {
            const size_t i = base + col;
            float g = 0.f;
            for (int q = 0; q < R; ++q) g += gr[q] * afac[(size_t)q * lda + col];
            float me = m[i], ve = v[i];
            const float wn = adam_elem(W[i], g * c, me, ve, K);
            W[i] = wn; m[i] = me; v[i] = ve;
            if (na) dot += wn * na[col];
        }
        if (na) {
            dot = wave_sum(dot);
            if (lane == 0) next_store(nx, r, dot + nx.bias[r]);
        }
    }
}
// generic-shape companion of the fused GEMV (k % 4 != 0 or unaligned): theta[row] = W[row,:] . a + bias[row]
__global__ __launch_bounds__(256) void rowdot_kernel(int rows, int k, const float* __restrict__ W, NextGemv nx) {
    const int grp = threadIdx.x >> 3, s = threadIdx.x & 7;
    for (int r = blockIdx.x * 32 + grp; r < rows; r += gridDim.x * 32) {
        float sum = 0.f;
        for (int c = s; c < k; c += 8) sum += W[(size_t)r * k + c] * nx.a[c];
        sum += __shfl_xor(sum, 4, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 1, 64);
        if (s == 0) next_store(nx, r, sum + nx.bias[r]);
    }
}

}  // namespace
int g_tune_adam_dense_cap = 2048;   // workgroups of the dense arena's Adam launch (caphn_tune key 31)
int g_tune_adam_cap = 16384;  // workgroups of one rank-1 Adam launch (grid-stride over rows).  Round 3, alternating blocks in one process
                              // (tools/ab_inproc.py): 4096 -> 8192 -11..-26 us per step, -> 16384 -18..-23, 32768 +12 over 8192 (round 2 only
                              // ever tried FEWER: 1024 / 2048 were slower too)
int g_tune_adam = 6;   // measured on 240000x480 after the occupancy fix: 6 (non-temporal, 2 rows/iteration) 497 us,
                       // 3 (non-temporal, 1 row) 508 us, 0 (plain) 541 us
namespace {
// ------------------------------------------------------------------ multi-tensor forms (caphn.optim.FusedAdam: parameters of a module
// that live in separate allocations).  One launch covers up to MT_MAX tensors: the descriptors travel as kernel arguments
// (gradient addresses change every step), a block finds its tensor by a scan over the block prefix.
constexpr int MT_MAX = 48;
struct MTDesc { float* p; float* m; float* v; const float* g; size_t n; };
struct MTJobs { int n; unsigned blk0[MT_MAX + 1]; MTDesc t[MT_MAX]; };
__device__ __forceinline__ int mt_find(const MTJobs& J, unsigned bid) {
    int i = 0;
    while (i + 1 < J.n && bid >= J.blk0[i + 1]) ++i;
    return i;
}
// blocks [0, nb_s): one SUMSQ_CHUNK of one gradient tensor each -> part[blk]; then the Gram-dot chunks, exactly as in
// grad_norm_partials_kernel (same slot layout behind nb_s, so grad_norm_finish_kernel finishes both)
__global__ __launch_bounds__(256) void grad_norm_multi_partials_kernel(MTJobs T, int nb_s, int R, GramJobs jobs,
                                                                       double* __restrict__ dense_part, double* __restrict__ gram_part) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    double mine = 0.0;
    if (bid < nb_s) {
        const int ti = mt_find(T, (unsigned)bid);
        const float* x = T.t[ti].g;
        const size_t n = T.t[ti].n, base = (size_t)(bid - T.blk0[ti]) * SUMSQ_CHUNK;
        float s = 0.f;
        if (caphn_aligned16_dev(x) && base + SUMSQ_CHUNK <= n) {
            const f32x4* x4 = reinterpret_cast<const f32x4*>(x + base);
#pragma unroll
            for (int i = 0; i < 8; ++i) { f32x4 v = x4[tid + 256 * i]; s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
        } else {
            for (size_t i = base + tid; i < n && i < base + SUMSQ_CHUNK; i += 256) s += x[i] * x[i];
        }
        mine = (double)s;
    } else {
        int q = bid - nb_s;
        const int chunk = q % NORM_CHUNKS; q /= NORM_CHUNKS;
        const int pair = q % (R * R); q /= (R * R);
        const int which = q & 1, job = q >> 1;
        const GramJob& J = jobs.j[job];
        const int r = pair / R, sidx = pair % R;
        const float* u = which ? J.afac + (size_t)r * J.lda : J.gfac + (size_t)r * J.ldg;
        const float* v = which ? J.afac + (size_t)sidx * J.lda : J.gfac + (size_t)sidx * J.ldg;
        const int len = which ? J.k : J.rows;
        const int per = (len + NORM_CHUNKS - 1) / NORM_CHUNKS;
        const int i0 = chunk * per, i1 = min(len, i0 + per);
        int i = i0 + tid;
        for (; i + 7 * 256 < i1; i += 8 * 256) {
            float uu[8], vv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { uu[e] = u[i + 256 * e]; vv[e] = v[i + 256 * e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) mine += (double)uu[e] * (double)vv[e];
        }
        for (; i < i1; i += 256) mine += (double)u[i] * (double)v[i];
    }
    mine = wave_sum_d(mine);
    if ((tid & 63) == 0) red[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) { if (bid < nb_s) dense_part[bid] = red[0] + red[1] + red[2] + red[3]; else gram_part[bid - nb_s] = red[0] + red[1] + red[2] + red[3]; }
}
constexpr int MT_CHUNK = 256 * 4 * 4;      // elements per block of the multi-tensor Adam
__global__ __launch_bounds__(256) void adam_multi_kernel(MTJobs T, const float* __restrict__ coef, AdamK k) {
    const float c = coef[0];
    if (k.dev) { k.lr_bc1 = k.dev[0]; k.sqrt_bc2 = k.dev[1]; }
    const int ti = mt_find(T, blockIdx.x);
    const MTDesc d = T.t[ti];
    const size_t base = (size_t)(blockIdx.x - T.blk0[ti]) * MT_CHUNK;
    const bool vec = caphn_aligned16_dev(d.p) && caphn_aligned16_dev(d.m) && caphn_aligned16_dev(d.v) && caphn_aligned16_dev(d.g);
    if (vec && base + MT_CHUNK <= d.n) {
        f32x4* p4 = reinterpret_cast<f32x4*>(d.p + base); f32x4* m4 = reinterpret_cast<f32x4*>(d.m + base);
        f32x4* v4 = reinterpret_cast<f32x4*>(d.v + base); const f32x4* g4 = reinterpret_cast<const f32x4*>(d.g + base);
        f32x4 pp[4], mm[4], vv[4], gg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int j = threadIdx.x + 256 * i; pp[i] = p4[j]; mm[i] = m4[j]; vv[i] = v4[j]; gg[i] = g4[j]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float me = mm[i][e], ve = vv[i][e]; pp[i][e] = adam_elem(pp[i][e], gg[i][e] * c, me, ve, k); mm[i][e] = me; vv[i][e] = ve; }
            const int j = threadIdx.x + 256 * i;
            p4[j] = pp[i]; m4[j] = mm[i]; v4[j] = vv[i];
        }
    } else {
        for (size_t i = base + threadIdx.x; i < d.n && i < base + MT_CHUNK; i += 256) {
            float me = d.m[i], ve = d.v[i]; d.p[i] = adam_elem(d.p[i], d.g[i] * c, me, ve, k); d.m[i] = me; d.v[i] = ve;
        }
    }
}

// Copy microbenchmark (bench.py: roofline.copy_ceiling_gbps): what this box's HBM gives a kernel that streams as many bytes in as
// out with the access pattern of the rank-1 Adam pass (non-temporal dwordx4 loads and stores, four in flight per lane)
__global__ __launch_bounds__(256) void stream_copy_kernel(size_t n4, const f32x4* __restrict__ src, f32x4* __restrict__ dst) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + 256 * u < n4) v[u] = __builtin_nontemporal_load(src + i + 256 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + 256 * u < n4) __builtin_nontemporal_store(v[u], dst + i + 256 * u);
    }
}

inline AdamK make_adam(const caphn_adam_hparams* hp) {
    AdamK k;
    const double bc1 = 1.0 - pow((double)hp->beta1, (double)hp->step);
    const double bc2 = 1.0 - pow((double)hp->beta2, (double)hp->step);
    k.lr_bc1 = (float)((double)hp->lr / bc1);
    k.b1 = hp->beta1; k.b2 = hp->beta2; k.eps = hp->eps;
    k.sqrt_bc2 = (float)sqrt(bc2);
    k.dev = hp->dev_scalars;
    k.zg = hp->zero_gfac;
    return k;
}

}  // namespace

extern "C" size_t caphn_ce_workspace_bytes(int rows) { return sizeof(float) * (size_t)(rows + 4); }

extern "C" int caphn_cross_entropy_rows(int rows, int V, const float* logits, const int64_t* targets, int64_t ignore_index,
                                        float* dlogits, int leave_ignored_rows, const int* n_valid_dev, void* ws,
                                        caphn_stream_t stream) {
    return caphn_cross_entropy_rows_ld(rows, V, V, logits, targets, ignore_index, dlogits, leave_ignored_rows, n_valid_dev, ws, stream);
}
extern "C" int caphn_cross_entropy_rows_ld(int rows, int V, int ld, const float* logits, const int64_t* targets, int64_t ignore_index,
                                           float* dlogits, int leave_ignored_rows, const int* n_valid_dev, void* ws,
                                           caphn_stream_t stream) {
    if (rows <= 0 || V <= 0 || ld < V || !logits || !targets || !dlogits || !ws) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* nvalid = static_cast<float*>(ws);
    float* row_loss = nvalid + 4;
    const int vec = (V % 4 == 0) && (ld % 4 == 0) && caphn_aligned16(logits) && caphn_aligned16(dlogits);
    if (!n_valid_dev) hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, s, rows, targets, ignore_index, nvalid);
    const int nv = vec ? (V / 4 + 255) / 256 : 0;
    if (nv >= 1 && nv <= 4) hipLaunchKernelGGL(ce_row_reg_kernel<4>, dim3(rows), dim3(256), 0, s, V, ld, logits, targets, ignore_index, nvalid, n_valid_dev, dlogits, row_loss, leave_ignored_rows);
    else if (nv > 4 && nv <= 10) hipLaunchKernelGGL(ce_row_reg_kernel<10>, dim3(rows), dim3(256), 0, s, V, ld, logits, targets, ignore_index, nvalid, n_valid_dev, dlogits, row_loss, leave_ignored_rows);
    else if (nv > 10 && nv <= 16) hipLaunchKernelGGL(ce_row_reg_kernel<16>, dim3(rows), dim3(256), 0, s, V, ld, logits, targets, ignore_index, nvalid, n_valid_dev, dlogits, row_loss, leave_ignored_rows);
    else hipLaunchKernelGGL(ce_row_kernel, dim3(rows), dim3(256), 0, s, V, ld, logits, targets, ignore_index, nvalid, n_valid_dev, dlogits, row_loss, vec);
    return caphn_launch_status();
}
extern "C" int caphn_cross_entropy_finish(int rows, const int* n_valid_dev, float* loss_out, void* ws, caphn_stream_t stream) {
    if (rows <= 0 || !loss_out || !ws) return CAPHN_EINVAL;
    float* nvalid = static_cast<float*>(ws);
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), rows, nvalid + 4, nvalid, n_valid_dev, loss_out);
    return caphn_launch_status();
}
extern "C" int caphn_cross_entropy_fwd_bwd(int rows, int V, const float* logits, const int64_t* targets,
                                           int64_t ignore_index, float* dlogits, float* loss_out,
                                           int leave_ignored_rows, void* ws, caphn_stream_t stream) {
    if (!loss_out) return CAPHN_EINVAL;
    int rc = caphn_cross_entropy_rows(rows, V, logits, targets, ignore_index, dlogits, leave_ignored_rows, nullptr, ws, stream);
    if (rc) return rc;
    return caphn_cross_entropy_finish(rows, nullptr, loss_out, ws, stream);
}

extern "C" size_t caphn_colsum_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return sizeof(float) * (size_t)colsum_slices(M, N) * N;
}
extern "C" int caphn_colsum_f32(int M, int N, const float* A, int lda, float* out, void* ws, caphn_stream_t stream) {
    if (M <= 0 || N <= 0 || !A || !out || !ws) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int S = colsum_slices(M, N);
    const int rows_per = (M + S - 1) / S;
    float* part = static_cast<float*>(ws);
    const int cb = (N + 63) / 64;
    if (S == 1) {
        hipLaunchKernelGGL(colsum_kernel, dim3(cb, 1), dim3(256), 0, s, M, N, A, lda, out, M);
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3(cb, S), dim3(256), 0, s, M, N, A, lda, part, rows_per);
        hipLaunchKernelGGL(colsum_kernel, dim3(cb, 1), dim3(256), 0, s, S, N, part, N, out, S);
    }
    return caphn_launch_status();
}

extern "C" int caphn_zero_f32(float* p, size_t n, caphn_stream_t stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!p) return CAPHN_EINVAL;
    if (n == 0) return CAPHN_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, (int)caphn_aligned16(p));
    return caphn_launch_status();
}

extern "C" int caphn_dropout_f32(size_t n, float p, unsigned long long seed, unsigned long long offset, const float* in, float* out,
                                 caphn_stream_t stream) {
    if (n == 0) return CAPHN_OK;
    if (!in || !out || !(p >= 0.f) || !(p < 1.f)) return CAPHN_EINVAL;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), n, p, 1.0f / (1.0f - p), seed,
                       offset, in, out);
    return caphn_launch_status();
}
extern "C" int caphn_add_dropout_f32(size_t n, const float* x, const float* branch, float p, unsigned long long seed,
                                     unsigned long long offset, float* out, caphn_stream_t stream) {
    if (n == 0) return CAPHN_OK;
    if (!x || !branch || !out || !(p >= 0.f) || !(p < 1.f)) return CAPHN_EINVAL;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), n, p, 1.0f / (1.0f - p), seed,
                       offset, x, branch, out);
    return caphn_launch_status();
}
extern "C" int caphn_scale_f32(size_t n, const float* x, const float* scale_dev, float* out, caphn_stream_t stream) {
    if (n == 0) return CAPHN_OK;
    if (!x || !scale_dev || !out) return CAPHN_EINVAL;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), n, x, scale_dev, out);
    return caphn_launch_status();
}
extern "C" int caphn_lrelu_bwd_f32(size_t n, const float* dy, const float* post, float* dx, caphn_stream_t stream) {
    if (n == 0) return CAPHN_OK;
    if (!dy || !post || !dx) return CAPHN_EINVAL;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), n, dy, post, dx);
    return caphn_launch_status();
}
extern "C" int caphn_axpy_f32(size_t n, float alpha, const float* x, float* y, caphn_stream_t stream) {
    if (n == 0) return CAPHN_OK;
    if (!x || !y) return CAPHN_EINVAL;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), n, alpha, x, y);
    return caphn_launch_status();
}
extern "C" int caphn_embedding_gather(int rows, int E, const float* table, const int64_t* idx, float* out, caphn_stream_t stream) {
    if (rows <= 0 || E <= 0 || !table || !idx || !out) return CAPHN_EINVAL;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), rows, E, table, idx, out);
    return caphn_launch_status();
}
int caphn_embedding_gather_strided(int rows, int E, const float* table, const int64_t* idx, int istride, float* out, int ostride, hipStream_t s) {
    if (rows <= 0 || E <= 0 || !table || !idx || !out) return CAPHN_EINVAL;
    hipLaunchKernelGGL(embed_gather_strided_kernel, dim3(rows), dim3(256), 0, s, rows, E, table, idx, istride, out, ostride);
    return caphn_launch_status();
}
extern int g_tune_deterministic;
int g_det_vocab = 0;      // rows of the embedding table the deterministic scatter scans (set with the mode: caphn_tune(13, V))
// V: rows of THIS table (the deterministic mode scans every one of them; with the table's own row count a process may hold several
// tables -- the captioner's vocabulary, a front-end's domain table, another model's -- without one global deciding for all)
extern "C" int caphn_embedding_scatter_add_v(int rows, int E, int V, const float* g, const int64_t* idx, float* table_grad,
                                             caphn_stream_t stream) {
    if (rows <= 0 || E <= 0 || V <= 0 || !g || !idx || !table_grad) return CAPHN_EINVAL;
    if (g_tune_deterministic && rows > 1) {
        // destination-major: one wave per table row scans the indices in order and adds its matches in that order (no atomics)
        hipLaunchKernelGGL(embed_scatter_det_kernel, dim3((V + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), rows, E,
                           V, g, idx, table_grad);
        return caphn_launch_status();
    }
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), rows, E, g, idx, table_grad);
    return caphn_launch_status();
}
// (without a row count: the deterministic mode falls back to the value given with caphn_tune(13, V); prefer the _v form)
extern "C" int caphn_embedding_scatter_add(int rows, int E, const float* g, const int64_t* idx, float* table_grad, caphn_stream_t stream) {
    if (rows <= 0 || E <= 0 || !g || !idx || !table_grad) return CAPHN_EINVAL;
    if (g_tune_deterministic && g_det_vocab > 0 && rows > 1) return caphn_embedding_scatter_add_v(rows, E, g_det_vocab, g, idx, table_grad, stream);
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), rows, E, g, idx, table_grad);
    return caphn_launch_status();
}

extern "C" int caphn_sumsq_blocks(size_t n) { return (int)((n + SUMSQ_CHUNK - 1) / SUMSQ_CHUNK); }
extern "C" int caphn_sumsq_f32(size_t n, const float* x, double* partial, caphn_stream_t stream) {
    if (n == 0 || !x || !partial) return CAPHN_EINVAL;
    const int nb = caphn_sumsq_blocks(n);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, static_cast<hipStream_t>(stream), n, x, partial, (int)caphn_aligned16(x));
    return caphn_launch_status();
}
extern "C" int caphn_rank_sumsq_f32(int R, int rows, int k, const float* gfac, size_t ldg, const float* afac, size_t lda,
                                    double* acc, double* ws, caphn_stream_t stream) {
    if (R <= 0 || R > RMAX || rows <= 0 || k <= 0 || !gfac || !afac || !acc || !ws) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GramJobs jobs; jobs.n = 1;
    jobs.j[0] = GramJob{gfac, ldg, afac, lda, rows, k};
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * R * R, s) != hipSuccess) return CAPHN_ELAUNCH;
    hipLaunchKernelGGL(rank_gram_kernel, dim3(R * R, 2, GRAM_CHUNKS), dim3(256), 0, s, R, jobs, ws);
    hipLaunchKernelGGL(rank_gram_finish_kernel, dim3(1), dim3(64), 0, s, R, 1, ws, acc);
    return caphn_launch_status();
}
extern "C" int caphn_rank_sumsq_multi_f32(int R, int n, const int* rows, const int* k, const float* const* gfac, const size_t* ldg,
                                          const float* const* afac, const size_t* lda, double* acc, double* ws,
                                          caphn_stream_t stream) {
    if (R <= 0 || R > RMAX || n <= 0 || n > CAPHN_MAX_HEADS || !rows || !k || !gfac || !ldg || !afac || !lda || !acc || !ws)
        return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GramJobs jobs; jobs.n = n;
    for (int i = 0; i < n; ++i) {
        if (rows[i] <= 0 || k[i] <= 0 || !gfac[i] || !afac[i]) return CAPHN_EINVAL;
        jobs.j[i] = GramJob{gfac[i], ldg[i], afac[i], lda[i], rows[i], k[i]};
    }
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * R * R * n, s) != hipSuccess) return CAPHN_ELAUNCH;
    hipLaunchKernelGGL(rank_gram_kernel, dim3(R * R, 2 * n, GRAM_CHUNKS), dim3(256), 0, s, R, jobs, ws);
    hipLaunchKernelGGL(rank_gram_finish_kernel, dim3(1), dim3(64), 0, s, R, n, ws, acc);
    return caphn_launch_status();
}
extern "C" size_t caphn_grad_norm_workspace_bytes(size_t n, int R, int njobs) {
    if (R <= 0 || R > RMAX || njobs < 0 || njobs > CAPHN_MAX_HEADS) return 0;
    return sizeof(double) * ((size_t)caphn_sumsq_blocks(n) + (size_t)njobs * 2 * R * R * NORM_CHUNKS);
}
extern "C" int caphn_grad_norm_coef(size_t n, const float* x, int R, int njobs, const int* rows, const int* k,
                                    const float* const* gfac, const size_t* ldg, const float* const* afac, const size_t* lda,
                                    double max_norm, double scale, float* coef_out, void* ws, caphn_stream_t stream) {
    if (n == 0 || !x || R <= 0 || R > RMAX || njobs < 0 || njobs > CAPHN_MAX_HEADS || !coef_out || !ws) return CAPHN_EINVAL;
    GramJobs jobs; jobs.n = njobs;
    for (int i = 0; i < njobs; ++i) {
        if (!rows || !k || !gfac || !ldg || !afac || !lda || rows[i] <= 0 || k[i] <= 0 || !gfac[i] || !afac[i]) return CAPHN_EINVAL;
        jobs.j[i] = GramJob{gfac[i], ldg[i], afac[i], lda[i], rows[i], k[i]};
    }
    const int nb_s = caphn_sumsq_blocks(n);
    const unsigned nb = (unsigned)nb_s + (unsigned)(njobs * 2 * R * R * NORM_CHUNKS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(grad_norm_partials_kernel, dim3(nb), dim3(256), 0, s, n, x, (int)caphn_aligned16(x), nb_s, R, jobs,
                       static_cast<double*>(ws));
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(256), 0, s, nb_s, R, njobs, static_cast<const double*>(ws), max_norm,
                       scale, coef_out);
    return caphn_launch_status();
}
extern "C" int caphn_clip_coef(int nparts, const double* partial, const double* extra, double max_norm, double scale,
                               float* coef_out, caphn_stream_t stream) {
    if (nparts < 0 || (nparts > 0 && !partial) || !coef_out) return CAPHN_EINVAL;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), nparts, partial, extra, max_norm, scale, coef_out);
    return caphn_launch_status();
}

extern "C" int caphn_adam_dense_f32(size_t n, float* p, float* m, float* v, const float* g, const float* coef,
                                    const caphn_adam_hparams* hp, caphn_stream_t stream) {
    if (n == 0 || !p || !m || !v || !g || !coef || !hp || hp->step < 1) return CAPHN_EINVAL;
    const int vec = caphn_aligned16(p) && caphn_aligned16(m) && caphn_aligned16(v) && caphn_aligned16(g);
    size_t nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adam_dense_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), n, p, m, v, g, coef, make_adam(hp), vec);
    return caphn_launch_status();
}
static int adam_rank_launch(int R, int rows, int k, float* W, float* m, float* v,
                            const float* gfac, size_t ldg, const float* afac, size_t lda,
                            const float* coef, const caphn_adam_hparams* hp, NextGemv nx, caphn_stream_t stream) {
    if (R <= 0 || R > RMAX || rows <= 0 || k <= 0 || !W || !m || !v || !gfac || !afac || !coef || !hp || hp->step < 1) return CAPHN_EINVAL;
    const int vec = (k % 4 == 0) && (lda % 4 == 0) && caphn_aligned16(W) && caphn_aligned16(m) && caphn_aligned16(v) && caphn_aligned16(afac)
                    && (!nx.a || caphn_aligned16(nx.a));
    long nb = ((long)rows + 3) / 4;
    if (nb > g_tune_adam_cap) nb = g_tune_adam_cap;
    if (nb < 1) nb = 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    AdamK K = make_adam(hp);
    if (K.zg && R != 1) return CAPHN_EINVAL;
    const bool rowpath = vec && k <= 2048;      // the kernels that clear the row factor themselves; the others get a fill behind them
    const bool zero_after = K.zg && !rowpath;
    if (zero_after) K.zg = 0;
    const bool longrow = !(vec && k <= 2048) && k >= 512;
    if (longrow) {
        long nl = ((long)rows + 3) / 4;
        if (nl > 8192) nl = 8192;
        // rows that are not 128-byte aligned share cache lines with their neighbours: non-temporal accesses then fetch /
        // write those lines twice (13.5 ms plain vs 15.7 ms non-temporal per step on hypernet.py's literal configuration)
        if (g_tune_adam == 7) hipLaunchKernelGGL((adam_rank_long_kernel<false, true>), dim3((unsigned)nl), dim3(256), 0, s, R, rows, k, W, m, v, gfac, ldg, afac, lda, coef, K, nx);
        else if (g_tune_adam == 3) hipLaunchKernelGGL((adam_rank_long_kernel<true, true>), dim3((unsigned)nl), dim3(256), 0, s, R, rows, k, W, m, v, gfac, ldg, afac, lda, coef, K, nx);
        else hipLaunchKernelGGL((adam_rank_long_kernel<false, false>), dim3((unsigned)nl), dim3(256), 0, s, R, rows, k, W, m, v, gfac, ldg, afac, lda, coef, K, nx);
        if (zero_after) return caphn_zero_f32(const_cast<float*>(gfac), (size_t)rows, stream);
        return caphn_launch_status();
    }
    const bool fused = vec && k <= 2048;
    NextGemv nk = fused ? nx : NextGemv{nullptr, nullptr, nullptr};
    const size_t a_bytes = (R > 1 && vec && k <= 2048) ? sizeof(float) * (size_t)R * k : 0;
    const int use_lds = a_bytes > 0 && a_bytes <= 60 * 1024;
    const size_t shm = use_lds ? a_bytes : 0;
#define ADAM_RANK_LAUNCH_Q(RB, NTL, NTS, Q) do { \
        if (R == 1) hipLaunchKernelGGL((adam_rank_kernel<RB, NTL, NTS, Q, false>), dim3((unsigned)nb), dim3(256), shm, s, \
                                       R, rows, k, W, m, v, gfac, ldg, afac, lda, coef, K, vec, nk, use_lds); \
        else hipLaunchKernelGGL((adam_rank_kernel<RB, NTL, NTS, Q, true>), dim3((unsigned)nb), dim3(256), shm, s, \
                                R, rows, k, W, m, v, gfac, ldg, afac, lda, coef, K, vec, nk, use_lds); } while (0)
#define ADAM_RANK_LAUNCH(RB, NTL, NTS) do { \
        if (k <= 256) ADAM_RANK_LAUNCH_Q(RB, NTL, NTS, 1); else if (k <= 512) ADAM_RANK_LAUNCH_Q(RB, NTL, NTS, 2); \
        else if (k <= 1024) ADAM_RANK_LAUNCH_Q(RB, NTL, NTS, 4); else ADAM_RANK_LAUNCH_Q(RB, NTL, NTS, 8); } while (0)
    // caphn_tune(1, v): 0 plain loads/stores, RB=1; 3 (default) non-temporal, RB=1; 6 non-temporal, RB=2
    // (fused pass, measured in the step with tools/ab_inproc.py and removed again: plain loads and stores +77 us per step, plain stores
    //  +59, plain loads +76, one row per iteration +90, four rows per iteration for k > 256 -4: profiles/r03_inproc_ab.txt)
    if (nk.a && g_tune_adam != 0) {
        // fused next-theta GEMV: with two (four for short rows) rows per iteration all loads are issued before
        // either row's shuffle reduction
        if (k > 256) ADAM_RANK_LAUNCH(2, true, true); else ADAM_RANK_LAUNCH(4, true, true);
    } else if (g_tune_adam == 0) ADAM_RANK_LAUNCH(1, false, false);
    else if (g_tune_adam == 3) ADAM_RANK_LAUNCH(1, true, true);
    else ADAM_RANK_LAUNCH(2, true, true);
#undef ADAM_RANK_LAUNCH_Q
#undef ADAM_RANK_LAUNCH
    if (nx.a && !fused) {
        long nr = ((long)rows + 31) / 32; if (nr > 2048) nr = 2048;
        hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)nr), dim3(256), 0, s, rows, k, W, nx);
    }
    if (zero_after) return caphn_zero_f32(const_cast<float*>(gfac), (size_t)rows, stream);
    return caphn_launch_status();
}
extern "C" int caphn_adam_rank_f32(int R, int rows, int k, float* W, float* m, float* v,
                                   const float* gfac, size_t ldg, const float* afac, size_t lda,
                                   const float* coef, const caphn_adam_hparams* hp, caphn_stream_t stream) {
    return adam_rank_launch(R, rows, k, W, m, v, gfac, ldg, afac, lda, coef, hp, NextGemv{nullptr, nullptr, nullptr}, stream);
}
extern "C" int caphn_adam_rank_gemv_f32(int R, int rows, int k, float* W, float* m, float* v,
                                        const float* gfac, size_t ldg, const float* afac, size_t lda,
                                        const float* coef, const caphn_adam_hparams* hp,
                                        const float* next_a, const float* next_bias, float* next_theta,
                                        caphn_stream_t stream) {
    if (!next_a || !next_bias || !next_theta) return CAPHN_EINVAL;
    return adam_rank_launch(R, rows, k, W, m, v, gfac, ldg, afac, lda, coef, hp, NextGemv{next_a, next_bias, next_theta}, stream);
}

extern "C" int caphn_adam_rank_multi_f32(int R, int njobs, const caphn_rank_job* jobs, const float* coef, const caphn_adam_hparams* hp,
                                         caphn_stream_t stream) {
    if (njobs <= 0 || !jobs || !coef || !hp || hp->step < 1 || R <= 0 || R > RMAX || (hp->zero_gfac && R != 1)) return CAPHN_EINVAL;
    // one launch when the members fit the row kernel's fast path with one width class (and all or none carry the fused GEMV)
    auto qclass = [](int k) { return k <= 256 ? 1 : k <= 512 ? 2 : k <= 1024 ? 4 : 8; };
    bool one = njobs >= 2 && njobs <= 4 && g_tune_adam != 0;
    int maxrows = 0; size_t a_bytes = 0;
    for (int i = 0; i < njobs; ++i) {
        const caphn_rank_job& j = jobs[i];
        if (j.rows <= 0 || j.k <= 0 || !j.W || !j.m || !j.v || !j.gfac || !j.afac) return CAPHN_EINVAL;
        if ((j.next_a != nullptr) != (j.next_theta != nullptr) || (j.next_a != nullptr) != (j.next_bias != nullptr)) return CAPHN_EINVAL;
        if (j.next_pack && (!j.next_a || j.pack_H <= 0 || j.pack_HA <= 0 || j.pack_HA > j.pack_H || j.pack_pitch < j.pack_H || j.pack_hrows <= 0 ||
                            (long)j.rows % ((long)j.pack_H * j.pack_H) != 0)) return CAPHN_EINVAL;
        const bool vec = (j.k % 4 == 0) && (j.lda % 4 == 0) && caphn_aligned16(j.W) && caphn_aligned16(j.m) && caphn_aligned16(j.v) &&
                         caphn_aligned16(j.afac) && (!j.next_a || caphn_aligned16(j.next_a));
        one = one && vec && j.k <= 2048 && qclass(j.k) == qclass(jobs[0].k) && ((j.next_a != nullptr) == (jobs[0].next_a != nullptr)) &&
              j.rows <= 4 * g_tune_adam_cap;
        maxrows = std::max(maxrows, j.rows);
        a_bytes = std::max(a_bytes, sizeof(float) * (size_t)R * j.k);
    }
    if (!one) {
        for (int i = 0; i < njobs; ++i) {
            const caphn_rank_job& j = jobs[i];
            int rc = adam_rank_launch(R, j.rows, j.k, j.W, j.m, j.v, j.gfac, j.ldg, j.afac, j.lda, coef, hp,
                                      NextGemv{j.next_a, j.next_bias, j.next_theta, j.next_pack, j.pack_H, j.pack_HA, j.pack_pitch, j.pack_hrows}, stream);
            if (rc != CAPHN_OK) return rc;
        }
        return CAPHN_OK;
    }
    RankJobs J;
    for (int i = 0; i < njobs; ++i) {
        const caphn_rank_job& j = jobs[i];
        J.j[i] = RankJob{j.W, j.m, j.v, j.gfac, j.ldg, j.afac, j.lda,
                         NextGemv{j.next_a, j.next_bias, j.next_theta, j.next_pack, j.pack_H, j.pack_HA, j.pack_pitch, j.pack_hrows}, j.rows, j.k};
    }
    const int use_lds = R > 1 && a_bytes <= 60 * 1024;
    const size_t shm = use_lds ? a_bytes : 0;
    const AdamK K = make_adam(hp);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int q = qclass(jobs[0].k);
    const bool gemv = jobs[0].next_a != nullptr;
    const int RB = gemv ? (q == 1 ? 4 : 2) : (g_tune_adam == 3 ? 1 : 2);       // as adam_rank_launch picks them
    const dim3 grid((unsigned)((maxrows + 4 * RB - 1) / (4 * RB)), (unsigned)njobs);
#define RANK_JOBS_Q(RBv, Q) do { \
        if (R == 1) hipLaunchKernelGGL((adam_rank_jobs_kernel<RBv, true, true, Q, false>), grid, dim3(256), shm, s, R, J, coef, K, use_lds); \
        else hipLaunchKernelGGL((adam_rank_jobs_kernel<RBv, true, true, Q, true>), grid, dim3(256), shm, s, R, J, coef, K, use_lds); } while (0)
#define RANK_JOBS(RBv) do { if (q == 1) RANK_JOBS_Q(RBv, 1); else if (q == 2) RANK_JOBS_Q(RBv, 2); else if (q == 4) RANK_JOBS_Q(RBv, 4); \
                            else RANK_JOBS_Q(RBv, 8); } while (0)
    if (RB == 4) RANK_JOBS(4); else if (RB == 2) RANK_JOBS(2); else RANK_JOBS(1);
#undef RANK_JOBS
#undef RANK_JOBS_Q
    return caphn_launch_status();
}

extern "C" int caphn_abi_version(void) { return 1; }
extern "C" int caphn_device_arch(char* buf, int buflen) {
    if (!buf || buflen <= 0) return CAPHN_EINVAL;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return CAPHN_ELAUNCH;
    int i = 0;
    for (; i < buflen - 1 && prop.gcnArchName[i] && prop.gcnArchName[i] != ':'; ++i) buf[i] = prop.gcnArchName[i];
    buf[i] = 0;
    return CAPHN_OK;
}

// ---- multi-tensor clip coefficient and Adam (parameters in separate allocations: caphn.optim.FusedAdam) ----
static size_t mt_norm_blocks(int nt, const size_t* n) {
    size_t b = 0;
    for (int i = 0; i < nt; ++i) b += (n[i] + SUMSQ_CHUNK - 1) / SUMSQ_CHUNK;
    return b;
}
extern "C" size_t caphn_grad_norm_multi_workspace_bytes(int ntensors, const size_t* n, int R, int njobs) {
    if (ntensors < 0 || (ntensors > 0 && !n) || R <= 0 || R > RMAX || njobs < 0 || njobs > CAPHN_MAX_HEADS) return 0;
    return sizeof(double) * (mt_norm_blocks(ntensors, n) + (size_t)njobs * 2 * R * R * NORM_CHUNKS);
}
extern "C" int caphn_grad_norm_multi(int ntensors, const float* const* g, const size_t* n, int R, int njobs, const int* rows,
                                     const int* k, const float* const* gfac, const size_t* ldg, const float* const* afac,
                                     const size_t* lda, double max_norm, double scale, float* coef_out, void* ws,
                                     caphn_stream_t stream) {
    if (ntensors < 0 || (ntensors > 0 && (!g || !n)) || R <= 0 || R > RMAX || njobs < 0 || njobs > CAPHN_MAX_HEADS || !coef_out || !ws ||
        ntensors + njobs == 0) return CAPHN_EINVAL;
    GramJobs jobs; jobs.n = njobs;
    for (int i = 0; i < njobs; ++i) {
        if (!rows || !k || !gfac || !ldg || !afac || !lda || rows[i] <= 0 || k[i] <= 0 || !gfac[i] || !afac[i]) return CAPHN_EINVAL;
        jobs.j[i] = GramJob{gfac[i], ldg[i], afac[i], lda[i], rows[i], k[i]};
    }
    for (int i = 0; i < ntensors; ++i) if (!g[i] || n[i] == 0) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nb_total = mt_norm_blocks(ntensors, n);
    const unsigned ngram = (unsigned)(njobs * 2 * R * R * NORM_CHUNKS);
    double* part = static_cast<double*>(ws);          // [nb_total dense slots | gram slots]: what grad_norm_finish_kernel reads
    // launches of at most MT_MAX tensors each; the Gram-dot blocks ride in the last one
    size_t done = 0;
    int t0 = 0;
    for (;;) {
        MTJobs T; T.n = 0;
        unsigned b = 0;
        while (t0 + T.n < ntensors && T.n < MT_MAX) {
            const int i = t0 + T.n;
            T.t[T.n] = MTDesc{nullptr, nullptr, nullptr, g[i], n[i]};
            T.blk0[T.n] = b;
            b += (unsigned)((n[i] + SUMSQ_CHUNK - 1) / SUMSQ_CHUNK);
            ++T.n;
        }
        T.blk0[T.n] = b;
        t0 += T.n;
        const bool last = t0 >= ntensors;
        const unsigned nb = b + (last ? ngram : 0u);
        if (nb > 0)
            hipLaunchKernelGGL(grad_norm_multi_partials_kernel, dim3(nb), dim3(256), 0, s, T, (int)b, R, jobs, part + done, part + nb_total);
        done += b;
        if (last) break;
    }
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(256), 0, s, (int)nb_total, R, njobs, static_cast<const double*>(ws), max_norm,
                       scale, coef_out);
    return caphn_launch_status();
}
extern "C" int caphn_adam_multi_f32(int ntensors, float* const* p, float* const* m, float* const* v, const float* const* g,
                                    const size_t* n, const float* coef, const caphn_adam_hparams* hp, caphn_stream_t stream) {
    if (ntensors <= 0 || !p || !m || !v || !g || !n || !coef || !hp || hp->step < 1) return CAPHN_EINVAL;
    for (int i = 0; i < ntensors; ++i) if (!p[i] || !m[i] || !v[i] || !g[i] || n[i] == 0) return CAPHN_EINVAL;
    const AdamK K = make_adam(hp);
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int t0 = 0; t0 < ntensors;) {
        MTJobs T; T.n = 0;
        unsigned b = 0;
        while (t0 + T.n < ntensors && T.n < MT_MAX) {
            const int i = t0 + T.n;
            T.t[T.n] = MTDesc{p[i], m[i], v[i], g[i], n[i]};
            T.blk0[T.n] = b;
            b += (unsigned)((n[i] + MT_CHUNK - 1) / MT_CHUNK);
            ++T.n;
        }
        T.blk0[T.n] = b;
        t0 += T.n;
        hipLaunchKernelGGL(adam_multi_kernel, dim3(b), dim3(256), 0, s, T, coef, K);
    }
    return caphn_launch_status();
}
extern "C" int caphn_stream_copy_f32(size_t n, const float* src, float* dst, caphn_stream_t stream) {
    if (n == 0 || (n & 3) || !src || !dst || !caphn_aligned16(src) || !caphn_aligned16(dst)) return CAPHN_EINVAL;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream), n >> 2,
                       reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst));
    return caphn_launch_status();
}
// grad_norm_partials + (finish fused into) adam_dense: the front of the fused optimiser in TWO launches (was four, five with the
// cross entropy's loss reduction, which rides along when ce_ws / loss_out are given: ce_rows rows as caphn_cross_entropy_rows left them)
extern "C" int caphn_grad_norm_adam_dense(size_t n, float* p, float* m, float* v, const float* g, int R, int njobs, const int* rows,
                                          const int* k, const float* const* gfac, const size_t* ldg, const float* const* afac,
                                          const size_t* lda, double max_norm, double scale, float* coef_out, void* ws,
                                          const caphn_adam_hparams* hp, int ce_rows, const void* ce_ws, const int* ce_n_valid_dev,
                                          float* loss_out, caphn_stream_t stream) {
    if (n == 0 || !p || !m || !v || !g || R <= 0 || R > RMAX || njobs < 0 || njobs > CAPHN_MAX_HEADS || !coef_out || !ws || !hp ||
        hp->step < 1 || (ce_rows > 0 && (!ce_ws || !loss_out))) return CAPHN_EINVAL;
    GramJobs jobs; jobs.n = njobs;
    for (int i = 0; i < njobs; ++i) {
        if (!rows || !k || !gfac || !ldg || !afac || !lda || rows[i] <= 0 || k[i] <= 0 || !gfac[i] || !afac[i]) return CAPHN_EINVAL;
        jobs.j[i] = GramJob{gfac[i], ldg[i], afac[i], lda[i], rows[i], k[i]};
    }
    const int nb_s = caphn_sumsq_blocks(n);
    const unsigned nb = (unsigned)nb_s + (unsigned)(njobs * 2 * R * R * NORM_CHUNKS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(grad_norm_partials_kernel, dim3(nb), dim3(256), 0, s, n, g, (int)caphn_aligned16(g), nb_s, R, jobs,
                       static_cast<double*>(ws));
    NormFin nf{static_cast<const double*>(ws), nb_s, R, njobs, max_norm, scale, coef_out, ce_rows,
               ce_rows > 0 ? static_cast<const float*>(ce_ws) + 4 : nullptr, static_cast<const float*>(ce_ws), ce_n_valid_dev, loss_out};
    const int vec = caphn_aligned16(p) && caphn_aligned16(m) && caphn_aligned16(v) && caphn_aligned16(g);
    size_t nbk = (n / 4 + 255) / 256;
    if (nbk > (size_t)g_tune_adam_dense_cap) nbk = (size_t)g_tune_adam_dense_cap;
    if (nbk < 1) nbk = 1;
    hipLaunchKernelGGL(adam_dense_clip_kernel, dim3((unsigned)nbk), dim3(256), 0, s, n, p, m, v, g, nf, make_adam(hp), vec);
    return caphn_launch_status();
}
