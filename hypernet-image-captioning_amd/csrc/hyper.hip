// Hypernetwork forward / backward for ONE input row (M = 1): every Linear is a GEMV.
// Reference: hypernet_attention.py:55-99 (shapes), :111-118 (forward).
//
// The two large second-layer matrices (240000x480 and 120000x240 fp32 = 576 MB at the canonical
// size) dominate: each is streamed from HBM exactly once per pass with 16-byte coalesced loads, one
// wave per row (a 480-float row is 1920 contiguous bytes = two wave-wide dwordx4 loads), several
// rows in flight per wave, wave-shuffle reduction.  Nothing here is MFMA-shaped: it is HBM-bound.
#include "common.h"
#include <algorithm>

namespace {

constexpr float LRELU = 0.01f;   // nn.LeakyReLU() default slope
__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : LRELU * v; }
__device__ __forceinline__ float lrelu_grad(float post) { return post > 0.f ? 1.f : LRELU; }

struct GemvJob {
    const float* W; const float* b; const float* x; float* y;
    int rows, k, act, vec;   // act: 1 = LeakyReLU; vec: 16-byte loads legal
    int block0, nblocks;
    float* xcopy;            // optional: the job's first block also stores x[0..k) here (the saved input of the backward)
};
struct GemvJobs { GemvJob j[CAPHN_MAX_HEADS]; int n; };

// ---------------------------------------------------------------- forward: y = act(W x + b)
// wave per row, RB rows per iteration, QMAX dwordx4 per lane per row (k <= 256*QMAX): RB*QMAX
// independent 16-byte loads in flight per lane.  RB == 8 reduces the 8 row sums together (three
// halving exchange stages then three plain ones: 10 shuffles per 8 rows); RB == 4 reduces row by row.
// NTL: non-temporal loads.  Variants are selected by caphn_tune (measured A/B).
template <int QMAX, int RB, bool NTL>
__device__ __forceinline__ void gemv_rows_wave(const GemvJob& J, int wave_g, int nwaves, int lane) {
    const int k4 = J.k >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(J.x);
    f32x4 xr[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        int c = lane + 64 * q;
        xr[q] = c < k4 ? x4[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int myrow = RB == 8 ? ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1) : lane;
    for (int r0 = wave_g * RB; r0 < J.rows; r0 += nwaves * RB) {
        f32x4 w[RB][QMAX];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const f32x4* row = reinterpret_cast<const f32x4*>(J.W + (size_t)(r0 + i) * J.k);
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                int c = lane + 64 * q;
                if (r0 + i < J.rows && c < k4) w[i][q] = NTL ? __builtin_nontemporal_load(row + c) : row[c];
                else w[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        float v[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < QMAX; ++q)
                s += w[i][q][0] * xr[q][0] + w[i][q][1] * xr[q][1] + w[i][q][2] * xr[q][2] + w[i][q][3] * xr[q][3];
            v[i] = s;
        }
        float mine = 0.f;
        if (RB == 8) {
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const int mask = 32 >> st, half = RB >> (st + 1);
                const bool hi = (lane & mask) != 0;
#pragma unroll
                for (int j = 0; j < half; ++j) {
                    const float send = hi ? v[j] : v[j + half];
                    const float keep = hi ? v[j + half] : v[j];
                    v[j] = keep + __shfl_xor(send, mask, 64);
                }
            }
            float s = v[0];
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 1, 64);
            mine = s;
        } else {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const float s = wave_sum(v[i]);
                if (lane == i) mine = s;
            }
        }
        const bool writer = RB == 8 ? (lane & 7) == 0 : lane < RB;
        if (writer && r0 + myrow < J.rows) {
            float o = mine + (J.b ? J.b[r0 + myrow] : 0.f);
            J.y[r0 + myrow] = J.act ? lrelu(o) : o;
        }
    }
}

// 8 lanes per row: any k, vector or scalar loads
__device__ __forceinline__ void gemv_rows_oct(const GemvJob& J, int lb, int tid) {
    const int grp = tid >> 3, s = tid & 7;
    for (int r = lb * 32 + grp; r < J.rows; r += J.nblocks * 32) {
        const float* row = J.W + (size_t)r * J.k;
        float sum = 0.f;
        if (J.vec) {
            const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
            const f32x4* x4 = reinterpret_cast<const f32x4*>(J.x);
            for (int c = s; c < (J.k >> 2); c += 8) {
                f32x4 a = r4[c], b = x4[c];
                sum += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
            }
        } else {
            for (int c = s; c < J.k; c += 8) sum += row[c] * J.x[c];
        }
        sum += __shfl_xor(sum, 4, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 1, 64);
        if (s == 0) {
            float v = sum + (J.b ? J.b[r] : 0.f);
            J.y[r] = J.act ? lrelu(v) : v;
        }
    }
}

// long rows of any width / alignment (hypernet.py's heads: k = 11250, 8437): one wave per row, lane-contiguous dword
// loads (256 B per wave-instruction whatever the row's alignment), 8 of W and 8 of x in flight per lane; x comes from
// L2 (45 KB does not fit the L1).  The 8-lanes-per-row fallback below reads 32 B per row per instruction and reached
// 2.9 TB/s on the 11 GB hypernet; this form does not depend on k % 4.
__device__ __forceinline__ void gemv_rows_long(const GemvJob& J, int lb, int tid) {
    const int lane = tid & 63, wave_g = lb * 4 + (tid >> 6), nwaves = J.nblocks * 4;
    for (int r = wave_g; r < J.rows; r += nwaves) {
        const float* row = J.W + (size_t)r * J.k;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int c = lane;
        for (; c + 64 * 7 < J.k; c += 64 * 8) {
            float w[8], x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { w[u] = __builtin_nontemporal_load(row + c + 64 * u); x[u] = J.x[c + 64 * u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += w[u] * x[u];
        }
        for (; c < J.k; c += 64) s[0] += __builtin_nontemporal_load(row + c) * J.x[c];
        float t = wave_sum(((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])));
        if (lane == 0) {
            const float v = t + (J.b ? J.b[r] : 0.f);
            J.y[r] = J.act ? lrelu(v) : v;
        }
    }
}

// QMAX (chunks of 256 columns per row) is a KERNEL template parameter picked on the host from the widest job:
// with all widths inlined into one kernel the register allocation is that of QMAX = 8 (185-256 VGPRs, 1-2
// waves/SIMD) whatever k is; per-width kernels need ~40-70 VGPRs and keep 8 waves/SIMD in flight.
template <int RB, bool NTL, int QMAX>
__global__ __launch_bounds__(256) void gemv_fwd_kernel(GemvJobs jobs) {
    int ji = 0;
    for (int i = 1; i < jobs.n; ++i) if ((int)blockIdx.x >= jobs.j[i].block0) ji = i;
    const GemvJob& J = jobs.j[ji];
    const int lb = blockIdx.x - J.block0, tid = threadIdx.x;
    if (J.xcopy && lb == 0)
        for (int i = tid; i < J.k; i += 256) J.xcopy[i] = J.x[i];
    if (J.vec && J.k >= 128 && J.k <= 256 * QMAX) {
        const int wave_g = lb * 4 + (tid >> 6), nwaves = J.nblocks * 4, lane = tid & 63;
        gemv_rows_wave<QMAX, RB, NTL>(J, wave_g, nwaves, lane);
    } else if (J.k >= 1024) {
        gemv_rows_long(J, lb, tid);
    } else {
        gemv_rows_oct(J, lb, tid);
    }
}

// ---------------------------------------------------------------- hn_base and the heads' first layers in ONE launch
// x -> a0 = lrelu(Wb0 x + b) -> base = lrelu(Wb2 a0 + b) -> a_i = lrelu(W1_i base + b1_i): three dependent GEMVs of 40-220 k
// multiply-adds each.  As three launches they are 20 us of kernels and two launch gaps on the optimiser's chain, between the dense
// Adam launch and the first rank-1 pass (which needs a_i of the NEXT step).  Here every workgroup computes a0 and base for itself
// in LDS (80 k multiply-adds, 320 KB of L2 reads per workgroup: nothing to exchange, no barrier between workgroups) and then its
// share of the heads' rows; workgroup 0 also stores x, a0 and base.  Per row the arithmetic is gemv_rows_wave's (lane partials
// over 16-byte chunks in order, wave_sum): results are bit-identical to the three-launch form.
int g_tune_acts_fused = 1;      // caphn_tune key 28: 1 (default) = one launch, 0 = three launches.  With every load of a phase in flight
                                // (RB = 16) the launch is -10 us per step, alternating blocks in one process (tools/ab_inproc.py, two boxes);
                                // its first form (four rows per iteration, 25 us alone) was +4
struct ActsFusedArgs {
    const float* x; int d_in, d_mid, he, nh, rtot;
    const float* w0; const float* b0; const float* w2; const float* b2;
    const float* w1[CAPHN_MAX_HEADS]; const float* b1[CAPHN_MAX_HEADS]; float* out[CAPHN_MAX_HEADS]; int k[CAPHN_MAX_HEADS];
    float* a_x; float* a_a0; float* a_base;
};
struct RowRef { const float* w; const float* b; float* og; int r; };
// RB rows per wave iteration: rows wv, wv + nw, ...; `in` (LDS) has kin floats; result to out_s (LDS, may be null) / the row's og.
// RB = 16 (widths <= 256) / 8: with 16 waves a 200-row layer is ONE iteration -- every load of the phase in flight at once (four
// rows per iteration were four dependent L2 round trips per phase: the kernel took 25 us alone)
template <int QMAX, typename Locate>
__device__ __forceinline__ void fused_rows(Locate locate, int nrows, int kin, const float* in, int wv, int nw, float* out_s, int lane) {
    constexpr int RB = QMAX == 1 ? 16 : 8;
    const int k4 = kin >> 2;
    f32x4 xr[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        const int c = lane + 64 * q;
        xr[q] = c < k4 ? reinterpret_cast<const f32x4*>(in)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int r0 = wv; r0 < nrows; r0 += nw * RB) {
        // (wv is wave-uniform and comes in through readfirstlane: row references live in scalar registers and are formed again at
        //  the store -- sixteen of them held in vector registers spilled)
        f32x4 w[RB][QMAX];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int r = r0 + i * nw;
            const RowRef rr = r < nrows ? locate(r) : RowRef{nullptr, nullptr, nullptr, 0};
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                const int c = lane + 64 * q;
                w[i][q] = (rr.w && c < k4) ? reinterpret_cast<const f32x4*>(rr.w + (size_t)rr.r * kin)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int r = r0 + i * nw;
            if (r >= nrows) continue;               // (wave-uniform)
            float sacc = 0.f;
#pragma unroll
            for (int q = 0; q < QMAX; ++q)
                sacc += w[i][q][0] * xr[q][0] + w[i][q][1] * xr[q][1] + w[i][q][2] * xr[q][2] + w[i][q][3] * xr[q][3];
            const float t = wave_sum(sacc);
            if (lane == 0) {
                const RowRef rr = locate(r);
                const float v = lrelu(t + rr.b[rr.r]);
                if (out_s) out_s[r] = v;
                if (rr.og) rr.og[rr.r] = v;
            }
        }
    }
}
template <int QMAX>
__global__ __launch_bounds__(1024) void hyper_acts_fused_kernel(ActsFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;
    float* a0s = xs + ((a.d_in + 3) & ~3);
    float* bs = a0s + ((a.d_mid + 3) & ~3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wg = blockIdx.x;
    const bool w0g = wg == 0;
    for (int i = tid; i < a.d_in; i += 1024) {
        const float v = a.x[i];
        xs[i] = v;
        if (w0g && a.a_x != a.x) a.a_x[i] = v;
    }
    __syncthreads();
    fused_rows<QMAX>([&](int r) { return RowRef{a.w0, a.b0, w0g ? a.a_a0 : nullptr, r}; }, a.d_mid, a.d_in, xs, wave, 16, a0s, lane);
    __syncthreads();
    fused_rows<QMAX>([&](int r) { return RowRef{a.w2, a.b2, w0g ? a.a_base : nullptr, r}; }, a.he, a.d_mid, a0s, wave, 16, bs, lane);
    __syncthreads();
    fused_rows<QMAX>([&](int r) {
        int i = 0;
        while (i + 1 < a.nh && r >= a.k[i]) { r -= a.k[i]; ++i; }
        return RowRef{a.w1[i], a.b1[i], a.out[i], r};
    }, a.rtot, a.he, bs, wg * 16 + wave, (int)gridDim.x * 16, nullptr, lane);
}
}  // namespace
int g_tune_gemv = 1;      // 0: plain loads  1 (default): non-temporal loads (115 vs 126 us on the canonical hypernet)
namespace {
template <int QMAX>
static void launch_gemv_fwd_q(const GemvJobs& jobs, int nblocks, hipStream_t s) {
    if (g_tune_gemv == 0) hipLaunchKernelGGL((gemv_fwd_kernel<4, false, QMAX>), dim3(nblocks), dim3(256), 0, s, jobs);
    else hipLaunchKernelGGL((gemv_fwd_kernel<4, true, QMAX>), dim3(nblocks), dim3(256), 0, s, jobs);
}
static void launch_gemv_fwd(const GemvJobs& jobs, int nblocks, hipStream_t s) {
    int kmax = 0;
    for (int i = 0; i < jobs.n; ++i) if (jobs.j[i].vec && jobs.j[i].k <= 2048) kmax = std::max(kmax, jobs.j[i].k);
    if (kmax <= 256) launch_gemv_fwd_q<1>(jobs, nblocks, s);
    else if (kmax <= 512) launch_gemv_fwd_q<2>(jobs, nblocks, s);
    else if (kmax <= 1024) launch_gemv_fwd_q<4>(jobs, nblocks, s);
    else launch_gemv_fwd_q<8>(jobs, nblocks, s);
}

// ---------------------------------------------------------------- backward: y = W^T d (W [rows,k])
// large matrices: lanes own column chunks, waves own rows; per-block partial sums -> ws, then reduce
struct GemvTJob {
    const float* W; const float* d; float* partial;   // partial [nblocks, k]
    int rows, k, vec, block0, nblocks;
};
struct GemvTJobs { GemvTJob j[CAPHN_MAX_HEADS]; int n; unsigned* bar = nullptr; };    // bar: the fused tail's phase counters, cleared here

template <int QMAX>
__device__ __forceinline__ void gemv_t_wave(const GemvTJob& J, int lb, int tid, float* red /* [4][k] */) {
    const int lane = tid & 63, wave = tid >> 6;
    const int k4 = J.k >> 2;
    f32x4 acc[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int RB = 4;
    const int wave_g = lb * 4 + wave, nwaves = J.nblocks * 4;
    for (int r0 = wave_g * RB; r0 < J.rows; r0 += nwaves * RB) {
        f32x4 w[RB][QMAX];
        float dv[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const bool ok = r0 + i < J.rows;
            dv[i] = ok ? J.d[r0 + i] : 0.f;
            const f32x4* row = reinterpret_cast<const f32x4*>(J.W + (size_t)(r0 + i) * J.k);
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                int c = lane + 64 * q;
                w[i][q] = (ok && c < k4) ? __builtin_nontemporal_load(row + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int q = 0; q < QMAX; ++q) acc[q] += w[i][q] * dv[i];
    }
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        int c = lane + 64 * q;
        if (c < k4) *reinterpret_cast<f32x4*>(red + wave * J.k + c * 4) = acc[q];
    }
    __syncthreads();
    for (int c = tid; c < J.k; c += 256)
        J.partial[(size_t)lb * J.k + c] = red[c] + red[J.k + c] + red[2 * J.k + c] + red[3 * J.k + c];
}

template <int QMAX>
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(GemvTJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    int ji = 0;
    for (int i = 1; i < jobs.n; ++i) if ((int)blockIdx.x >= jobs.j[i].block0) ji = i;
    const GemvTJob& J = jobs.j[ji];
    const int lb = blockIdx.x - J.block0, tid = threadIdx.x;
    if (jobs.bar && blockIdx.x == 0 && tid < 4) jobs.bar[tid] = 0u;
    if (J.vec && J.k <= 256 * QMAX) {
        gemv_t_wave<QMAX>(J, lb, tid, red);
    } else {
        // any width / alignment: this block's row range, 256 x CJ columns at a time -- CJ independent dword loads per
        // thread and row (lane-contiguous, 1 KB per wave instruction group), instead of one dependent load per row
        // (1.8 TB/s on hypernet.py's 11250- and 8437-wide heads)
        const int per = (J.rows + J.nblocks - 1) / J.nblocks;
        const int ra = lb * per, rb = min(J.rows, ra + per);
        constexpr int CJ = 8;
        for (int c0 = tid; c0 < J.k; c0 += 256 * CJ) {
            float s[CJ];
#pragma unroll
            for (int j = 0; j < CJ; ++j) s[j] = 0.f;
            const bool full = c0 + 256 * (CJ - 1) < J.k;
            if (full) {
                for (int r = ra; r < rb; ++r) {
                    const float dr = J.d[r];
                    const float* row = J.W + (size_t)r * J.k + c0;
#pragma unroll
                    for (int j = 0; j < CJ; ++j) s[j] += row[256 * j] * dr;
                }
            } else {
                for (int r = ra; r < rb; ++r) {
                    const float dr = J.d[r];
                    const float* row = J.W + (size_t)r * J.k + c0;
#pragma unroll
                    for (int j = 0; j < CJ; ++j) if (c0 + 256 * j < J.k) s[j] += row[256 * j] * dr;
                }
            }
#pragma unroll
            for (int j = 0; j < CJ; ++j) if (c0 + 256 * j < J.k) J.partial[(size_t)lb * J.k + c0 + 256 * j] = s[j];
        }
    }
}

// Column reductions with 1024-thread blocks laid out as 64 columns x 16 row lanes: every thread
// strides over rows, the 16 partial sums of a column meet in LDS.  (A thread-per-column loop over
// hundreds of rows is a chain of dependent-latency loads: 100-300 us for a few hundred KB.)
constexpr int RL = 16;
__device__ __forceinline__ float col_reduce_finish(float acc, float (*red)[65]) {
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    red[rl][c] = acc;
    __syncthreads();
    float s = 0.f;
    if (rl == 0) {
#pragma unroll
        for (int i = 0; i < RL; ++i) s += red[i][c];
    }
    return s;
}

// out[c] = (sum_blocks partial[blk][c]) * lrelu'(post[c]); written to up to two sinks
struct ReduceJob { const float* partial; const float* post; float* out0; float* out1; int k, nblocks; };
struct ReduceJobs { ReduceJob j[CAPHN_MAX_HEADS]; int n; };
__global__ __launch_bounds__(1024) void gemv_t_reduce_kernel(ReduceJobs jobs) {
    __shared__ float red[RL][65];
    const ReduceJob& J = jobs.j[blockIdx.y];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    if (blockIdx.x * 64 >= J.k) return;
    float acc = 0.f;
    if (c < J.k)
        for (int b = rl; b < J.nblocks; b += RL) acc += J.partial[(size_t)b * J.k + c];
    float s = col_reduce_finish(acc, red);
    if (rl == 0 && c < J.k) {
        if (J.post) s *= lrelu_grad(J.post[c]);
        if (J.out0) J.out0[c] = s;
        if (J.out1) J.out1[c] = s;
    }
}

// small matrices: y[c] = (sum_jobs sum_r W_j[r][c] d_j[r]) * lrelu'(post[c])
struct SmallTJob { const float* W; const float* d; int rows; };
struct SmallTArgs { SmallTJob j[CAPHN_MAX_HEADS]; int n; int k; const float* post; float* out0; float* out1; int acc0 = 0; };
// 1024 threads = 8 columns x 128 row lanes (was 64 x 16: with k = 200 that is four workgroups, each thread walking 70
// rows one dependent load at a time -- 17-43 us for 0.9 MB, on the tail of the hypernet VJP branch)
constexpr int STC = 8, STR = 128;
__global__ __launch_bounds__(1024) void gemv_t_small_kernel(SmallTArgs a) {
    __shared__ float red[STR][STC + 1];
    const int cl = threadIdx.x & (STC - 1), rl = threadIdx.x / STC;
    const int c = blockIdx.x * STC + cl;
    float acc = 0.f;
    if (c < a.k)
        for (int i = 0; i < a.n; ++i) {
            const SmallTJob& J = a.j[i];
            for (int r = rl; r < J.rows; r += STR) acc += J.W[(size_t)r * a.k + c] * J.d[r];
        }
    red[rl][cl] = acc;
    __syncthreads();
    // 128 partial sums per column: one wave folds them (lane = row lane, two per lane), shuffle reduction per column
    if (threadIdx.x < 64 * STC) {
        const int col = threadIdx.x >> 6, lane = threadIdx.x & 63;
        float s = wave_sum(red[lane][col] + red[lane + 64][col]);
        const int cc = blockIdx.x * STC + col;
        if (lane == 0 && cc < a.k) {
            if (a.post) s *= lrelu_grad(a.post[cc]);
            if (a.out0) { if (a.acc0) atomicAdd(a.out0 + cc, s); else a.out0[cc] = s; }
            if (a.out1) a.out1[cc] = s;
        }
    }
}

// dense outer products out_j[r][c] = g_j[r] * a_j[c]
struct OuterJob { const float* g; const float* a; float* out; int rows, k; long block0; };
struct OuterJobs { OuterJob j[2 * CAPHN_MAX_HEADS]; int n; };
__global__ __launch_bounds__(256) void outer_kernel(OuterJobs jobs) {
    int ji = 0;
    for (int i = 1; i < jobs.n; ++i) if ((long)blockIdx.x >= jobs.j[i].block0) ji = i;
    const OuterJob& J = jobs.j[ji];
    const size_t n = (size_t)J.rows * J.k;
    size_t idx = ((size_t)blockIdx.x - J.block0) * 1024 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i, idx += 256)
        if (idx < n) J.out[idx] = J.g[idx / J.k] * J.a[idx % J.k];
}


// ---------------------------------------------------------------- fused tail of the backward (one launch)
// Behind the transposed GEMV the VJP needs four tiny dependent steps -- reduce the per-block partials to dz_i, dbase = sum_i
// W1_i^T dz_i, da0 = Wb2^T dzb2, dx = Wb0^T dzb0 -- and the rank-1 weight gradients of the small layers: five launches, 1 MB of
// traffic, 60-75 us on the chain to the optimiser when the chip is busy with the weight-gradient GEMMs.  Here they are phases of
// ONE launch of <= 64 workgroups separated by counter barriers.  Inter-workgroup data (dz, dzb2, dzb0: a few KB) is written with
// agent-scope (sc1, write-through) stores, drained by every storing wave (s_waitcnt vmcnt(0)) in front of the workgroup barrier
// that precedes the workgroup's ONE counter add, and read with agent-scope loads behind the poll + workgroup barrier
// (MI355X_MICROARCH.md, hand-off table, first row) -- no fences, so no write-back of whatever the GEMMs running beside this
// kernel have dirtied in the L2.  Every workgroup must become resident for the barriers to complete: the grid is at most 64
// workgroups and nothing it waits for depends on this stream; the polls are bounded in wall-clock time like the pair kernels'
// (device error word, CAPHN_ETIMEOUT).  Summation orders are fixed: results are reproducible run to run.
int g_tune_vjp_blocks = 512;    // caphn_tune key 30: workgroups per head of the transposed GEMV (64 .. 4096)
int g_tune_hyper_tail = 0;      // caphn_tune key 27: 1 = this kernel, 0 (default) = the five-launch form (measured equal in the step,
                                // 29 us against 31 us + four gaps alone: the branches behind BPTT are throughput-bound)
struct TailHead { const float* partial; const float* post; const float* w1; float* g_b1; float* g_w1; int k, nblocks, dzo; };
struct TailArgs {
    TailHead h[CAPHN_MAX_HEADS]; int nh;
    int he, d_mid, d_in;
    const float* base_act;      // acts: hn_base output (post-activation) -- the column factor of g_w1 and the mask of dzb2
    const float* a0_act;        // acts: hn_base hidden layer (post-activation)
    const float* x_in;          // acts: the input row
    const float* base_w2; const float* base_w0;
    float* g_base_b2; float* g_base_b0; float* g_base_w2; float* g_base_w0; float* g_x; int x_acc;
    float* dz; float* dzb2; float* dzb0;       // workspace (exchanged between workgroups)
    unsigned* bar;
    int* err; long long limit;
};
__device__ __forceinline__ float ld_x(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_x(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tail_barrier(unsigned* cnt, unsigned target, int* err, long long limit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        for (int spin = 1; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin) {
            __builtin_amdgcn_s_sleep(1);
            if ((spin & 255) == 0 && wall_clock64() - t0 > limit) {
                if (err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
}
// The exchanged vector (dz, dzb2 or dzb0: produced by OTHER workgroups of this launch) is brought into LDS once per phase with
// agent-scope loads, all in flight at once; the phases' loops then run on plain loads.  (Agent-scope loads inside the loops were
// kept in program order by the compiler: nine dependent L2 round trips per thread and phase -- the kernel took 38 us alone.)
__device__ __forceinline__ void tail_stage(const float* src, int n, float* dst) {
    for (int i = threadIdx.x; i < n; i += 1024) dst[i] = ld_x(src + i);
    __syncthreads();
}
// y[c] = (sum_r W[r][c] d[r]) * lrelu'(post[c]) for the 8 columns of `unit`, d in LDS (rows entries; W row-major [rows, k])
__device__ __forceinline__ void tail_smallt(const float* W, const float* ds, int rows, int k, const float* post, float* out_x, float* out_g,
                                            int acc_g, int unit, float (*red)[STC + 1]) {
    const int cl = threadIdx.x & (STC - 1), rl = threadIdx.x / STC;
    const int c = unit * STC + cl;
    float acc = 0.f;
    if (c < k) {
        int r = rl;
        for (; r + 3 * STR < rows; r += 4 * STR) {           // four independent loads in flight
            const float w0 = W[(size_t)r * k + c], w1 = W[(size_t)(r + STR) * k + c], w2 = W[(size_t)(r + 2 * STR) * k + c],
                        w3 = W[(size_t)(r + 3 * STR) * k + c];
            acc += w0 * ds[r]; acc += w1 * ds[r + STR]; acc += w2 * ds[r + 2 * STR]; acc += w3 * ds[r + 3 * STR];
        }
        for (; r < rows; r += STR) acc += W[(size_t)r * k + c] * ds[r];
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (threadIdx.x < 64 * STC) {
        const int col = threadIdx.x >> 6, lane = threadIdx.x & 63;
        float s = wave_sum(red[lane][col] + red[lane + 64][col]);
        const int cc = unit * STC + col;
        if (lane == 0 && cc < k) {
            if (post) s *= lrelu_grad(post[cc]);
            if (out_x) st_x(out_x + cc, s);
            if (out_g) { if (acc_g) atomicAdd(out_g + cc, s); else out_g[cc] = s; }
        }
    }
    __syncthreads();
}
// out[r][c] = g[r] * a[c] for the 4096 elements of `unit`; g in LDS
__device__ __forceinline__ void tail_outer(const float* gs, const float* a, float* out, int rows, int k, int unit) {
    const size_t n = (size_t)rows * k;
    size_t idx = (size_t)unit * 4096 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i, idx += 1024)
        if (idx < n) out[idx] = gs[idx / k] * a[idx % k];
}
__global__ __launch_bounds__(1024) void hyper_tail_kernel(TailArgs a) {
    __shared__ float red[STR][STC + 1];            // 1152 floats; phase 1 uses it as [RL][65] (1040)
    extern __shared__ __attribute__((aligned(16))) float vec_s[];      // the phase's exchanged vector: max(sum k_i, he, d_mid) floats
    const int wg = blockIdx.x, NB = gridDim.x, tid = threadIdx.x;
    {   // phase 1: dz_i[c] = (sum_b partial_i[b][c]) * lrelu'(a_i[c]); units of 64 columns dealt round-robin
        float (*r16)[65] = reinterpret_cast<float (*)[65]>(&red[0][0]);
        int u0 = 0;
        for (int i = 0; i < a.nh; ++i) {
            const TailHead& H = a.h[i];
            const int nu = (H.k + 63) / 64;
            for (int u = 0; u < nu; ++u) {
                if ((u0 + u) % NB != wg) continue;
                const int c = u * 64 + (tid & 63), rl = tid >> 6;
                float acc = 0.f;
                if (c < H.k) {
                    int b = rl;
                    for (; b + 7 * RL < H.nblocks; b += 8 * RL) {        // eight independent loads in flight, added in order
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = H.partial[(size_t)(b + q * RL) * H.k + c];
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc += v[q];
                    }
                    for (; b < H.nblocks; b += RL) acc += H.partial[(size_t)b * H.k + c];
                }
                r16[rl][tid & 63] = acc;
                __syncthreads();
                if (rl == 0 && c < H.k) {
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < RL; ++q) s += r16[q][tid & 63];
                    s *= lrelu_grad(H.post[c]);
                    st_x(a.dz + H.dzo + c, s);
                    if (H.g_b1) H.g_b1[c] = s;
                }
                __syncthreads();
            }
            u0 += nu;
        }
    }
    tail_barrier(a.bar + 0, NB, a.err, a.limit);
    {   // phase 2: dzb2 = (sum_i W1_i^T dz_i) * lrelu'(base);  g_w1_i = dz_i (x) base
        const TailHead& HL = a.h[a.nh - 1];
        const int ktot = HL.dzo + HL.k;                  // dz_i are laid out back to back (4-float aligned starts)
        tail_stage(a.dz, ktot, vec_s);
        const int nu = (a.he + STC - 1) / STC;
        for (int u = wg; u < nu; u += NB) {
            // (the heads' first layers W1_i [k_i, he] one after the other: same column c, rows continue through the heads)
            const int cl = tid & (STC - 1), rl = tid / STC, c = u * STC + cl;
            float acc = 0.f;
            if (c < a.he)
                for (int i = 0; i < a.nh; ++i) {
                    const float* W = a.h[i].w1; const float* ds = vec_s + a.h[i].dzo; const int rows = a.h[i].k, k = a.he;
                    int r = rl;
                    for (; r + 3 * STR < rows; r += 4 * STR) {
                        const float w0 = W[(size_t)r * k + c], w1 = W[(size_t)(r + STR) * k + c], w2 = W[(size_t)(r + 2 * STR) * k + c],
                                    w3 = W[(size_t)(r + 3 * STR) * k + c];
                        acc += w0 * ds[r]; acc += w1 * ds[r + STR]; acc += w2 * ds[r + 2 * STR]; acc += w3 * ds[r + 3 * STR];
                    }
                    for (; r < rows; r += STR) acc += W[(size_t)r * k + c] * ds[r];
                }
            red[rl][cl] = acc;
            __syncthreads();
            if (tid < 64 * STC) {
                const int col = tid >> 6, lane = tid & 63;
                float sres = wave_sum(red[lane][col] + red[lane + 64][col]);
                const int cc = u * STC + col;
                if (lane == 0 && cc < a.he) {
                    sres *= lrelu_grad(a.base_act[cc]);
                    st_x(a.dzb2 + cc, sres);
                    if (a.g_base_b2) a.g_base_b2[cc] = sres;
                }
            }
            __syncthreads();
        }
        int v0 = 0;                                  // outer-product units go to the workgroups from the far end
        for (int i = 0; i < a.nh; ++i) {
            const int nv = (int)(((size_t)a.h[i].k * a.he + 4095) / 4096);
            if (a.h[i].g_w1)
                for (int v = 0; v < nv; ++v)
                    if ((v0 + v) % NB == NB - 1 - wg) tail_outer(vec_s + a.h[i].dzo, a.base_act, a.h[i].g_w1, a.h[i].k, a.he, v);
            v0 += nv;
        }
    }
    tail_barrier(a.bar + 1, NB, a.err, a.limit);
    {   // phase 3: dzb0 = (Wb2^T dzb2) * lrelu'(a0);  g_base_w2 = dzb2 (x) a0
        tail_stage(a.dzb2, a.he, vec_s);
        const int nu = (a.d_mid + STC - 1) / STC;
        for (int u = wg; u < nu; u += NB) tail_smallt(a.base_w2, vec_s, a.he, a.d_mid, a.a0_act, a.dzb0, a.g_base_b0, 0, u, red);
        if (a.g_base_w2) {
            const int nv = (int)(((size_t)a.he * a.d_mid + 4095) / 4096);
            for (int v = 0; v < nv; ++v)
                if (v % NB == NB - 1 - wg) tail_outer(vec_s, a.a0_act, a.g_base_w2, a.he, a.d_mid, v);
        }
    }
    if (!a.g_x && !a.g_base_w0) return;
    tail_barrier(a.bar + 2, NB, a.err, a.limit);
    {   // phase 4: dx = Wb0^T dzb0;  g_base_w0 = dzb0 (x) x
        tail_stage(a.dzb0, a.d_mid, vec_s);
        if (a.g_x) {
            const int nu = (a.d_in + STC - 1) / STC;
            for (int u = wg; u < nu; u += NB) tail_smallt(a.base_w0, vec_s, a.d_mid, a.d_in, nullptr, nullptr, a.g_x, a.x_acc, u, red);
        }
        if (a.g_base_w0) {
            const int nv = (int)(((size_t)a.d_mid * a.d_in + 4095) / 4096);
            for (int v = 0; v < nv; ++v)
                if (v % NB == NB - 1 - wg) tail_outer(vec_s, a.x_in, a.g_base_w0, a.d_mid, a.d_in, v);
        }
    }
}
inline int gemv_blocks(int rows, int k, int vec) {
    // enough waves to cover the chip; big jobs grid-stride
    long want = (vec && k >= 128) ? ((long)rows + 15) / 16 : ((long)rows + 31) / 32;
    if (want < 1) want = 1;
    if (want > 2048) want = 2048;
    return (int)want;
}

inline int d_in(const caphn_hyper_desc* d) { return d->d_in > 0 ? d->d_in : d->he; }
inline int d_mid(const caphn_hyper_desc* d) { return d->d_mid > 0 ? d->d_mid : d->he; }
struct ActsLayout { int x, a0, base, a[CAPHN_MAX_HEADS], total; };
inline ActsLayout acts_layout(const caphn_hyper_desc* d) {
    ActsLayout L;
    // every segment starts 16-byte aligned so the GEMVs may use dwordx4 loads of their input
    auto up4 = [](int v) { return (v + 3) & ~3; };
    int o = 0;
    L.x = o; o += up4(d_in(d)); L.a0 = o; o += up4(d_mid(d)); L.base = o; o += up4(d->he);
    for (int i = 0; i < d->n_heads; ++i) { L.a[i] = o; o += up4(d->k[i]); }
    L.total = o;
    return L;
}
inline bool desc_ok(const caphn_hyper_desc* d) {
    if (!d || d->he <= 0 || d->n_heads <= 0 || d->n_heads > CAPHN_MAX_HEADS || d->d_in < 0 || d->d_mid < 0) return false;
    if (!d->base_w0 || !d->base_b0 || !d->base_w2 || !d->base_b2) return false;
    for (int i = 0; i < d->n_heads; ++i)
        if (d->k[i] <= 0 || d->w[i] <= 0 || !d->w1[i] || !d->b1[i] || !d->w2[i] || !d->b2[i]) return false;
    return true;
}
inline int vec_ok(const float* W, const float* x, int k) {
    return (k % 4 == 0) && caphn_aligned16(W) && caphn_aligned16(x);
}

}  // namespace

extern "C" int caphn_hyper_acts_floats(const caphn_hyper_desc* d) {
    if (!d) return CAPHN_EINVAL;
    return acts_layout(d).total;
}

static int hyper_forward_impl(const caphn_hyper_desc* d, const float* x, float* theta, float* acts, caphn_stream_t stream);
extern "C" int caphn_hyper_forward(const caphn_hyper_desc* d, const float* x, float* theta, float* acts,
                                   caphn_stream_t stream) {
    if (!theta) return CAPHN_EINVAL;
    return hyper_forward_impl(d, x, theta, acts, stream);
}
extern "C" int caphn_hyper_forward_acts(const caphn_hyper_desc* d, const float* x, float* acts, caphn_stream_t stream) {
    return hyper_forward_impl(d, x, nullptr, acts, stream);
}
static int hyper_forward_impl(const caphn_hyper_desc* d, const float* x, float* theta, float* acts, caphn_stream_t stream) {
    if (!desc_ok(d) || !x || !acts) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const ActsLayout L = acts_layout(d);
    // (x is kept in acts for the backward: the first GEMV's first block stores it -- a hipMemcpyAsync here cost a
    //  runtime blit kernel plus ~10-30 us of queue bubble in front of the optimiser's rank-1 passes)
    auto one = [&](const float* W, const float* b, const float* in, float* out, int rows, int k, int act, float* xcopy) {
        GemvJobs jobs; jobs.n = 1;
        GemvJob& J = jobs.j[0];
        J.W = W; J.b = b; J.x = in; J.y = out; J.rows = rows; J.k = k; J.act = act; J.xcopy = xcopy;
        J.vec = vec_ok(W, in, k); J.block0 = 0; J.nblocks = gemv_blocks(rows, k, J.vec);
        launch_gemv_fwd(jobs, J.nblocks, s);
    };
    // the three small layers in one launch when every width fits the wave-per-row 16-byte path (the canonical sizes do)
    bool fused = g_tune_acts_fused != 0;
    {
        const int widths[3] = {d_in(d), d_mid(d), d->he};
        for (int wdt : widths) fused = fused && wdt % 4 == 0 && wdt >= 128 && wdt <= 512;
        fused = fused && (long)d_mid(d) * d_in(d) + (long)d->he * d_mid(d) <= 262144 && caphn_aligned16(x) && caphn_aligned16(acts) &&
                caphn_aligned16(d->base_w0) && caphn_aligned16(d->base_w2);
        for (int i = 0; i < d->n_heads; ++i) fused = fused && caphn_aligned16(d->w1[i]);
    }
    if (fused) {
        ActsFusedArgs a;
        a.x = x; a.d_in = d_in(d); a.d_mid = d_mid(d); a.he = d->he; a.nh = d->n_heads; a.rtot = 0;
        a.w0 = d->base_w0; a.b0 = d->base_b0; a.w2 = d->base_w2; a.b2 = d->base_b2;
        for (int i = 0; i < d->n_heads; ++i) { a.w1[i] = d->w1[i]; a.b1[i] = d->b1[i]; a.out[i] = acts + L.a[i]; a.k[i] = d->k[i]; a.rtot += d->k[i]; }
        a.a_x = acts + L.x; a.a_a0 = acts + L.a0; a.a_base = acts + L.base;
        const int nb = std::max(1, std::min(64, (a.rtot + 15) / 16));
        const size_t shm = sizeof(float) * (size_t)(((a.d_in + 3) & ~3) + ((a.d_mid + 3) & ~3) + ((a.he + 3) & ~3));
        const int wmax = std::max(a.d_in, std::max(a.d_mid, a.he));
        if (wmax <= 256) hipLaunchKernelGGL(hyper_acts_fused_kernel<1>, dim3(nb), dim3(1024), shm, s, a);
        else hipLaunchKernelGGL(hyper_acts_fused_kernel<2>, dim3(nb), dim3(1024), shm, s, a);
    } else {
    one(d->base_w0, d->base_b0, x, acts + L.a0, d_mid(d), d_in(d), 1, x == acts + L.x ? nullptr : acts + L.x);
    one(d->base_w2, d->base_b2, acts + L.a0, acts + L.base, d->he, d_mid(d), 1, nullptr);
    {   // first layers of all heads in one launch
        GemvJobs jobs; jobs.n = d->n_heads; int b0 = 0;
        for (int i = 0; i < d->n_heads; ++i) {
            GemvJob& J = jobs.j[i];
            J.W = d->w1[i]; J.b = d->b1[i]; J.x = acts + L.base; J.y = acts + L.a[i];
            J.rows = d->k[i]; J.k = d->he; J.act = 1; J.vec = vec_ok(J.W, J.x, J.k); J.xcopy = nullptr;
            J.block0 = b0; J.nblocks = gemv_blocks(J.rows, J.k, J.vec); b0 += J.nblocks;
        }
        launch_gemv_fwd(jobs, b0, s);
    }
    }
    if (theta) {   // second layers: theta = cat_i (W2_i a_i + b2_i)
        GemvJobs jobs; jobs.n = d->n_heads; int b0 = 0; size_t off = 0;
        for (int i = 0; i < d->n_heads; ++i) {
            GemvJob& J = jobs.j[i];
            J.W = d->w2[i]; J.b = d->b2[i]; J.x = acts + L.a[i]; J.y = theta + off; off += d->w[i];
            J.rows = d->w[i]; J.k = d->k[i]; J.act = 0; J.vec = vec_ok(J.W, J.x, J.k); J.xcopy = nullptr;
            J.block0 = b0; J.nblocks = gemv_blocks(J.rows, J.k, J.vec); b0 += J.nblocks;
        }
        launch_gemv_fwd(jobs, b0, s);
    }
    return caphn_launch_status();
}

namespace {
struct BwdWs { size_t partial[CAPHN_MAX_HEADS]; int nblocks[CAPHN_MAX_HEADS]; size_t dz[CAPHN_MAX_HEADS];
               size_t dzb2, dzb0, bar, total; };
constexpr int VJP_BLOCKS_MAX = 4096;
inline BwdWs bwd_ws(const caphn_hyper_desc* d) {
    BwdWs w; size_t o = 0;
    for (int i = 0; i < d->n_heads; ++i) {
        // (the offsets reserve room for VJP_BLOCKS_MAX blocks whatever caphn_tune key 30 says now: a workspace sized earlier stays valid)
        const long want = ((long)d->w[i] + 15) / 16;
        int nb = (int)std::min<long>(g_tune_vjp_blocks, want);
        if (nb < 1) nb = 1;
        w.nblocks[i] = nb;
        w.partial[i] = o; o += caphn_align_up((size_t)std::max<long>(1, std::min<long>(VJP_BLOCKS_MAX, want)) * d->k[i], 4);
    }
    for (int i = 0; i < d->n_heads; ++i) { w.dz[i] = o; o += caphn_align_up(d->k[i], 4); }
    w.dzb2 = o; o += caphn_align_up(d->he, 4);
    w.dzb0 = o; o += caphn_align_up(d_mid(d), 4);
    o = caphn_align_up(o, 32);
    w.bar = o; o += 32;               // the fused tail's phase counters, on a 128-byte line of their own
    w.total = o;
    return w;
}
}  // namespace

extern "C" size_t caphn_hyper_backward_workspace_bytes(const caphn_hyper_desc* d) {
    if (!d || d->n_heads <= 0 || d->n_heads > CAPHN_MAX_HEADS) return 0;
    return bwd_ws(d).total * sizeof(float);
}

extern "C" int caphn_hyper_backward(const caphn_hyper_desc* d, const float* dtheta, const float* acts,
                                    const caphn_hyper_grads* g, void* ws_, caphn_stream_t stream) {
    if (!desc_ok(d) || !dtheta || !acts || !g || !ws_) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const ActsLayout L = acts_layout(d);
    const BwdWs W = bwd_ws(d);
    float* ws = static_cast<float*>(ws_);
    const int nh = d->n_heads;
    size_t toff[CAPHN_MAX_HEADS]; { size_t o = 0; for (int i = 0; i < nh; ++i) { toff[i] = o; o += d->w[i]; } }

    // bias grads of the second layers are dtheta itself
    for (int i = 0; i < nh; ++i)
        if (g->g_b2[i] && g->g_b2[i] != dtheta + toff[i])
            if (hipMemcpyAsync(g->g_b2[i], dtheta + toff[i], sizeof(float) * d->w[i], hipMemcpyDeviceToDevice, s) != hipSuccess)
                return CAPHN_ELAUNCH;

    {   // da_i = W2_i^T dtheta_i  (streams the big matrices once), partial sums per block
        GemvTJobs jobs; jobs.n = nh; int b0 = 0; int kmax = 0;
        for (int i = 0; i < nh; ++i) {
            GemvTJob& J = jobs.j[i];
            J.W = d->w2[i]; J.d = dtheta + toff[i]; J.partial = ws + W.partial[i];
            J.rows = d->w[i]; J.k = d->k[i]; J.vec = (J.k % 4 == 0) && caphn_aligned16(J.W);
            J.block0 = b0; J.nblocks = W.nblocks[i]; b0 += J.nblocks;
            kmax = std::max(kmax, J.k);
        }
        size_t vec_floats = std::max(d->he, d_mid(d));
        { size_t kt = 0; for (int i = 0; i < nh; ++i) kt += caphn_align_up(d->k[i], 4); vec_floats = std::max(vec_floats, kt); }
        const bool fused_tail = g_tune_hyper_tail != 0 && vec_floats * sizeof(float) <= 64 * 1024;
        if (fused_tail) jobs.bar = reinterpret_cast<unsigned*>(ws + W.bar);
        int kv = 0;
        for (int i = 0; i < nh; ++i) if (jobs.j[i].vec && jobs.j[i].k <= 2048) kv = std::max(kv, jobs.j[i].k);
        const size_t shm = sizeof(float) * 4 * (kv > 0 ? kv : 1);      // only the wave path (k <= 2048, aligned) uses LDS
        if (kv <= 256) hipLaunchKernelGGL(gemv_t_partial_kernel<1>, dim3(b0), dim3(256), shm, s, jobs);
        else if (kv <= 512) hipLaunchKernelGGL(gemv_t_partial_kernel<2>, dim3(b0), dim3(256), shm, s, jobs);
        else if (kv <= 1024) hipLaunchKernelGGL(gemv_t_partial_kernel<4>, dim3(b0), dim3(256), shm, s, jobs);
        else hipLaunchKernelGGL(gemv_t_partial_kernel<8>, dim3(b0), dim3(256), shm, s, jobs);
        if (fused_tail) {
            TailArgs a;
            a.nh = nh; a.he = d->he; a.d_mid = d_mid(d); a.d_in = d_in(d);
            a.base_act = acts + L.base; a.a0_act = acts + L.a0; a.x_in = acts + L.x;
            a.base_w2 = d->base_w2; a.base_w0 = d->base_w0;
            a.g_base_b2 = g->g_base_b2; a.g_base_b0 = g->g_base_b0; a.g_base_w2 = g->g_base_w2; a.g_base_w0 = g->g_base_w0;
            a.g_x = g->g_x; a.x_acc = g->x_accumulate;
            a.dz = ws + W.dz[0]; a.dzb2 = ws + W.dzb2; a.dzb0 = ws + W.dzb0;
            a.bar = reinterpret_cast<unsigned*>(ws + W.bar);
            a.err = caphn_errword(); a.limit = g_tune_xch_timeout;
            int units = (d->he + STC - 1) / STC, p1 = 0, outer = 0;
            for (int i = 0; i < nh; ++i) {
                TailHead& H = a.h[i];
                H.partial = ws + W.partial[i]; H.post = acts + L.a[i]; H.w1 = d->w1[i]; H.g_b1 = g->g_b1[i]; H.g_w1 = g->g_w1[i];
                H.k = d->k[i]; H.nblocks = W.nblocks[i]; H.dzo = (int)(W.dz[i] - W.dz[0]);
                p1 += (H.k + 63) / 64;
                if (H.g_w1) outer += (int)(((size_t)H.k * d->he + 4095) / 4096);
            }
            units = std::max(units + outer, std::max(p1, std::max((d_mid(d) + STC - 1) / STC, (d_in(d) + STC - 1) / STC)));
            const int nb = std::max(1, std::min(64, units));
            hipLaunchKernelGGL(hyper_tail_kernel, dim3(nb), dim3(1024), vec_floats * sizeof(float), s, a);
            for (int i = 0; i < nh; ++i)     // second-layer weight grads only when asked for (dense 576 MB at the canonical size)
                if (g->g_w2[i]) {
                    OuterJobs o2; o2.n = 1;
                    OuterJob& J = o2.j[0];
                    J.g = dtheta + toff[i]; J.a = acts + L.a[i]; J.out = g->g_w2[i]; J.rows = d->w[i]; J.k = d->k[i]; J.block0 = 0;
                    long nbo = ((long)J.rows * J.k + 1023) / 1024;
                    hipLaunchKernelGGL(outer_kernel, dim3((unsigned)nbo), dim3(256), 0, s, o2);
                }
            return caphn_launch_status();
        }
        // dz_i = da_i * lrelu'(a_i)  (also the first-layer bias grad)
        ReduceJobs rj; rj.n = nh;
        for (int i = 0; i < nh; ++i) {
            ReduceJob& R = rj.j[i];
            R.partial = ws + W.partial[i]; R.post = acts + L.a[i]; R.out0 = ws + W.dz[i]; R.out1 = g->g_b1[i];
            R.k = d->k[i]; R.nblocks = W.nblocks[i];
        }
        hipLaunchKernelGGL(gemv_t_reduce_kernel, dim3((kmax + 63) / 64, nh), dim3(1024), 0, s, rj);
    }
    {   // dbase = sum_i W1_i^T dz_i ; dzb2 = dbase * lrelu'(base)
        SmallTArgs a; a.n = nh; a.k = d->he; a.post = acts + L.base; a.out0 = ws + W.dzb2; a.out1 = g->g_base_b2;
        for (int i = 0; i < nh; ++i) { a.j[i].W = d->w1[i]; a.j[i].d = ws + W.dz[i]; a.j[i].rows = d->k[i]; }
        hipLaunchKernelGGL(gemv_t_small_kernel, dim3((d->he + STC - 1) / STC), dim3(1024), 0, s, a);
    }
    {   // da0 = Wb2^T dzb2 ; dzb0 = da0 * lrelu'(a0)
        SmallTArgs a; a.n = 1; a.k = d_mid(d); a.post = acts + L.a0; a.out0 = ws + W.dzb0; a.out1 = g->g_base_b0;
        a.j[0].W = d->base_w2; a.j[0].d = ws + W.dzb2; a.j[0].rows = d->he;
        hipLaunchKernelGGL(gemv_t_small_kernel, dim3((d_mid(d) + STC - 1) / STC), dim3(1024), 0, s, a);
    }
    if (g->g_x) {   // dx = Wb0^T dzb0
        SmallTArgs a; a.n = 1; a.k = d_in(d); a.post = nullptr; a.out0 = g->g_x; a.out1 = nullptr; a.acc0 = g->x_accumulate;
        a.j[0].W = d->base_w0; a.j[0].d = ws + W.dzb0; a.j[0].rows = d_mid(d);
        hipLaunchKernelGGL(gemv_t_small_kernel, dim3((d_in(d) + STC - 1) / STC), dim3(1024), 0, s, a);
    }
    {   // dense weight grads (rank-1 outer products)
        OuterJobs oj; oj.n = 0; long b0 = 0;
        auto add = [&](const float* gv, const float* av, float* out, int rows, int k) {
            if (!out) return;
            OuterJob& J = oj.j[oj.n++];
            J.g = gv; J.a = av; J.out = out; J.rows = rows; J.k = k; J.block0 = b0;
            b0 += ((long)rows * k + 1023) / 1024;
        };
        add(ws + W.dzb0, acts + L.x, g->g_base_w0, d_mid(d), d_in(d));
        add(ws + W.dzb2, acts + L.a0, g->g_base_w2, d->he, d_mid(d));
        for (int i = 0; i < nh; ++i) add(ws + W.dz[i], acts + L.base, g->g_w1[i], d->k[i], d->he);
        if (oj.n) hipLaunchKernelGGL(outer_kernel, dim3((unsigned)b0), dim3(256), 0, s, oj);
        // second-layer weight grads only when asked for (dense 576 MB at the canonical size)
        for (int i = 0; i < nh; ++i)
            if (g->g_w2[i]) {
                OuterJobs o2; o2.n = 1;
                OuterJob& J = o2.j[0];
                J.g = dtheta + toff[i]; J.a = acts + L.a[i]; J.out = g->g_w2[i]; J.rows = d->w[i]; J.k = d->k[i]; J.block0 = 0;
                long nb = ((long)J.rows * J.k + 1023) / 1024;
                hipLaunchKernelGGL(outer_kernel, dim3((unsigned)nb), dim3(256), 0, s, o2);
            }
    }
    return caphn_launch_status();
}

extern "C" int caphn_outer_f32(int rows, int k, const float* gv, const float* av, float* out, caphn_stream_t stream) {
    if (rows <= 0 || k <= 0 || !gv || !av || !out) return CAPHN_EINVAL;
    OuterJobs o2; o2.n = 1;
    OuterJob& J = o2.j[0];
    J.g = gv; J.a = av; J.out = out; J.rows = rows; J.k = k; J.block0 = 0;
    long nb = ((long)rows * k + 1023) / 1024;
    hipLaunchKernelGGL(outer_kernel, dim3((unsigned)nb), dim3(256), 0, static_cast<hipStream_t>(stream), o2);
    return caphn_launch_status();
}

// Tuning knob for the microbenchmark (tools/microbench_stream.py); not part of the stable ABI surface
// beyond its declaration.  key 0: forward GEMV variant, key 1: rank-Adam variant (misc.hip),
// key 2: GEMM back end (0 fp32 MFMA, 1 split-bf16 MFMA).
extern int g_tune_adam;
extern int g_tune_gemm;
extern int g_tune_rec_rotate;
extern int g_tune_fork;
extern int g_tune_gemm_xcd;
extern int g_tune_gemm_fast;
extern int g_tune_gemm_planes;
extern int g_tune_rec_pair;
extern int g_tune_gemm_single;
extern int g_tune_gemm_tile;
extern int g_tune_deterministic;
extern int g_tune_adam_cap;
extern int g_tune_chain_main;
extern int g_tune_rec_cache;
extern int g_tune_splitk_target;
extern int g_tune_gemm_order;
extern int g_tune_branch_mask;
extern int g_tune_gemm_db;
extern int g_tune_gemm_ws;
extern int g_tune_adam_dense_cap;
extern int g_tune_sk_dhs, g_tune_sk_vocab_w;
extern int g_tune_join_chain;
extern int g_tune_gemm_kres;
extern int g_tune_gemm_waves;
extern int g_tune_hops;
extern int g_tune_vocab_order;
extern int g_det_vocab;
int caphn_rec_pair_debug_skip(int v);
int caphn_rec_pair_debug_opts(int v);
extern "C" int caphn_tune(int key, int value) {
    if (key == 0) { g_tune_gemv = value; return CAPHN_OK; }
    if (key == 1) { g_tune_adam = value; return CAPHN_OK; }
    if (key == 2) { g_tune_gemm = value; return CAPHN_OK; }
    if (key == 3) { g_tune_rec_rotate = value; return CAPHN_OK; }
    if (key == 4) { g_tune_fork = value; return CAPHN_OK; }
    if (key == 6) { g_tune_gemm_xcd = value; return CAPHN_OK; }
    if (key == 7) { g_tune_gemm_fast = value; return CAPHN_OK; }
    if (key == 8) { g_tune_gemm_planes = value; return CAPHN_OK; }
    if (key == 9) { g_tune_rec_pair = value; return CAPHN_OK; }
    if (key == 10) return caphn_rec_pair_debug_skip(value);
    if (key == 11) { g_tune_gemm_single = value; return CAPHN_OK; }
    if (key == 12) { g_tune_gemm_tile = value; return CAPHN_OK; }
    if (key == 13) { g_tune_deterministic = value > 0; g_det_vocab = value; return CAPHN_OK; }
    if (key == 21) { g_tune_vocab_order = value != 0; return CAPHN_OK; }
    if (key == 20) { if (value < 0 || value > 7) return CAPHN_EINVAL; g_tune_branch_mask = value; return CAPHN_OK; }
    if (key == 18) { if (value < 0 || value > 2) return CAPHN_EINVAL; g_tune_gemm_order = value; return CAPHN_OK; }
    if (key == 17) { if (value < 1 || value > 65536) return CAPHN_EINVAL; g_tune_splitk_target = value; return CAPHN_OK; }
    if (key == 16) { if (value < 0 || value > 2) return CAPHN_EINVAL; g_tune_rec_cache = value; return CAPHN_OK; }
    if (key == 15) { g_tune_chain_main = value != 0; return CAPHN_OK; }
    if (key == 26) { g_tune_hops = value != 0; return CAPHN_OK; }
    if (key == 27) { g_tune_hyper_tail = value != 0; return CAPHN_OK; }
    if (key == 28) { g_tune_acts_fused = value != 0; return CAPHN_OK; }
    if (key == 36) { g_tune_gemm_kres = value != 0; return CAPHN_OK; }
    if (key == 35) { g_tune_join_chain = value != 0; return CAPHN_OK; }
    if (key == 33) { if (value < 0 || value > 64) return CAPHN_EINVAL; g_tune_sk_dhs = value; return CAPHN_OK; }
    if (key == 34) { if (value < 0 || value > 64) return CAPHN_EINVAL; g_tune_sk_vocab_w = value; return CAPHN_OK; }
    if (key == 30) { if (value < 64 || value > 4096) return CAPHN_EINVAL; g_tune_vjp_blocks = value; return CAPHN_OK; }
    if (key == 31) { if (value < 64 || value > 65535) return CAPHN_EINVAL; g_tune_adam_dense_cap = value; return CAPHN_OK; }
    if (key == 29) { if (value != 0 && value != 64 && value != 128) return CAPHN_EINVAL; g_tune_gemm_ws = value; return CAPHN_OK; }
    if (key == 25) { if (value != 0 && value != 5 && value != 6) return CAPHN_EINVAL; g_tune_gemm_waves = value; return CAPHN_OK; }
    if (key == 24) return caphn_rec_pair_debug_opts(value);
    if (key == 23) { if (value < 0 || value > 7) return CAPHN_EINVAL; g_tune_gemm_db = value; return CAPHN_OK; }
    if (key == 22) { if (value < 1) return CAPHN_EINVAL; g_tune_xch_timeout = 100ll * value; return CAPHN_OK; }
    if (key == 14) { if (value < 64 || value > 65535) return CAPHN_EINVAL; g_tune_adam_cap = value; return CAPHN_OK; }
    return CAPHN_EINVAL;
}
