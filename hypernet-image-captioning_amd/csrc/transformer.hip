// N4 (CATR, baseline/transformer.py): the two non-GEMM pieces of a transformer layer.
//   * LayerNorm forward / backward          (nn.LayerNorm: baseline/transformer.py:19,27,138-139,199-201,281)
//   * multi-head attention core             (nn.MultiheadAttention inside :137,197-199: softmax(q k^T / sqrt(dh) + masks) v)
// Projections and feed-forward layers are caphn_gemm_f32.  Everything is fp32.
//
// Attention is tiny next to the layer's GEMMs (d = 256, 8 heads of 32, 49-128 positions: 16 MFLOP per sample and layer against
// 270 MFLOP in the feed-forward), so it runs on the VALU with one side of the product resident in LDS and one wave per row of
// the other side; the three kernels (forward, dq, dk/dv) are the same skeleton with the roles of the sides exchanged, which
// keeps the backward free of atomics (deterministic) at the price of recomputing the probabilities twice.
#include "common.h"
#include <algorithm>

namespace {

// ---------------------------------------------------------------------------------------------- LayerNorm
// one wave per row; NJ columns per lane (d <= 64 * NJ)
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(int rows, int d, const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63, wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int r = wave_g; r < rows; r += nwaves) {
        const float* xr = x + (size_t)r * d;
        float v[NJ];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int c = lane + 64 * j; v[j] = c < d ? xr[c] : 0.f; s += v[j]; }
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int c = lane + 64 * j; const float t = c < d ? v[j] - mean : 0.f; q += t * t; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);          // biased variance, as torch
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            if (c < d) y[(size_t)r * d + c] = (v[j] - mean) * rstd * gamma[c] + beta[c];
        }
        if (lane == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-block partial sums of dy * xhat and dy
template <int NJ>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(int rows, int d, const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ partial /* [gridDim.x][2d] */) {
    __shared__ float red[4][64 * NJ * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wave_g = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    float ag[NJ], ab[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
    for (int r = wave_g; r < rows; r += nwaves) {
        const float m = mean[r], rs = rstd[r];
        float xh[NJ], g[NJ];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            const bool ok = c < d;
            const float dyv = ok ? dy[(size_t)r * d + c] : 0.f;
            xh[j] = ok ? (x[(size_t)r * d + c] - m) * rs : 0.f;
            g[j] = ok ? dyv * gamma[c] : 0.f;
            s1 += g[j]; s2 += g[j] * xh[j];
            ag[j] += dyv * xh[j]; ab[j] += dyv;
        }
        s1 = wave_sum(s1) / (float)d; s2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = lane + 64 * j;
            if (c < d) dx[(size_t)r * d + c] = rs * (g[j] - s1 - xh[j] * s2);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { red[wave][lane + 64 * j] = ag[j]; red[wave][64 * NJ + lane + 64 * j] = ab[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 64 * NJ; i += 256) {
        const int half = i / (64 * NJ), c = i - half * 64 * NJ;
        if (c < d) partial[(size_t)blockIdx.x * 2 * d + half * d + c] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

// ---------------------------------------------------------------------------------------------- attention
struct AttnArgs {
    caphn_attn_dims d;
    const float* q; const float* k; const float* v; const float* o; const float* dO;
    const float* attn_mask; const unsigned char* key_padding;
    float* out0; float* out1;          // fwd: o, lse   dq: dq, -   dkv: dk, dv
    const float* lse; const float* D;  // backward inputs
    int rpb;                           // rows of the wave side per block
};
// Every block stages the whole LDS side of its (batch, head) (32 KB at 128 positions), so the wave side should not be cut
// finer than the chip needs: 16 rows per block made the staging the dominant cost (attention 7.7 ms of a 34 ms CATR step)
constexpr int MAX_RPB = 64;
inline int attn_rows_per_block(int bh, int rows, int most = 32) {
    // backward kernels: 64 rows push a 128-position block past 48 KB of LDS (2 instead of 4 blocks per CU), measured slower;
    // the forward kernel stages less per row and takes 64
    int rpb = most;
    while (rpb > 8 && (long)bh * ((rows + rpb - 1) / rpb) < 1024) rpb >>= 1;
    return rpb;
}

__device__ __forceinline__ size_t at(int t, int b, int h, int ldt, int ldb, int dh) { return (size_t)t * ldt + (size_t)b * ldb + (size_t)h * dh; }

// LDS rows have a pitch of DH + 4 floats: 16-byte aligned for ds_read_b128, and consecutive rows start 4 banks apart, so
// the 8 (16) lanes that read one row as float4 and the 8 (4) row groups of a wave instruction tile the 32 banks exactly.
// (With scalar reads at pitch DH + 1 every FMA cost one to two LDS instructions and the kernels ran at a third of this.)
template <int DH> struct Pitch { static constexpr int v = DH + 4; };

// dot product of a register vector with LDS row j
template <int DH>
__device__ __forceinline__ float dot_row(const float (&r)[DH], const float* __restrict__ S, int j) {
    const f32x4* p = reinterpret_cast<const f32x4*>(S + (size_t)j * Pitch<DH>::v);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e4 = 0; e4 < DH / 4; e4 += 2) {
        const f32x4 a = p[e4], b = p[e4 + 1];
        s0 += r[4 * e4] * a[0] + r[4 * e4 + 1] * a[1] + r[4 * e4 + 2] * a[2] + r[4 * e4 + 3] * a[3];
        s1 += r[4 * e4 + 4] * b[0] + r[4 * e4 + 5] * b[1] + r[4 * e4 + 6] * b[2] + r[4 * e4 + 7] * b[3];
    }
    return s0 + s1;
}

// sum over the LDS side: out[e] = sum_j w[j] * S[j][e].  Lanes = (e4, part): DH/4 lanes read one row as float4, 64/(DH/4)
// rows per wave instruction.  Every lane returns the complete sums of its four columns 4*e4 .. 4*e4+3.
template <int DH>
__device__ __forceinline__ f32x4 weighted_rows(const float* __restrict__ w, const float* __restrict__ S, int n, int lane) {
    constexpr int E4 = DH / 4, PARTS = 64 / E4;
    const int e4 = lane % E4, part = lane / E4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = part; j < n; j += PARTS) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(S + (size_t)j * Pitch<DH>::v + 4 * e4);
        const float wj = w[j];
        acc[0] += wj * x[0]; acc[1] += wj * x[1]; acc[2] += wj * x[2]; acc[3] += wj * x[3];
    }
#pragma unroll
    for (int m = E4; m < 64; m <<= 1) {
        acc[0] += __shfl_xor(acc[0], m, 64); acc[1] += __shfl_xor(acc[1], m, 64);
        acc[2] += __shfl_xor(acc[2], m, 64); acc[3] += __shfl_xor(acc[3], m, 64);
    }
    return acc;
}
// lanes 0 .. DH/4-1 store the four columns they own
template <int DH>
__device__ __forceinline__ void store_cols(float* __restrict__ dst, const f32x4& v, int dh, int lane) {
    if (lane < DH / 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (4 * lane + i < dh) dst[4 * lane + i] = v[i];
    }
}

// MODE 0: forward (wave side q, LDS side K, V)   MODE 1: dq (wave side q, dO; LDS side K, V)
template <int DH, int MODE>
__global__ __launch_bounds__(256) void attn_q_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const caphn_attn_dims& d = a.d;
    const int bh = blockIdx.y, b = bh / d.nh, h = bh - b * d.nh;
    const int tk = d.tk, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PT = Pitch<DH>::v;
    float* Ks = lds; float* Vs = Ks + (size_t)tk * PT; float* ps = Vs + (size_t)tk * PT + (size_t)wave * tk;
    // the block's own rows: read per row from global memory they put a 1-2 us load in front of every row of every wave
    float* Rq = Vs + (size_t)tk * PT + (size_t)4 * tk; float* Rg = Rq + a.rpb * DH; float* RD = Rg + a.rpb * DH;   // Rg, RD: dq only
    const int r0 = blockIdx.x * a.rpb, nr = min(a.rpb, d.tq - r0);
    for (int i = threadIdx.x; MODE == 1 && i < nr * DH; i += 256) {
        const int rl = i / DH, e = i - rl * DH;
        const bool ok = e < d.dh;
        Rq[i] = ok ? a.q[at(r0 + rl, b, h, d.q_ldt, d.q_ldb, d.dh) + e] : 0.f;
        Rg[i] = ok ? a.dO[at(r0 + rl, b, h, d.o_ldt, d.o_ldb, d.dh) + e] : 0.f;
    }
    if (MODE == 1) {
        for (int rl = threadIdx.x; rl < nr; rl += 256) {
            const float* gp = a.dO + at(r0 + rl, b, h, d.o_ldt, d.o_ldb, d.dh);
            const float* op = a.o + at(r0 + rl, b, h, d.o_ldt, d.o_ldb, d.dh);
            float sD = 0.f;
            for (int e = 0; e < d.dh; ++e) sD += gp[e] * op[e];
            RD[rl] = sD;
        }
    }
    for (int i = threadIdx.x; i < tk * DH; i += 256) {
        const int j = i / DH, e = i - j * DH;
        const bool ok = e < d.dh;
        Ks[j * PT + e] = ok ? a.k[at(j, b, h, d.k_ldt, d.k_ldb, d.dh) + e] : 0.f;
        Vs[j * PT + e] = ok ? a.v[at(j, b, h, d.v_ldt, d.v_ldb, d.dh) + e] : 0.f;
    }
    __syncthreads();
    const int r_end = min(d.tq, (int)(blockIdx.x + 1) * a.rpb);
    for (int r = blockIdx.x * a.rpb + wave; r < r_end; r += 4) {
        float qr[DH], gr[DH];
        const int rl = r - r0;
        if (MODE == 0) {       // wave-uniform address: the compiler keeps the row in SGPRs (scalar loads), cheaper than LDS here
            const float* qp = a.q + at(r, b, h, d.q_ldt, d.q_ldb, d.dh);
#pragma unroll
            for (int e = 0; e < DH; ++e) qr[e] = e < d.dh ? qp[e] : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < DH; ++e) qr[e] = Rq[rl * DH + e];
        }
        float Dr = 0.f, lse = 0.f;
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < DH; ++e) gr[e] = Rg[rl * DH + e];
            Dr = RD[rl];
            lse = a.lse[(size_t)bh * d.tq + r];
        }
        float mx = -INFINITY;
        for (int j = lane; j < tk; j += 64) {
            float s = dot_row<DH>(qr, Ks, j) * d.scale;
            if (a.attn_mask) s += a.attn_mask[(size_t)r * tk + j];
            if (a.key_padding && a.key_padding[(size_t)b * tk + j]) s = -INFINITY;
            if (MODE == 0) { ps[j] = s; mx = fmaxf(mx, s); }
            else {
                const float p = lse == -INFINITY ? 0.f : caphn_exp(s - lse);
                float dp = dot_row<DH>(gr, Vs, j);
                if (d.dropout_p > 0.f)
                    dp *= caphn_keep_scale(d.seed, ((unsigned long long)bh * d.tq + r) * tk + j, d.dropout_p, 1.0f / (1.0f - d.dropout_p));
                ps[j] = p * (dp - Dr) * d.scale;                       // ds
            }
        }
        if (MODE == 0) {
            mx = wave_max(mx);
            const float m0 = mx == -INFINITY ? 0.f : mx;               // a fully masked row yields zeros (torch: NaN)
            float l = 0.f;
            for (int j = lane; j < tk; j += 64) { const float p = caphn_exp(ps[j] - m0); ps[j] = p; l += p; }
            l = wave_sum(l);
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            const bool drop = d.dropout_p > 0.f;
            const float ik = drop ? 1.0f / (1.0f - d.dropout_p) : 1.f;
            const unsigned long long row_idx = ((unsigned long long)bh * d.tq + r) * tk;
            for (int j = lane; j < tk; j += 64) ps[j] *= drop ? inv * caphn_keep_scale(d.seed, row_idx + j, d.dropout_p, ik) : inv;
            __builtin_amdgcn_wave_barrier();
            const f32x4 oe = weighted_rows<DH>(ps, Vs, tk, lane);
            store_cols<DH>(a.out0 + at(r, b, h, d.o_ldt, d.o_ldb, d.dh), oe, d.dh, lane);
            if (lane == 0) a.out1[(size_t)bh * d.tq + r] = l > 0.f ? m0 + logf(l) : -INFINITY;
            __builtin_amdgcn_wave_barrier();            // ps is rewritten for the wave's next row
        } else {
            __builtin_amdgcn_wave_barrier();
            const f32x4 dq = weighted_rows<DH>(ps, Ks, tk, lane);
            store_cols<DH>(a.out0 + at(r, b, h, d.q_ldt, d.q_ldb, d.dh), dq, d.dh, lane);
        }
    }
}

// dk / dv: wave side = key rows, LDS side = Q, dO (+ lse, D per query)
template <int DH>
__global__ __launch_bounds__(256) void attn_kv_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const caphn_attn_dims& d = a.d;
    const int bh = blockIdx.y, b = bh / d.nh, h = bh - b * d.nh;
    const int tq = d.tq, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PT = Pitch<DH>::v;
    float* Qs = lds; float* Gs = Qs + (size_t)tq * PT;
    float* lse_s = Gs + (size_t)tq * PT; float* D_s = lse_s + tq;
    float* ps = D_s + tq + (size_t)wave * 2 * tq; float* dss = ps + tq;
    float* Rk = D_s + tq + (size_t)8 * tq; float* Rv = Rk + a.rpb * DH;
    const int j0 = blockIdx.x * a.rpb, nj = min(a.rpb, d.tk - j0);
    for (int i = threadIdx.x; i < nj * DH; i += 256) {
        const int jl = i / DH, e = i - jl * DH;
        const bool ok = e < d.dh;
        Rk[i] = ok ? a.k[at(j0 + jl, b, h, d.k_ldt, d.k_ldb, d.dh) + e] : 0.f;
        Rv[i] = ok ? a.v[at(j0 + jl, b, h, d.v_ldt, d.v_ldb, d.dh) + e] : 0.f;
    }
    for (int i = threadIdx.x; i < tq * DH; i += 256) {
        const int r = i / DH, e = i - r * DH;
        const bool ok = e < d.dh;
        Qs[r * PT + e] = ok ? a.q[at(r, b, h, d.q_ldt, d.q_ldb, d.dh) + e] : 0.f;
        Gs[r * PT + e] = ok ? a.dO[at(r, b, h, d.o_ldt, d.o_ldb, d.dh) + e] : 0.f;
    }
    for (int r = threadIdx.x; r < tq; r += 256) {
        const float* gp = a.dO + at(r, b, h, d.o_ldt, d.o_ldb, d.dh);
        const float* op = a.o + at(r, b, h, d.o_ldt, d.o_ldb, d.dh);
        float s = 0.f;
        for (int e = 0; e < d.dh; ++e) s += gp[e] * op[e];
        D_s[r] = s; lse_s[r] = a.lse[(size_t)bh * tq + r];
    }
    __syncthreads();
    const int j_end = min(d.tk, (int)(blockIdx.x + 1) * a.rpb);
    for (int j = blockIdx.x * a.rpb + wave; j < j_end; j += 4) {
        float kr[DH], vr[DH];
        const int jl = j - j0;
#pragma unroll
        for (int e = 0; e < DH; ++e) { kr[e] = Rk[jl * DH + e]; vr[e] = Rv[jl * DH + e]; }
        const bool padded = a.key_padding && a.key_padding[(size_t)b * d.tk + j];
        for (int r = lane; r < tq; r += 64) {
            float s = dot_row<DH>(kr, Qs, r) * d.scale;
            const float dp = dot_row<DH>(vr, Gs, r);
            if (a.attn_mask) s += a.attn_mask[(size_t)r * d.tk + j];
            const float l = lse_s[r];
            const float p = (padded || l == -INFINITY || s == -INFINITY) ? 0.f : caphn_exp(s - l);
            const float m = d.dropout_p > 0.f
                ? caphn_keep_scale(d.seed, ((unsigned long long)bh * tq + r) * d.tk + j, d.dropout_p, 1.0f / (1.0f - d.dropout_p)) : 1.f;
            ps[r] = p * m;                                             // dropped probabilities, for dV
            dss[r] = p * (dp * m - D_s[r]) * d.scale;
        }
        __builtin_amdgcn_wave_barrier();
        const f32x4 dv = weighted_rows<DH>(ps, Gs, tq, lane);
        const f32x4 dk = weighted_rows<DH>(dss, Qs, tq, lane);
        store_cols<DH>(a.out1 + at(j, b, h, d.v_ldt, d.v_ldb, d.dh), dv, d.dh, lane);
        store_cols<DH>(a.out0 + at(j, b, h, d.k_ldt, d.k_ldb, d.dh), dk, d.dh, lane);
    }
}

constexpr size_t ATTN_LDS_MAX = 156 * 1024;
inline int attn_dh_class(int dh) { return dh <= 32 ? 32 : (dh <= 64 ? 64 : 0); }
inline size_t attn_q_lds(int side, int DH, int rpb = MAX_RPB, bool bwd = true) {
    return sizeof(float) * ((size_t)2 * side * (DH + 4) + (size_t)4 * side + (bwd ? (size_t)2 * rpb * DH + rpb : 0));
}
inline size_t attn_kv_lds(int side, int DH, int rpb = MAX_RPB) {
    return sizeof(float) * ((size_t)2 * side * (DH + 4) + (size_t)2 * side + (size_t)8 * side + (size_t)2 * rpb * DH);
}
inline bool attn_dims_ok(const caphn_attn_dims* d) {
    return d && d->bs > 0 && d->nh > 0 && d->dh > 0 && d->tq > 0 && d->tk > 0 && attn_dh_class(d->dh) != 0 &&
           d->dropout_p >= 0.f && d->dropout_p < 1.f;
}

template <typename K>
static int launch_attn(K kernel, dim3 grid, size_t lds, const AttnArgs& a, hipStream_t s) {
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return CAPHN_ELAUNCH;
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, a);
    return caphn_launch_status();
}

}  // namespace

extern "C" size_t caphn_layernorm_bwd_workspace_bytes(int rows, int d) {
    if (rows <= 0 || d <= 0) return 0;
    const int nb = std::min(512, (rows + 3) / 4);
    return caphn_align_up(sizeof(float) * (size_t)nb * 2 * d, 256) + caphn_colsum_workspace_bytes(nb, d);
}
extern "C" int caphn_layernorm_fwd(int rows, int d, const float* x, const float* gamma, const float* beta, float eps, float* y,
                                   float* mean, float* rstd, caphn_stream_t stream) {
    if (rows <= 0 || d <= 0 || d > 1024 || !x || !gamma || !beta || !y || !mean || !rstd) return CAPHN_EINVAL;
    const int nb = std::min(2048, (rows + 3) / 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d <= 256) hipLaunchKernelGGL(layernorm_fwd_kernel<4>, dim3(nb), dim3(256), 0, s, rows, d, x, gamma, beta, eps, y, mean, rstd);
    else hipLaunchKernelGGL(layernorm_fwd_kernel<16>, dim3(nb), dim3(256), 0, s, rows, d, x, gamma, beta, eps, y, mean, rstd);
    return caphn_launch_status();
}
extern "C" int caphn_layernorm_bwd(int rows, int d, const float* x, const float* gamma, const float* mean, const float* rstd,
                                   const float* dy, float* dx, float* dgamma, float* dbeta, void* ws, caphn_stream_t stream) {
    if (rows <= 0 || d <= 0 || d > 1024 || !x || !gamma || !mean || !rstd || !dy || !dx || !dgamma || !dbeta || !ws) return CAPHN_EINVAL;
    const int nb = std::min(512, (rows + 3) / 4);
    float* partial = static_cast<float*>(ws);
    void* cws = static_cast<char*>(ws) + caphn_align_up(sizeof(float) * (size_t)nb * 2 * d, 256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d <= 256) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(nb), dim3(256), 0, s, rows, d, x, gamma, mean, rstd, dy, dx, partial);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<16>, dim3(nb), dim3(256), 0, s, rows, d, x, gamma, mean, rstd, dy, dx, partial);
    int rc = caphn_colsum_f32(nb, d, partial, 2 * d, dgamma, cws, stream);
    if (rc != CAPHN_OK) return rc;
    return caphn_colsum_f32(nb, d, partial + d, 2 * d, dbeta, cws, stream);
}

extern "C" int caphn_attention_supported(const caphn_attn_dims* d) {
    if (!attn_dims_ok(d)) return 0;
    const int DH = attn_dh_class(d->dh);
    return attn_q_lds(d->tk, DH) <= ATTN_LDS_MAX && attn_kv_lds(d->tq, DH) <= ATTN_LDS_MAX;
}
extern "C" int caphn_attention_fwd(const caphn_attn_dims* d, const float* q, const float* k, const float* v, const float* attn_mask,
                                   const unsigned char* key_padding, float* o, float* lse, caphn_stream_t stream) {
    if (!attn_dims_ok(d) || !q || !k || !v || !o || !lse) return CAPHN_EINVAL;
    const int DH = attn_dh_class(d->dh);
    if (attn_q_lds(d->tk, DH) > ATTN_LDS_MAX) return CAPHN_EINVAL;
    AttnArgs a{}; a.d = *d; a.q = q; a.k = k; a.v = v; a.attn_mask = attn_mask; a.key_padding = key_padding; a.out0 = o; a.out1 = lse;
    a.rpb = attn_rows_per_block(d->bs * d->nh, d->tq, MAX_RPB);
    const size_t lds = attn_q_lds(d->tk, DH, a.rpb, false);
    const dim3 grid((d->tq + a.rpb - 1) / a.rpb, d->bs * d->nh);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return DH == 32 ? launch_attn(attn_q_kernel<32, 0>, grid, lds, a, s) : launch_attn(attn_q_kernel<64, 0>, grid, lds, a, s);
}
extern "C" int caphn_attention_bwd(const caphn_attn_dims* d, const float* q, const float* k, const float* v, const float* attn_mask,
                                   const unsigned char* key_padding, const float* o, const float* lse, const float* d_o,
                                   float* dq, float* dk, float* dv, caphn_stream_t stream) {
    if (!attn_dims_ok(d) || !q || !k || !v || !o || !lse || !d_o || !dq || !dk || !dv) return CAPHN_EINVAL;
    const int DH = attn_dh_class(d->dh);
    const size_t lq = attn_q_lds(d->tk, DH), lkv = attn_kv_lds(d->tq, DH);
    if (lq > ATTN_LDS_MAX || lkv > ATTN_LDS_MAX) return CAPHN_EINVAL;
    AttnArgs a{}; a.d = *d; a.q = q; a.k = k; a.v = v; a.o = o; a.dO = d_o; a.lse = lse; a.attn_mask = attn_mask; a.key_padding = key_padding;
    hipStream_t s = static_cast<hipStream_t>(stream);
    a.out0 = dq; a.out1 = nullptr;
    a.rpb = attn_rows_per_block(d->bs * d->nh, d->tq);
    const dim3 gq((d->tq + a.rpb - 1) / a.rpb, d->bs * d->nh);
    const size_t lq1 = attn_q_lds(d->tk, DH, a.rpb, true);
    int rc = DH == 32 ? launch_attn(attn_q_kernel<32, 1>, gq, lq1, a, s) : launch_attn(attn_q_kernel<64, 1>, gq, lq1, a, s);
    if (rc != CAPHN_OK) return rc;
    a.out0 = dk; a.out1 = dv;
    a.rpb = attn_rows_per_block(d->bs * d->nh, d->tk);
    const dim3 gk((d->tk + a.rpb - 1) / a.rpb, d->bs * d->nh);
    const size_t lkv1 = attn_kv_lds(d->tq, DH, a.rpb);
    return DH == 32 ? launch_attn(attn_kv_kernel<32>, gk, lkv1, a, s) : launch_attn(attn_kv_kernel<64>, gk, lkv1, a, s);
}
