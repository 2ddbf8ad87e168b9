// fp32 GEMM on the gfx950 matrix pipe: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulate -- bit-for-bit a k-ordered fmaf chain, so fp32 parity with the reference holds).
//
// One kernel template covers the three operand layouts of an nn.Linear forward/backward
// (include/caphn.h: caphn_gemm_f32).  Tiling is for 64-lane waves: 256 threads = 4 waves in a
// 2x2 grid, each wave owns a (BM/2)x(BN/2) block of 32x32 MFMA tiles.  K is consumed in slabs of
// 32 staged through LDS; an operand whose K index is contiguous in memory keeps K contiguous in
// LDS so one ds_read_b128 feeds four MFMA steps (the K order inside a slab is permuted
// identically for A and B, which a sum does not care about).  Global loads of slab t+1 are
// issued before the MFMAs of slab t and written to LDS after them.
#include "common.h"
#include "gemm_internal.h"

int g_tune_gemm = 1;      // 1: split-bf16 MFMA back end (default: equal-or-better accuracy, 5-18 % faster); 0: fp32 MFMA

namespace {

// BK = K-slab depth: 32, or 40 when K is a multiple of 40 but not of 32 (K = 200: 5 exact slabs
// instead of 6.25).  K-contiguous LDS rows are padded to BK+4 floats (144 B / 176 B pitch: odd
// multiples of 16 B, so the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots).
template <int BMN, bool KC, int BK>
struct Tile {
    static constexpr int KPAD = BK + 4;
    static constexpr int K4 = BK / 4;
    static constexpr int TOTAL = BMN * K4;              // float4 per slab
    static constexpr int NV = (TOTAL + 255) / 256;      // float4 per thread per slab
    static constexpr int LDM = BMN + 4;
    static constexpr int FLOATS = KC ? BMN * KPAD : BK * LDM;

    // global -> registers.  rows = extent of the M/N index, base row r0, slab start k0.
    __device__ static __forceinline__ void load(f32x4 (&r)[NV], const float* __restrict__ G, int ld,
                                                int rows, int K, int r0, int k0, int vec, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int idx = tid + 256 * i;
            if (TOTAL % 256 != 0 && idx >= TOTAL) { r[i] = v; continue; }
            if (KC) {               // memory [rows, K]
                int row = r0 + idx / K4;
                int k = k0 + (idx % K4) * 4;
                if (row < rows) {
                    const float* p = G + (size_t)row * ld + k;
                    if (vec && k + 3 < K) v = *reinterpret_cast<const f32x4*>(p);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (k + e < K) v[e] = p[e];
                    }
                }
            } else {                // memory [K, rows]
                int m = r0 + (idx % (BMN / 4)) * 4;
                int k = k0 + idx / (BMN / 4);
                if (k < K) {
                    const float* p = G + (size_t)k * ld + m;
                    if (vec && m + 3 < rows) v = *reinterpret_cast<const f32x4*>(p);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (m + e < rows) v[e] = p[e];
                    }
                }
            }
            r[i] = v;
        }
    }
    __device__ static __forceinline__ void store(const f32x4 (&r)[NV], float* __restrict__ S, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            if (TOTAL % 256 != 0 && idx >= TOTAL) continue;
            if (KC) *reinterpret_cast<f32x4*>(S + (idx / K4) * KPAD + (idx % K4) * 4) = r[i];
            else *reinterpret_cast<f32x4*>(S + (idx / (BMN / 4)) * LDM + (idx % (BMN / 4)) * 4) = r[i];
        }
    }
    // fragment of 4 k-steps for the 32 rows starting at `row` (lane i = lane&31, kh = lane>>5)
    __device__ static __forceinline__ f32x4 frag(const float* __restrict__ S, int row, int kq, int kh) {
        if (KC) return *reinterpret_cast<const f32x4*>(S + row * KPAD + kq * 8 + kh * 4);
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = S[(kq * 8 + kh * 4 + s) * LDM + row];
        return v;
    }
};

template <int BM, int BN, bool TA, bool TB, int BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    using TileA = Tile<BM, !TA, BK>;     // A [M,K] is K-contiguous unless transposed
    using TileB = Tile<BN, TB, BK>;      // B stored [N,K] (tb) is K-contiguous
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float lds[TileA::FLOATS + TileB::FLOATS];
    float* As = lds;
    float* Bs = lds + TileA::FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    const int nslab_total = (g.K + BK - 1) / BK;
    int slab0 = 0, slab1 = nslab_total;
    if (g.splitk > 1) {
        slab0 = blockIdx.z * g.slabs_per_split;
        slab1 = min(nslab_total, slab0 + g.slabs_per_split);
        if (slab0 >= slab1) return;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 ra[TileA::NV], rb[TileB::NV];
    TileA::load(ra, g.A, g.lda, g.M, g.K, m0, slab0 * BK, g.vecA, tid);
    TileB::load(rb, g.B, g.ldb, g.N, g.K, n0, slab0 * BK, g.vecB, tid);
    TileA::store(ra, As, tid);
    TileB::store(rb, Bs, tid);
    __syncthreads();

    for (int slab = slab0; slab < slab1; ++slab) {
        const bool more = slab + 1 < slab1;
        if (more) {
            TileA::load(ra, g.A, g.lda, g.M, g.K, m0, (slab + 1) * BK, g.vecA, tid);
            TileB::load(rb, g.B, g.ldb, g.N, g.K, n0, (slab + 1) * BK, g.vecB, tid);
        }
#pragma unroll
        for (int kq = 0; kq < BK / 8; ++kq) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = TileA::frag(As, wm * WM + a * 32 + li, kq, kh);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = TileB::frag(Bs, wn * WN + b * 32 + li, kq, kh);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            TileA::store(ra, As, tid);
            TileB::store(rb, Bs, tid);
            __syncthreads();
        }
    }

    // epilogue: acc register r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
    const bool atomic = g.splitk > 1;
    const bool add_bias = (g.flags & CAPHN_GEMM_BIAS) && (!atomic || blockIdx.z == 0);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + wn * WN + b * 32 + li;
            if (col >= g.N) continue;
            const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row >= g.M) continue;
                float v = acc[a][b][r] + bv;
                float* c = g.C + (size_t)row * g.ldc + col;
                if (atomic) { atomicAdd(c, v); continue; }
                if (g.flags & CAPHN_GEMM_ACCUM) v += *c;
                if (g.flags & CAPHN_GEMM_RELU) v = fmaxf(v, 0.f);
                if (g.flags & CAPHN_GEMM_LRELU) v = v > 0.f ? v : 0.01f * v;
                if (g.flags & CAPHN_GEMM_MASK) v = (g.mask[(size_t)row * g.ldmask + col] > 0.f) ? v : 0.f;
                *c = v;
            }
        }
}

template <int BM, int BN, int BK>
int launch_cfg(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.splitk > 1 ? g.splitk : 1);
    dim3 block(256);
    if (!ta && tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true, BK>), grid, block, 0, s, g);
    else if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false, BK>), grid, block, 0, s, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false, BK>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true, BK>), grid, block, 0, s, g);
    return caphn_launch_status();
}

}  // namespace

int caphn_gemm_mapped(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                      float* C, int ldc, const float* bias, int flags, int splitk,
                      const int* row_map, const int* dev_count, int map_mode, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || !row_map || !dev_count) return CAPHN_EINVAL;
    if ((flags & CAPHN_GEMM_BIAS) && !bias) return CAPHN_EINVAL;
    if (splitk > 1 && (flags & 0xFFFF & ~CAPHN_GEMM_BIAS)) return CAPHN_EINVAL;
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.mask = nullptr; g.ldmask = 0; g.flags = flags;
    const int nslab = (K + 31) / 32;
    if (splitk > nslab) splitk = nslab;
    g.splitk = splitk > 1 ? splitk : 1;
    g.slabs_per_split = (nslab + g.splitk - 1) / g.splitk;
    g.vecA = caphn_aligned16(A) && (lda % 4 == 0);
    g.vecB = caphn_aligned16(B) && (ldb % 4 == 0);
    g.row_map = row_map; g.dev_count = dev_count; g.map_mode = map_mode; g.colsum_a = nullptr; g.kmap_lds = 0;
    g.Ap = nullptr; g.Bp = nullptr; g.ldap = g.ldbp = 0; g.psa = g.psb = 0;
    return caphn_gemm_bf16x3_launch(g, ta, tb, s);      // the row subset lives in the split-bf16 back end
}

int caphn_gemm_tn_colsum(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         float* colsum_out, int splitk, const int* rowmap, void* cws, bool prezeroed, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || !colsum_out) return CAPHN_EINVAL;
    const int nslab = (K + 31) / 32;
    if (splitk > nslab) splitk = nslab;
    if (splitk < 1) splitk = 1;
    if ((splitk > 1 || rowmap) && !prezeroed) {
        if (ldc == N) { int rc = caphn_zero_f32(C, (size_t)M * N, s); if (rc) return rc; }
        else if (hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N, M, s) != hipSuccess) return CAPHN_ELAUNCH;
    }
    if (g_tune_gemm != 1 && !rowmap) {      // fp32 back end: no fusion (a row subset always runs in the split-bf16 back end,
                                            // and only the fused sum skips the unwritten rows of ignored targets)
        int rc = caphn_gemm_f32(1, 0, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, 0, splitk, s);
        if (rc) return rc;
        return caphn_colsum_f32(K, M, A, lda, colsum_out, cws, s);
    }
    if (!prezeroed) { int rc = caphn_zero_f32(colsum_out, (size_t)M, s); if (rc) return rc; }
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = nullptr; g.mask = nullptr; g.ldmask = 0; g.flags = 0;
    g.splitk = splitk; g.slabs_per_split = (nslab + splitk - 1) / splitk;
    g.vecA = caphn_aligned16(A) && (lda % 4 == 0);
    g.vecB = caphn_aligned16(B) && (ldb % 4 == 0);
    g.row_map = rowmap ? rowmap + 4 : nullptr; g.dev_count = rowmap; g.map_mode = rowmap ? 2 : 0;
    g.colsum_a = colsum_out; g.kmap_lds = 0;
    g.Ap = nullptr; g.Bp = nullptr; g.ldap = g.ldbp = 0; g.psa = g.psb = 0;
    return caphn_gemm_bf16x3_launch(g, 1, 0, s);
}

// Operands that may carry pre-split planes.  Falls back to the plain entry points whenever the split-bf16 back end is off
// or an operand has no planes.  The caller has zero-filled C when splitk > 1 or a row map is used.
int caphn_gemm_x(int ta, int tb, int M, int N, int K, const Opnd& A, const Opnd& B, float* C, int ldc, const GemmX& x, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || !A.f || !B.f || !C) return CAPHN_EINVAL;
    if ((x.flags & CAPHN_GEMM_BIAS) && !x.bias) return CAPHN_EINVAL;
    if ((x.flags & CAPHN_GEMM_MASK) && !x.mask) return CAPHN_EINVAL;
    if (g_tune_gemm != 1) {      // fp32 MFMA back end: no planes, no row maps, no fused column sums
        if (x.rowmap || x.colsum) return CAPHN_EINVAL;
        return caphn_gemm_f32(ta, tb, M, N, K, A.f, A.ld, B.f, B.ld, C, ldc, x.bias, x.mask, x.ldmask, x.flags, x.splitk, s);
    }
    int splitk = x.splitk;
    if (splitk > 1 && (x.flags & 0xFFFF & ~CAPHN_GEMM_BIAS)) return CAPHN_EINVAL;
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.A = A.f; g.lda = A.ld; g.B = B.f; g.ldb = B.ld; g.C = C; g.ldc = ldc;
    g.bias = x.bias; g.mask = x.mask; g.ldmask = x.ldmask; g.flags = x.flags;
    const int nslab = (K + 31) / 32;
    if (splitk > nslab) splitk = nslab;
    g.splitk = splitk > 1 ? splitk : 1;
    g.slabs_per_split = (nslab + g.splitk - 1) / g.splitk;
    g.vecA = caphn_aligned16(A.f) && (A.ld % 4 == 0);
    g.vecB = caphn_aligned16(B.f) && (B.ld % 4 == 0);
    g.row_map = x.rowmap ? x.rowmap + 4 : nullptr; g.dev_count = x.rowmap; g.map_mode = x.rowmap ? x.map_mode : 0;
    g.colsum_a = ta ? x.colsum : nullptr; g.kmap_lds = 0;
    const bool both = A.p && B.p;
    g.Ap = both ? A.p : nullptr; g.ldap = A.ldp; g.psa = A.ps;
    g.Bp = both ? B.p : nullptr; g.ldbp = B.ldp; g.psb = B.ps;
    if (both && x.Kp > K) {          // K not a multiple of 8: the planes are zero-padded to Kp
        g.K = x.Kp;
        if (!caphn_gemm_planes_ok(g, ta, tb)) g.K = K;
        else {
            const int ns = (g.K + 31) / 32;
            g.slabs_per_split = (ns + g.splitk - 1) / g.splitk;
        }
    }
    if (g.K == K && (K % 8) != 0) { g.Ap = nullptr; g.Bp = nullptr; }
    return caphn_gemm_bf16x3_launch(g, ta, tb, s);
}

extern "C" int caphn_split3_bf16(const float* src, int rows, int cols, int ld, void* planes, int ldp, size_t plane_stride,
                                 int zero_rows, caphn_stream_t stream) {
    if (!src || !planes || rows <= 0 || cols <= 0 || ld < cols || zero_rows < 0) return CAPHN_EINVAL;
    SplitJob j{src, ld, planes, ldp, plane_stride, rows, cols, zero_rows};
    return caphn_split3_launch(&j, 1, static_cast<hipStream_t>(stream));
}
extern "C" int caphn_gemm_planes_f32(int ta, int tb, int M, int N, int K,
                                     const float* A, int lda, const void* Ap, int ldap, size_t psa,
                                     const float* B, int ldb, const void* Bp, int ldbp, size_t psb,
                                     float* C, int ldc, const float* bias, const float* mask, int ldmask, int flags, int splitk,
                                     int Kp, caphn_stream_t stream) {
    GemmX x;
    x.bias = bias; x.mask = mask; x.ldmask = ldmask; x.flags = flags; x.splitk = splitk; x.Kp = Kp;
    return caphn_gemm_x(ta, tb, M, N, K, Opnd(A, lda, Ap, ldap, psa), Opnd(B, ldb, Bp, ldbp, psb), C, ldc, x,
                        static_cast<hipStream_t>(stream));
}

extern "C" int caphn_gemm_f32(int ta, int tb, int M, int N, int K,
                              const float* A, int lda, const float* B, int ldb,
                              float* C, int ldc, const float* bias,
                              const float* mask, int ldmask, int flags, int splitk,
                              caphn_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return CAPHN_EINVAL;
    if ((flags & CAPHN_GEMM_BIAS) && !bias) return CAPHN_EINVAL;
    if ((flags & CAPHN_GEMM_MASK) && !mask) return CAPHN_EINVAL;
    if (splitk > 1 && (flags & 0xFFFF & ~CAPHN_GEMM_BIAS)) return CAPHN_EINVAL;
    GemmArgs g;
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.mask = mask; g.ldmask = ldmask; g.flags = flags;
    g.row_map = nullptr; g.dev_count = nullptr; g.map_mode = 0; g.colsum_a = nullptr; g.kmap_lds = 0;
    g.Ap = nullptr; g.Bp = nullptr; g.ldap = g.ldbp = 0; g.psa = g.psb = 0;
    if (g_tune_gemm == 1) {          // split-bf16 back end: three bf16 planes per operand, 6 MFMAs per K=16
        const int nslab = (K + 31) / 32;
        if (splitk > nslab) splitk = nslab;
        g.splitk = splitk > 1 ? splitk : 1;
        g.slabs_per_split = (nslab + g.splitk - 1) / g.splitk;
        g.vecA = caphn_aligned16(A) && (lda % 4 == 0);
        g.vecB = caphn_aligned16(B) && (ldb % 4 == 0);
        return caphn_gemm_bf16x3_launch(g, ta, tb, static_cast<hipStream_t>(stream));
    }
    const int BK = (K % 40 == 0 && K % 32 != 0) ? 40 : 32;
    const int nslab = (K + BK - 1) / BK;
    if (splitk > nslab) splitk = nslab;
    g.splitk = splitk > 1 ? splitk : 1;
    g.slabs_per_split = (nslab + g.splitk - 1) / g.splitk;
    g.vecA = caphn_aligned16(A) && (lda % 4 == 0);
    g.vecB = caphn_aligned16(B) && (ldb % 4 == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // fp32 MFMA is slow enough (64 cycles per 32x32x2) that LDS/L2 reuse is not the limit: what matters is
    // balance over the 256 CUs.  128x128 tiles only when they alone give >= 4 workgroups per CU.
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128) * g.splitk;
    if (tiles128 >= 1024) return BK == 40 ? launch_cfg<128, 128, 40>(g, ta, tb, s) : launch_cfg<128, 128, 32>(g, ta, tb, s);
    return BK == 40 ? launch_cfg<64, 64, 40>(g, ta, tb, s) : launch_cfg<64, 64, 32>(g, ta, tb, s);
}
