// K-resident variant of the split-bf16 GEMM for the NT products with a SHORT contraction (K <= 208): C = A B^T (+ bias), A [M, K] and
// B [N, K] both K-contiguous -- the vocabulary logits (Hs [B T, 200] x fc.weight [9684, 200], models/decoderlstm.py:105) and the other
// K = 200 products of the forward (feature_fc.2, W_a f, G, the x-side gates: :22-26, attention.py:34, :100).
//
// The general kernel (gemm_bf16x3.hip) re-stages BOTH operands per 32-deep slab and per 64x64 tile: with K = 200 a tile is seven
// slabs, every slab two barriers, and per MFMA it issues 9 vector instructions of split arithmetic and reads 1 KB of fragments from
// LDS -- its matrix pipe is busy 21-27 % of the time.  Here a wave keeps the fragments of ITS 32 rows of A for the WHOLE K in
// registers (13 k-steps x 3 planes x 4 VGPRs = 156), split once per workgroup, and walks over 32-column tiles of B: only B is
// staged (split on the fly, three bf16 planes, whole K, 41 KB of LDS), one stage and two barriers per 78 MFMAs instead of per 12,
// 1.5 vector instructions and 0.5 KB of LDS reads per MFMA.  Arithmetic as in the general kernel: every fp32 operand as three bf16
// planes, the six cross products >= 2^-16, fp32 accumulation, smallest terms first.
#include "common.h"
#include "gemm_internal.h"
#include "split3.h"

int g_tune_gemm_kres = 0;       // caphn_tune key 36: 1 = on.  OFF: alone the vocabulary logits run 1.09x faster (62.5 vs 68.2 us at 1660 live rows),
                                // in the step -- where the grid covers all 2560 rows and the workgroups of the dead row panels leave at once --
                                // +12 us (tools/ab_inproc.py); the N = 200 .. 600 products 0.9x.  Kept as the measured "different kernel class":
                                // DESIGN.md section 6, tools/kres_phase_profile.py
#ifdef CAPHN_GEMM_PROFILE
__device__ unsigned long long d_kres_prof[8];      // workgroup (0,0), wave 0: shader-clock sums per phase
extern "C" int caphn_debug_kres_prof(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(d_kres_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(d_kres_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define KSTAMP(i) do { if (kprof) { unsigned long long _n = clock64(); kpc[i] += _n - klast; klast = _n; } } while (0)
#else
#define KSTAMP(i) do { } while (0)
#endif

namespace {

constexpr int KR_BM = 128;          // rows of A per workgroup (4 waves x 32)
constexpr int KR_BN = 32;           // columns of C per tile
constexpr int KR_KS = 13;           // k-steps of 16: K <= 208
constexpr int KR_PITCH = 216;       // bf16 elements per LDS row: 432 bytes = 27 x 16 -- the 16 lanes of a ds_read_b128 group (rows r, r + 1, ..)
                                    // start at 27 r mod 16 distinct 16-byte slots of the 256-byte bank row
constexpr int KR_PLANE = KR_BN * KR_PITCH;
constexpr int KR_NV = 7;            // 16-byte chunks of a B tile per thread: ceil(32 x 52 / 256)

// KT: the contraction length, a compile-time constant (the divisions of the tile maps fold away: fourteen registers a lane that the
// software-pipelined epilogue needs).  Instantiated for K = 200, the width every product of this family has in the reference's
// configurations (feature_out = embedding_dim = hidden_dim = 200); other lengths take the general kernel.
template <int KT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_kres_kernel(GemmArgs g, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];      // 3 planes x 32 rows x KR_PITCH
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kh = lane >> 5;
    int M = g.M;
    const int* rmap = nullptr;
    if (g.map_mode == 1) { M = min(M, g.dev_count[0]); rmap = g.row_map; }
    const int m0 = blockIdx.y * KR_BM;
    if (m0 >= M) return;
    const int ntiles = (g.N + KR_BN - 1) / KR_BN;
    const int t0 = blockIdx.x * tiles_per_chunk, t1 = min(ntiles, t0 + tiles_per_chunk);
    if (t0 >= t1) return;
    constexpr int K = KT, K4 = KT >> 2;
    static_assert(KT % 4 == 0 && KT >= 16 && KT <= 16 * KR_KS, "K-resident GEMM: 16 <= K <= 208, K % 4 == 0");
#ifdef CAPHN_GEMM_PROFILE
    const bool kprof = tid == 0 && blockIdx.x == 0 && blockIdx.y == 0;
    unsigned long long kpc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, klast = kprof ? clock64() : 0;
#endif

    // the LDS columns K .. KR_PITCH-1 are never staged: zero once (the A fragments hold zeros there too)
    for (int i = tid; i < 3 * KR_BN * (KR_PITCH - K); i += 256) {
        const int p = i / (KR_BN * (KR_PITCH - K)), rem = i - p * (KR_BN * (KR_PITCH - K));
        const int n = rem / (KR_PITCH - K), c = rem - n * (KR_PITCH - K);
        lds[p * KR_PLANE + n * KR_PITCH + K + c] = (__bf16)0.f;
    }

    // ---- my rows' fragments of A for the whole K: lane (li, kh) holds k = 16 ks + 8 kh + (0..7) of row m0 + 32 wave + li
    bf16x8 fa[KR_KS][3];
    {
        const int arow = min(m0 + wave * 32 + li, M - 1);       // (rows past the edge repeat the last one: never stored)
        const float* ap = g.A + (size_t)(rmap ? rmap[arow] : arow) * g.lda + kh * 8;
        constexpr int HALF = 7;
#pragma unroll
        for (int h0 = 0; h0 < KR_KS; h0 += HALF) {
            f32x4 v[HALF][2];
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                const int ks = h0 + j, k = ks * 16 + kh * 8;
                v[j][0] = (ks < KR_KS && k + 3 < K) ? *reinterpret_cast<const f32x4*>(ap + ks * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                v[j][1] = (ks < KR_KS && k + 7 < K) ? *reinterpret_cast<const f32x4*>(ap + ks * 16 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                const int ks = h0 + j;
                if (ks < KR_KS) {
                    const Split4 s0 = split3(v[j][0]), s1 = split3(v[j][1]);
                    fa[ks][0] = __builtin_shufflevector(s0.hi, s1.hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    fa[ks][1] = __builtin_shufflevector(s0.mid, s1.mid, 0, 1, 2, 3, 4, 5, 6, 7);
                    fa[ks][2] = __builtin_shufflevector(s0.lo, s1.lo, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
        }
    }
    // where my 16 accumulator registers go: register r of lane (li, kh) is C[row0 + (r & 3) + 8 (r >> 2) + 4 kh][n0 + li].  The 128
    // physical row numbers of the workgroup sit in LDS behind the planes (sixteen registers a lane otherwise: the kernel is at its
    // 256-VGPR budget with the A fragments)
    int* crow_s = reinterpret_cast<int*>(lds + 3 * KR_PLANE);
    if (tid < KR_BM) {
        const int row = m0 + tid;
        crow_s[tid] = row < M ? (rmap ? rmap[row] : row) : -1;
    }
    // my chunks of a B tile: chunk idx = tid + 256 i covers floats 4 c .. 4 c + 3 of tile row n (n = idx / K4, c = idx % K4)
    const bool relu = (g.flags & CAPHN_GEMM_RELU) != 0, lrelu = (g.flags & CAPHN_GEMM_LRELU) != 0, add_bias = (g.flags & CAPHN_GEMM_BIAS) != 0;

    auto store_one = [&](float a, int r, int col, float bv) {
        const int cr = crow_s[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
        if (cr >= 0) {
            float v = a + bv;
            if (relu) v = fmaxf(v, 0.f);
            if (lrelu) v = v > 0.f ? v : 0.01f * v;
            g.C[(unsigned)cr * (unsigned)g.ldc + (unsigned)col] = v;       // (the launcher checked that C spans < 2^31 elements)
        }
    };
    f32x4 rb[KR_NV];
    auto bload = [&](int t) {
        const int n0 = t * KR_BN;
#pragma unroll
        for (int i = 0; i < KR_NV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < KR_BN * K4) {
                const int nn = idx / K4, c = idx - nn * K4;
                const int n = min(n0 + nn, g.N - 1);
                rb[i] = *reinterpret_cast<const f32x4*>(g.B + ((unsigned)n * (unsigned)g.ldb + (unsigned)c * 4u));      // (< 4 GB: launcher)
            }
        }
    };
    bload(t0);
    KSTAMP(0);                                     // prologue: A fragments, maps, first B loads issued
    for (int t = t0; t < t1; ++t) {
        // stage tile t: split on the fly, three planes, whole K
#pragma unroll
        for (int i = 0; i < KR_NV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < KR_BN * K4) {
                const int lo = (idx / K4) * (KR_PITCH - KT) + idx * 4;       // n pitch + 4 c with idx = n K4 + c
                const Split4 s = split3(rb[i]);
                *reinterpret_cast<bf16x4*>(lds + lo) = s.hi;
                *reinterpret_cast<bf16x4*>(lds + KR_PLANE + lo) = s.mid;
                *reinterpret_cast<bf16x4*>(lds + 2 * KR_PLANE + lo) = s.lo;
            }
        }
        KSTAMP(1);                                 // wait for the tile's loads, split, LDS stores
        __syncthreads();
        KSTAMP(2);
        if (t + 1 < t1) bload(t + 1);              // in flight during this tile's MFMAs
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const __bf16* brow = lds + li * KR_PITCH + kh * 8;
#pragma unroll
        for (int ks = 0; ks < KR_KS; ++ks) {
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(brow + ks * 16);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(brow + KR_PLANE + ks * 16);
            const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(brow + 2 * KR_PLANE + ks * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][2], b0, acc, 0, 0, 0);   // lo  . hi
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0], b2, acc, 0, 0, 0);   // hi  . lo
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][1], b1, acc, 0, 0, 0);   // mid . mid
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][1], b0, acc, 0, 0, 0);   // mid . hi
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0], b1, acc, 0, 0, 0);   // hi  . mid
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][0], b0, acc, 0, 0, 0);   // hi  . hi
        }
        __builtin_amdgcn_sched_barrier(0);
        KSTAMP(3);                                 // next loads issued, fragment reads, MFMAs
        __syncthreads();                           // the LDS image is free for tile t + 1
        KSTAMP(4);
        // (tried and equal: the stores of tile t interleaved with the MFMAs of tile t + 1 through a copy of the accumulators in LDS --
        //  a store that cannot issue blocks the MFMAs behind it; non-temporal stores; odd chunks started half a tile late)
        const int col = t * KR_BN + li;
        if (col < g.N) {
            const float bv = add_bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) store_one(acc[r], r, col, bv);
        }
        KSTAMP(5);                                 // epilogue stores issued
    }
#ifdef CAPHN_GEMM_PROFILE
    if (kprof) { for (int i = 0; i < 6; ++i) atomicAdd(&d_kres_prof[i], kpc[i]); atomicAdd(&d_kres_prof[6], (unsigned long long)(t1 - t0)); atomicAdd(&d_kres_prof[7], 1ull); }
#endif
}

}  // namespace

extern "C" int caphn_debug_kres_occupancy(void) {
    int n = 0;
    const size_t lds = sizeof(__bf16) * 3 * KR_PLANE + sizeof(int) * KR_BM;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(gemm_kres_kernel<200>), 256, lds) != hipSuccess) return -1;
    return n;
}
// CAPHN_OK: launched.  1: not applicable (the caller takes the general kernel).
int caphn_gemm_kres_launch(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    if (!g_tune_gemm_kres || ta || !tb) return 1;
    if (g.K != 200 || (g.lda & 3) || (g.ldb & 3) || !g.vecA || !g.vecB) return 1;
    if (g.splitk > 1 || g.map_mode == 2 || g.colsum_a || g.Ap || g.Bp) return 1;
    if (g.flags & ~(CAPHN_GEMM_BIAS | CAPHN_GEMM_RELU | CAPHN_GEMM_LRELU | (1 << 20) | (1 << 21) | (1 << 22))) return 1;
    if (g.M < 64 || g.N < 1024) return 1;       // (N = 200 .. 600: two tiles a workgroup do not amortise the A split: 0.9x, measured)
    if ((double)g.N * g.ldb * 4.0 >= 4.0e9 || (double)g.M * g.ldc >= 2.0e9) return 1;
    const int panels = (g.M + KR_BM - 1) / KR_BM, ntiles = (g.N + KR_BN - 1) / KR_BN;
    // two workgroups per CU (250 VGPRs): aim at ~512 workgroups, at least two tiles each so that the A split is amortised
    int chunks = (512 + panels - 1) / panels;
    if (chunks > (ntiles + 1) / 2) chunks = (ntiles + 1) / 2;
    if (chunks < 1) chunks = 1;
    const int tpc = (ntiles + chunks - 1) / chunks;
    chunks = (ntiles + tpc - 1) / tpc;
    const size_t lds = sizeof(__bf16) * 3 * KR_PLANE + sizeof(int) * KR_BM;
    hipLaunchKernelGGL(gemm_kres_kernel<200>, dim3((unsigned)chunks, (unsigned)panels), dim3(256), lds, s, g, tpc);
    return caphn_launch_status();
}
