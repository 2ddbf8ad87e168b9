// fp32-accurate GEMM on the bf16 matrix pipe ("split-bf16", 3 planes, 6 products).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  Every fp32 operand is split exactly into
// three bf16 planes  x = hi + mid + lo  (8 + 8 + 8 significand bits; bf16 keeps fp32's exponent range,
// so no scaling is needed), and the product is rebuilt from the six cross terms whose weight is
// >= 2^-16 relative:  hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid   (dropped: mid.lo, lo.mid,
// lo.lo <= 2^-24 relative -- the size of an fp32 rounding).  bf16 x bf16 products are exact in fp32 and the
// MFMA accumulates in fp32, so the result has fp32-class accuracy: per K = 16 it costs 6
// v_mfma_f32_32x32x16_bf16 (192 cycles) instead of 8 v_mfma_f32_32x32x2_f32 (512 cycles).
//
// Same operand layouts / epilogue / split-K as gemm_f32.hip.  The split happens while staging the fp32
// global tile into LDS (v_cvt_pk_bf16_f32, round to nearest even); LDS holds three K-contiguous planes per
// operand (80-byte row pitch: conflict-free ds_read_b128 fragments).
#include "common.h"
#include "gemm_internal.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;
constexpr int PITCH = 40;          // bf16 elements per LDS row (32 + 8 pad): 80 B = 5 x 16 B

struct Split4 { bf16x4 hi, mid, lo; };
__device__ __forceinline__ Split4 split3(f32x4 v) {
    Split4 s;
    s.hi = __builtin_convertvector(v, bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(s.hi, f32x4);
    s.mid = __builtin_convertvector(r1, bf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(s.mid, f32x4);
    s.lo = __builtin_convertvector(r2, bf16x4);
    return s;
}

// LDS image of one operand tile: three bf16 planes.
//  KC (K contiguous in memory): [BMN rows][PITCH] -- a fragment is one ds_read_b128 of a row.
//  !KC (K is the slow memory index, e.g. dY of a weight gradient): kept as stored, [32 k][PITCHM] with the
//      M/N index contiguous, so staging writes are plain 8-byte stores; the MFMA fragment (8 consecutive k of
//      one row) comes from two ds_read_b64_tr_b16 transposing reads.  PITCHM = BMN + 32 elements
//      (row pitch = 64 B mod 256 B: the 4 k-rows a 32-lane half touches land on disjoint bank windows).
template <int BMN, bool KC>
struct TileS {
    static constexpr int NV = BMN / 32;                 // float4 per thread per slab
    static constexpr int PITCHM = BMN + 32;
    static constexpr int PLANE = KC ? BMN * PITCH : BK * PITCHM;   // bf16 elements per plane
    static constexpr int ELEMS = 3 * PLANE;

    // pr: physical rows (KC) ; kmap: physical row of a logical k (!KC), may be null
    // physical rows of this thread's NV logical rows (KC operands); looked up once, not per K-slab
    __device__ static __forceinline__ void phys_rows(int (&pr)[NV], int rows, int r0, int tid, const int* __restrict__ rmap) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = r0 + ((tid + 256 * i) >> 3);
            pr[i] = (rmap && row < rows) ? rmap[row] : row;
        }
    }
    __device__ static __forceinline__ void load(f32x4 (&r)[NV], const float* __restrict__ G, int ld,
                                                int rows, int K, int r0, int k0, int vec, int tid,
                                                const int (&pr)[NV], const int* __restrict__ kmap) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int idx = tid + 256 * i;
            if (KC) {               // memory [rows, K]
                const int row = r0 + (idx >> 3);
                const int k = k0 + (idx & 7) * 4;
                if (row < rows) {
                    const float* p = G + (size_t)pr[i] * ld + k;
                    if (vec && k + 3 < K) v = *reinterpret_cast<const f32x4*>(p);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (k + e < K) v[e] = p[e];
                    }
                }
            } else {                // memory [K, rows]
                const int m = r0 + (idx % (BMN / 4)) * 4;
                const int k = k0 + idx / (BMN / 4);
                if (k < K) {
                    const float* p = G + (size_t)(kmap ? kmap[k] : k) * ld + m;
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
    __device__ static __forceinline__ void store(const f32x4 (&r)[NV], __bf16* __restrict__ S, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            const Split4 s = split3(r[i]);
            __bf16* p = KC ? S + (idx >> 3) * PITCH + (idx & 7) * 4
                           : S + (idx / (BMN / 4)) * PITCHM + (idx % (BMN / 4)) * 4;
            *reinterpret_cast<bf16x4*>(p) = s.hi;
            *reinterpret_cast<bf16x4*>(p + PLANE) = s.mid;
            *reinterpret_cast<bf16x4*>(p + 2 * PLANE) = s.lo;
        }
    }
    // MFMA operand of lane l (r = l&31, h = l>>5) for k-step ks: elements k = 16 ks + 8 h + (0..7) of row `row0 + r`
    __device__ static __forceinline__ bf16x8 frag(const __bf16* __restrict__ S, int plane, int row0, int ks, int lane) {
        if (KC) {
            return *reinterpret_cast<const bf16x8*>(S + plane * PLANE + (row0 + (lane & 31)) * PITCH + ks * 16 + (lane >> 5) * 8);
        } else {
            // 16-lane group g reads a 4(k) x 16(m) block; lane 4q+p supplies the address of k-row q, columns 4p..4p+3
            // and receives column (lane&15) of the four rows.
            const int g = lane >> 4, i = lane & 15;
            const int col = row0 + 16 * (g & 1) + 4 * (i & 3);
            const int kb = 16 * ks + 8 * (g >> 1) + (i >> 2);
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            const __bf16* p = S + plane * PLANE + kb * PITCHM + col;
            const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
            const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * PITCHM));
            return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
};

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmArgs g) {
    using TileA = TileS<BM, !TA>;
    using TileB = TileS<BN, TB>;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) __bf16 lds[TileA::ELEMS + TileB::ELEMS];
    __bf16* As = lds;
    __bf16* Bs = lds + TileA::ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    // row subset: effective extents come from device memory (no host round trip)
    const int* rmapA = nullptr; const int* kmap = nullptr;
    if (g.map_mode == 1) { g.M = min(g.M, g.dev_count[0]); rmapA = g.row_map; if (m0 >= g.M) return; }
    if (g.map_mode == 2) { g.K = min(g.K, g.dev_count[0]); kmap = g.row_map; }

    const int nslab_total = (g.K + BK - 1) / BK;
    int slab0 = 0, slab1 = nslab_total;
    if (g.splitk > 1) {
        const int sps = g.map_mode == 2 ? (nslab_total + g.splitk - 1) / g.splitk : g.slabs_per_split;
        slab0 = blockIdx.z * sps;
        slab1 = min(nslab_total, slab0 + sps);
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
    int pra[TileA::NV], prb[TileB::NV];
    TileA::phys_rows(pra, g.M, m0, tid, rmapA);
    TileB::phys_rows(prb, g.N, n0, tid, nullptr);
    TileA::load(ra, g.A, g.lda, g.M, g.K, m0, slab0 * BK, g.vecA, tid, pra, kmap);
    TileB::load(rb, g.B, g.ldb, g.N, g.K, n0, slab0 * BK, g.vecB, tid, prb, kmap);
    TileA::store(ra, As, tid);
    TileB::store(rb, Bs, tid);
    __syncthreads();

    for (int slab = slab0; slab < slab1; ++slab) {
        const bool more = slab + 1 < slab1;
        if (more && !(g.flags & (1 << 19))) {
            TileA::load(ra, g.A, g.lda, g.M, g.K, m0, (slab + 1) * BK, g.vecA, tid, pra, kmap);
            TileB::load(rb, g.B, g.ldb, g.N, g.K, n0, (slab + 1) * BK, g.vecB, tid, prb, kmap);
        }
        const int nks = (g.K - slab * BK) > 16 ? 2 : 1;      // skip the all-zero second k-step of a short tail
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks < nks) {
                bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int p = 0; p < 3; ++p) fa[a][p] = TileA::frag(As, p, wm * WM + a * 32, ks, lane);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int p = 0; p < 3; ++p) fb[b][p] = TileB::frag(Bs, p, wn * WN + b * 32, ks, lane);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        f32x16 c = acc[a][b];            // smallest terms first
                        if (g.flags & (1 << 16)) {       // ablation: keep the fragments alive, skip the MFMAs
                            asm volatile("" :: "v"(fa[a][0]), "v"(fa[a][1]), "v"(fa[a][2]), "v"(fb[b][0]), "v"(fb[b][1]), "v"(fb[b][2]));
                            continue;
                        }
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], c, 0, 0, 0);   // lo  . hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], c, 0, 0, 0);   // hi  . lo
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], c, 0, 0, 0);   // mid . mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], c, 0, 0, 0);   // mid . hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], c, 0, 0, 0);   // hi  . mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], c, 0, 0, 0);   // hi  . hi
                        acc[a][b] = c;
                    }
            }
        }
        __syncthreads();
        if (more) {
            if (!(g.flags & (1 << 18))) {
                TileA::store(ra, As, tid);
                TileB::store(rb, Bs, tid);
            }
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
                float* c = g.C + (size_t)(rmapA ? rmapA[row] : row) * g.ldc + col;
                if ((g.flags & (1 << 17)) && v != 12345.678f) continue;      // ablation: no epilogue stores
                if (atomic) { atomicAdd(c, v); continue; }
                if (g.flags & CAPHN_GEMM_ACCUM) v += *c;
                if (g.flags & CAPHN_GEMM_RELU) v = fmaxf(v, 0.f);
                if (g.flags & CAPHN_GEMM_MASK) v = (g.mask[(size_t)row * g.ldmask + col] > 0.f) ? v : 0.f;
                *c = v;
            }
        }
}

template <int BM, int BN>
int launch_cfg(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.splitk > 1 ? g.splitk : 1);
    dim3 block(256);
    if (!ta && tb) hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, false, true>), grid, block, 0, s, g);
    else if (!ta && !tb) hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, false, false>), grid, block, 0, s, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, true, false>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, true, true>), grid, block, 0, s, g);
    return caphn_launch_status();
}

}  // namespace

int g_tune_gemm_tile = 0;     // 0: 128x128 when >= 512 such tiles else 64x64; 1: also try 128x64 when >= 512 such tiles
int caphn_gemm_bf16x3_launch(GemmArgs g, int ta, int tb, hipStream_t s) {
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.splitk;
    if (tiles128 >= 512) return launch_cfg<128, 128>(g, ta, tb, s);
    if (g_tune_gemm_tile == 1) {
        const long t12864 = (long)((g.M + 127) / 128) * ((g.N + 63) / 64) * g.splitk;
        if (t12864 >= 512) return launch_cfg<128, 64>(g, ta, tb, s);
    }
    return launch_cfg<64, 64>(g, ta, tb, s);
}
