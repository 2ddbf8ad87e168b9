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
#include "split3.h"
#include <type_traits>

// per-phase shader-clock stamps of workgroup (0,0,0), wave 0 (tools/gemm_phase_profile.py): compile with -DCAPHN_GEMM_PROFILE
#ifdef CAPHN_GEMM_PROFILE
__device__ unsigned long long d_gemm_prof[10];
__device__ unsigned long long d_gemm_wgt[2 * 8192];     // per workgroup: wall-clock (100 MHz) at entry and after the epilogue
extern "C" int caphn_debug_gemm_wgtimes(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(d_gemm_wgt), sizeof(unsigned long long) * 2 * (n < 8192 ? n : 8192)) == hipSuccess ? 0 : -1;
}
#define GSTAMP(i) do { if (gprof) { unsigned long long _n = clock64(); gpc[i] += _n - glast; glast = _n; } } while (0)
extern "C" int caphn_debug_gemm_prof(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(d_gemm_prof), sizeof(unsigned long long) * 10) != hipSuccess) return -1;
    if (reset) { unsigned long long z[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(d_gemm_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define GSTAMP(i) do { } while (0)
#endif

extern int g_tune_gemm_waves;
namespace {

constexpr int BK = 32;
constexpr int PITCH = 32;          // bf16 elements per LDS row of a K-contiguous operand: no pad -- the four 16-byte chunks of
                                   // row r are stored at chunk ^ ((r >> 2) & 3), which makes the 16-lane groups of a
                                   // ds_read_b128 fragment read (rows r, r+1, .. of one chunk) hit 16 distinct 16-byte slots of
                                   // the 256-byte bank row, and the staging stores of two rows fill one 128-byte half.
                                   // (The 80-byte padded pitch this replaces showed 33 % bank-conflict cycles in SQ_LDS_BANK_CONFLICT.)
__device__ __forceinline__ int kc_off(int row, int k) { return row * PITCH + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7); }

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
    __device__ static __forceinline__ void phys_rows(int (&pr)[NV], int rows, int r0, int tid, const int* __restrict__ rmap,
                                                     bool clamp) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int row = r0 + ((tid + 256 * i) >> 3);
            if (clamp) row = min(row, rows - 1);
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
    // Branch-free variant for 16-byte aligned operands (ld % 4 == 0, K % 4 == 0, extents >= 4): every thread always
    // issues exactly NV dwordx4 loads, from CLAMPED coordinates -- rows past the edge re-read the last valid row (their
    // products land in accumulator rows/columns the epilogue never stores); a K tail is zeroed later, in store_tail, NOT here:
    // a select at the load site makes the compiler wait for the data right after requesting it.  A
    // static load count lets the compiler wait with s_waitcnt vmcnt(n > 0) for the OLDER register set only; with
    // the bounds checks as branches it had to drain everything (vmcnt(0)), which cancelled the look-ahead.
    // kl (K-slow operands only): LDS copy of the row map for logical k in [kbase, ...): the physical row comes from a
    // ds_read, so the number of VMEM loads stays static
    // byte offsets of this thread's NV loads for the branch-free path, formed ONCE per workgroup: the per-slab address is then the
    // uniform base plus a 32-bit offset (the launcher takes this path only while an operand spans < 4 GB); a 64-bit multiply
    // per load and slab was 0.5 k of a 128x128 slab's 4.9 k cycles
    __device__ static __forceinline__ void fast_bases(unsigned (&bo)[NV], int ld, int rows, int r0, int tid, const int (&pr)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            if (KC) bo[i] = ((unsigned)pr[i] * (unsigned)ld + (unsigned)((idx & 7) * 4)) * 4u;          // + clamped k0
            else bo[i] = (unsigned)min(r0 + (idx % (BMN / 4)) * 4, rows - 4) * 4u;                       // + physical k row * ld
        }
    }
    __device__ static __forceinline__ void load_fast_b(f32x4 (&r)[NV], const float* __restrict__ G, const unsigned (&bo)[NV], int ld,
                                                       int K, int k0, int tid, const int* kl, int kbase) {
        const char* base = reinterpret_cast<const char*>(G);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            unsigned off;
            if (KC) off = bo[i] + (unsigned)min(k0, K - 4 - (idx & 7) * 4) * 4u;
            else {
                int kc = min(k0 + idx / (BMN / 4), K - 1);
                if (kl) kc = kl[kc - kbase];
                off = bo[i] + (unsigned)kc * (unsigned)ld * 4u;
            }
            issue(r[i], reinterpret_cast<const float*>(base + off));
        }
    }
    __device__ static __forceinline__ void load_fast(f32x4 (&r)[NV], const float* __restrict__ G, int ld,
                                                     int rows, int K, int r0, int k0, int tid, const int (&pr)[NV],
                                                     const int* kl, int kbase) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            if (KC) {               // memory [rows, K]; pr[] is already clamped
                const int k = k0 + (idx & 7) * 4;
                const int kc = min(k, K - 4);
                issue(r[i], G + (size_t)pr[i] * ld + kc);
            } else {                // memory [K, rows]
                const int m = min(r0 + (idx % (BMN / 4)) * 4, rows - 4);
                const int k = k0 + idx / (BMN / 4);
                int kc = min(k, K - 1);
                if (kl) kc = kl[kc - kbase];
                issue(r[i], G + (size_t)kc * ld + m);
            }
        }
    }
    // A plain load, tracked by the compiler's wait-count pass.  (Issuing it from inline assembly with hand-placed
    // s_waitcnt vmcnt(n) removes two conservative waits at the top of the loop and measured 3 % faster, but it is
    // unsound: the register allocator may COPY a register set between the load and the wait -- it did, at the loop's
    // back edge -- reading registers whose loads are still in flight.  The failure is intermittent.)
    __device__ static __forceinline__ void issue(f32x4& r, const float* p) { r = *reinterpret_cast<const f32x4*>(p); }
    // reduced-precision side mode: one plane, operands rounded to bf16 (v_cvt_pk_bf16_f32, round to nearest even)
    __device__ static __forceinline__ void store_single(const f32x4 (&r)[NV], __bf16* __restrict__ S, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            const bf16x4 v = {(__bf16)r[i][0], (__bf16)r[i][1], (__bf16)r[i][2], (__bf16)r[i][3]};
            __bf16* p = KC ? S + kc_off(idx >> 3, (idx & 7) * 4)
                           : S + (idx / (BMN / 4)) * PITCHM + (idx % (BMN / 4)) * 4;
            *reinterpret_cast<bf16x4*>(p) = v;
        }
    }
    // two-plane side mode (MODE 4): hi by truncation (exact), mid = the remainder ROUNDED to bf16 -- hi + mid carries 16 significand
    // bits, the best two-plane representation; the lo plane is not written
    __device__ static __forceinline__ void store2(const f32x4 (&r)[NV], __bf16* __restrict__ S, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            union { f32x4 f; unsigned u[4]; } x, h;
            x.f = r[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) h.u[e] = x.u[e] & 0xffff0000u;
            const f32x4 rem = x.f - h.f;
            union { unsigned u[2]; bf16x4 b; } hi;
            hi.u[0] = pack_hi16(x.u[0], x.u[1]); hi.u[1] = pack_hi16(x.u[2], x.u[3]);
            const bf16x4 mid = {(__bf16)rem[0], (__bf16)rem[1], (__bf16)rem[2], (__bf16)rem[3]};
            __bf16* p = KC ? S + kc_off(idx >> 3, (idx & 7) * 4)
                           : S + (idx / (BMN / 4)) * PITCHM + (idx % (BMN / 4)) * 4;
            *reinterpret_cast<bf16x4*>(p) = hi.b;
            *reinterpret_cast<bf16x4*>(p + PLANE) = mid;
        }
    }
    __device__ static __forceinline__ void store(const f32x4 (&r)[NV], __bf16* __restrict__ S, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            const Split4 s = split3(r[i]);
            __bf16* p = KC ? S + kc_off(idx >> 3, (idx & 7) * 4)
                           : S + (idx / (BMN / 4)) * PITCHM + (idx % (BMN / 4)) * 4;
            *reinterpret_cast<bf16x4*>(p) = s.hi;
            *reinterpret_cast<bf16x4*>(p + PLANE) = s.mid;
            *reinterpret_cast<bf16x4*>(p + 2 * PLANE) = s.lo;
        }
    }
    // K-tail slab of the branch-free path: zero the elements whose k >= K, then stage
    __device__ static __forceinline__ void mask_tail(f32x4 (&r)[NV], int tid, int k0, int K) {
        if (k0 + BK > K) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = tid + 256 * i;
                const int k = KC ? k0 + (idx & 7) * 4 : k0 + idx / (BMN / 4);
                if (k >= K) r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    // ---- pre-split operands (GemmArgs::Ap / Bp): a thread moves NVP 16-byte chunks (8 bf16) per plane per slab, global ->
    // registers -> LDS, no arithmetic.  KC: 4 chunks per row; !KC: BMN / 8 chunks per k-row.
    static constexpr int NVP = BMN / 64;
    __device__ static __forceinline__ void phys_rows_p(int (&pr)[NVP], int rows, int r0, int tid, const int* __restrict__ rmap) {
#pragma unroll
        for (int i = 0; i < NVP; ++i) {
            const int row = min(r0 + ((tid + 256 * i) >> 2), rows - 1);
            pr[i] = rmap ? rmap[row] : row;
        }
    }
    __device__ static __forceinline__ void load_p(bf16x8 (&r)[NVP][3], const __bf16* __restrict__ P, int ldp, size_t ps,
                                                  int rows, int K, int r0, int k0, int tid, const int (&pr)[NVP],
                                                  const int* kl, int kbase) {
#pragma unroll
        for (int i = 0; i < NVP; ++i) {
            const int idx = tid + 256 * i;
            size_t off;
            if (KC) {               // memory [rows, K]; pr[] is clamped (and mapped)
                const int kc = min(k0 + (idx & 3) * 8, K - 8);
                off = (size_t)pr[i] * ldp + kc;
            } else {                // memory [K, rows]; the last chunk of a row may reach into the pad of the leading dimension
                const int m = min(r0 + (idx % (BMN / 8)) * 8, ((rows + 7) & ~7) - 8);
                int kc = min(k0 + idx / (BMN / 8), K - 1);
                if (kl) kc = kl[kc - kbase];
                off = (size_t)kc * ldp + m;
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) r[i][pl] = *reinterpret_cast<const bf16x8*>(P + pl * ps + off);
        }
    }
    __device__ static __forceinline__ void store_p(const bf16x8 (&r)[NVP][3], __bf16* __restrict__ S, int tid, int k0, int K) {
#pragma unroll
        for (int i = 0; i < NVP; ++i) {
            const int idx = tid + 256 * i;
            const int k = KC ? k0 + (idx & 3) * 8 : k0 + idx / (BMN / 8);
            const bool dead = k >= K;                 // K tail of the last slab: zeros (K % 8 == 0, so whole chunks)
            __bf16* p = KC ? S + kc_off(idx >> 2, (idx & 3) * 8)
                           : S + (idx / (BMN / 8)) * PITCHM + (idx % (BMN / 8)) * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                bf16x8 v = r[i][pl];
                if (dead) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<bf16x8*>(p + pl * PLANE) = v;
            }
        }
    }
    // MFMA operand of lane l (r = l&31, h = l>>5) for k-step ks: elements k = 16 ks + 8 h + (0..7) of row `row0 + r`
    __device__ static __forceinline__ bf16x8 frag(const __bf16* __restrict__ S, int plane, int row0, int ks, int lane) {
        if (KC) {
            return *reinterpret_cast<const bf16x8*>(S + plane * PLANE + kc_off(row0 + (lane & 31), ks * 16 + (lane >> 5) * 8));
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

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// MODE 0: fp32 operands split into three planes on every use of a tile, six products (fp32-class accuracy; the default)
// MODE 1: pre-split operands (GemmArgs::Ap / Bp), six products
// MODE 2: ONE product on operands rounded to bf16 (round to nearest even) at staging -- the reduced-precision side mode
//         (caphn_tune key 11; never the default, never the headline measurement)
template <int BM, int BN, bool TA, bool TB, int MODE>
__device__ __forceinline__ void gemm_bf16x3_body(GemmArgs g) {
    constexpr bool PL = MODE == 1;
    constexpr bool SINGLE = MODE == 2;
    constexpr bool WS = MODE == 5;          // wave-specialised: 512 threads -- waves 4..7 load / split / stage, waves 0..3 only read
                                            // fragments and issue MFMAs; two LDS images, one barrier per slab (see the WS loop)
    constexpr bool DB = MODE == 3 || WS;    // MODE 0 with TWO LDS images of a slab (ping-pong): see mainloop_db
    constexpr bool TWO = MODE == 4;         // two planes (hi + mid, 16 significand bits), three products: the bf16x2 side mode
    using TileA = TileS<BM, !TA>;
    using TileB = TileS<BN, TB>;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int TILE_ELEMS = TileA::ELEMS + TileB::ELEMS;
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];      // TILE_ELEMS (up to 77 KB), twice that with DB
    __bf16* As = lds;
    __bf16* Bs = lds + TileA::ELEMS;
    __bf16* As1 = DB ? lds + TILE_ELEMS : As;
    __bf16* Bs1 = DB ? lds + TILE_ELEMS + TileA::ELEMS : Bs;

    // (WS: `tid` is the index inside the role's 256 threads -- the staging code and the multiply / epilogue code both count to 256)
    const bool producer = WS && threadIdx.x >= 256;
    const int tid = WS ? (threadIdx.x & 255) : threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kh = lane >> 5;
#ifdef CAPHN_GEMM_PROFILE
    const unsigned long long gstart = clock64();
    const unsigned gwg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (tid == 0 && gwg < 8192) d_gemm_wgt[2 * gwg] = wall_clock64();
#endif
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so neighbouring tiles
    // -- which share an operand panel -- would land on eight different L2s and each fetch the panel from HBM.  Give
    // every XCD a contiguous run of the (n fastest, then m, then k-split) tile order instead (bijective remap for
    // any grid size).  Placement is a speed matter only.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.flags & (1 << 20)) {
        const int nx = gridDim.x, ny = gridDim.y, nwg = nx * ny * (int)gridDim.z;
        const int orig = bx + nx * (by + ny * bz);
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        if (g.flags & (1 << 22)) { by = wgid % ny; bx = (wgid / ny) % nx; }     // m fastest: the B panel of a tile column is reused
        else { bx = wgid % nx; by = (wgid / nx) % ny; }                          // n fastest: the A panel of a tile row is reused
        bz = wgid / (nx * ny);
    }
    const int m0 = by * BM, n0 = bx * BN;
    // row subset: effective extents come from device memory (no host round trip)
    const int* rmapA = nullptr; const int* kmap = nullptr;
    if (g.map_mode == 1) { g.M = min(g.M, g.dev_count[0]); rmapA = g.row_map; if (m0 >= g.M) return; }
    if (g.map_mode == 2) { g.K = min(g.K, g.dev_count[0]); kmap = g.row_map; }

    const int nslab_total = (g.K + BK - 1) / BK;
    int slab0 = 0, slab1 = nslab_total;
    if (g.splitk > 1) {
        const int sps = g.map_mode == 2 ? (nslab_total + g.splitk - 1) / g.splitk : g.slabs_per_split;
        slab0 = bz * sps;
        slab1 = min(nslab_total, slab0 + sps);
        if (slab0 >= slab1) return;
    }

    constexpr bool DUAL = TM * TN == 1;
    f32x16 acc[TM][TN], acc2[DUAL ? TM : 1][DUAL ? TN : 1];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; if (DUAL) acc2[a][b][r] = 0.f; }

    // Software pipeline, two K-slabs of global loads in flight: while slab s is multiplied out of LDS, slab s+1 sits
    // in one register set (issued an iteration ago) and slab s+2 is being requested into the other.  With one slab
    // of look-ahead every workgroup stalled on HBM latency each slab (the loads' L2 hit rate is ~80-85 %, so nearly
    // every 1 KB wave-load waits for at least one miss): SQ_WAIT_ANY was 27-47 % of wave time.
    f32x4 ra0[TileA::NV], rb0[TileB::NV], ra1[TileA::NV], rb1[TileB::NV];
    int pra[TileA::NV], prb[TileB::NV];
    // a K row map (weight gradients over live rows only) goes to LDS once per workgroup, so the mapped GEMM keeps the
    // branch-free path; only TN has both operands K-slow
    const int* kl = nullptr;
    const int kbase = slab0 * BK;
    bool fast = !(g.flags & (1 << 21)) && g.vecA && g.vecB && g.M >= 4 && g.N >= 4 && g.K >= 4 && (g.K & 3) == 0 &&
                (!TA || (g.M & 3) == 0) && (TB || (g.N & 3) == 0);
    if (kmap) {
        const int need = (slab1 - slab0) * BK;
        if (fast && TA && !TB && g.kmap_lds >= need) {
            int* kls = reinterpret_cast<int*>(lds + (DB ? 2 : 1) * TILE_ELEMS);
            for (int i = tid; i < need; i += 256) kls[i] = kmap[min(kbase + i, g.K - 1)];
            __syncthreads();
            kl = kls;
        } else fast = false;
    }
    TileA::phys_rows(pra, g.M, m0, tid, rmapA, fast);
    TileB::phys_rows(prb, g.N, n0, tid, nullptr, fast);
    unsigned boa[TileA::NV], bob[TileB::NV];
    if (fast) { TileA::fast_bases(boa, g.lda, g.M, m0, tid, pra); TileB::fast_bases(bob, g.ldb, g.N, n0, tid, prb); }
    else {
#pragma unroll
        for (int i = 0; i < TileA::NV; ++i) boa[i] = 0;
#pragma unroll
        for (int i = 0; i < TileB::NV; ++i) bob[i] = 0;
    }
    auto gload = [&](f32x4 (&ra)[TileA::NV], f32x4 (&rb)[TileB::NV], int slab, auto fast_c) {
        if constexpr (decltype(fast_c)::value) {
            TileA::load_fast_b(ra, g.A, boa, g.lda, g.K, slab * BK, tid, kl, kbase);
            TileB::load_fast_b(rb, g.B, bob, g.ldb, g.K, slab * BK, tid, kl, kbase);
        } else {
            TileA::load(ra, g.A, g.lda, g.M, g.K, m0, slab * BK, g.vecA, tid, pra, kmap);
            TileB::load(rb, g.B, g.ldb, g.N, g.K, n0, slab * BK, g.vecB, tid, prb, kmap);
        }
    };
    auto multiply = [&](int slab, const __bf16* As, const __bf16* Bs) {
        const int nks = (g.K - slab * BK) > 16 ? 2 : 1;      // skip the all-zero second k-step of a short tail
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks < nks) {
                constexpr int NPL = SINGLE ? 1 : TWO ? 2 : 3;
                bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int p = 0; p < NPL; ++p) fa[a][p] = TileA::frag(As, p, wm * WM + a * 32, ks, lane);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int p = 0; p < NPL; ++p) fb[b][p] = TileB::frag(Bs, p, wn * WN + b * 32, ks, lane);
                if constexpr (SINGLE) {
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
                } else if constexpr (TWO) {
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            f32x16 c = acc[a][b];            // small terms first; dropped: mid.mid and everything with lo (<= 2^-16 relative)
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], c, 0, 0, 0);   // mid . hi
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], c, 0, 0, 0);   // hi  . mid
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], c, 0, 0, 0);   // hi  . hi
                            acc[a][b] = c;
                        }
                } else
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        if constexpr (DUAL) {
                            // one 32x32 tile per wave: twelve MFMAs in a row on one accumulator are a RAW chain
                            // (SQ_WAIT_INST_ANY 47 % of wave time); keep the small cross terms in a second
                            // accumulator -- two independent chains, and the small terms are summed among themselves
                            f32x16 c = acc[a][b], e = acc2[a][b];
                            e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], e, 0, 0, 0);   // lo  . hi
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], c, 0, 0, 0);   // mid . hi
                            e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], e, 0, 0, 0);   // hi  . lo
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], c, 0, 0, 0);   // hi  . mid
                            e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], e, 0, 0, 0);   // mid . mid
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], c, 0, 0, 0);   // hi  . hi
                            acc[a][b] = c; acc2[a][b] = e;
                        } else {
                            f32x16 c = acc[a][b];            // smallest terms first
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
        }
    };
    // fused bias gradient: the n-tile-0 workgroups add up the fp32 A tiles they stage (A is [K, M] here, so a thread
    // always holds the same four columns: 256 % (BM / 4) == 0)
    const bool do_cs = TA && g.colsum_a != nullptr && bx == 0;
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    auto stage = [&](f32x4 (&ra)[TileA::NV], f32x4 (&rb)[TileB::NV], int slab, auto fast_c, __bf16* As, __bf16* Bs) {
        if constexpr (decltype(fast_c)::value) {
            TileA::mask_tail(ra, tid, slab * BK, g.K);
            TileB::mask_tail(rb, tid, slab * BK, g.K);
        }
        if constexpr (TA) {
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < TileA::NV; ++i) csum += ra[i];
            }
        }
        if constexpr (SINGLE) { TileA::store_single(ra, As, tid); TileB::store_single(rb, Bs, tid); }
        else if constexpr (TWO) { TileA::store2(ra, As, tid); TileB::store2(rb, Bs, tid); }
        else { TileA::store(ra, As, tid); TileB::store(rb, Bs, tid); }
    };
    auto mainloop = [&](auto fc) {
        // In the branch-free path the look-ahead loads are issued UNCONDITIONALLY (slab index clamped to the last one,
        // a redundant L2 hit at the tail): only then is the number of loads in flight static and the compiler's wait
        // before staging a set becomes vmcnt(#loads of the newer set) instead of vmcnt(0).
        constexpr bool F = decltype(fc)::value;
        const int last = slab1 - 1;
#ifdef CAPHN_GEMM_PROFILE
        const bool gprof = tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
        unsigned long long gpc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, glast = gprof ? clock64() : 0;
#endif
        gload(ra0, rb0, slab0, fc);
        if (F || slab0 + 1 < slab1) gload(ra1, rb1, min(slab0 + 1, last), fc);
        stage(ra0, rb0, slab0, fc, As, Bs);
        __syncthreads();
        GSTAMP(0);                                  // prologue: first loads, first stage
        for (int slab = slab0; slab < slab1; slab += 2) {
            // LDS: slab.  set 1: slab+1 (in flight).  set 0: free
            if (F || slab + 2 < slab1) gload(ra0, rb0, min(slab + 2, last), fc);
            GSTAMP(1);                              // issue of the look-ahead loads
            multiply(slab, As, Bs);
            __builtin_amdgcn_sched_barrier(0);      // keep the staging (and its vmcnt wait) behind the MFMAs
            GSTAMP(2);                              // fragment reads + MFMAs
            __syncthreads();
            GSTAMP(3);                              // barrier after the multiply
            if (slab + 1 >= slab1) break;
#ifdef CAPHN_GEMM_PROFILE
            if constexpr (F) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TileA::NV + TileB::NV)); GSTAMP(9); }   // pure wait for the older set
#endif
            stage(ra1, rb1, slab + 1, fc, As, Bs);
            GSTAMP(4);                              // wait for the older register set, split, LDS stores
            __syncthreads();
            GSTAMP(5);                              // barrier after the stage
            // LDS: slab+1.  set 0: slab+2 (in flight).  set 1: free
            if (F || slab + 3 < slab1) gload(ra1, rb1, min(slab + 3, last), fc);
            GSTAMP(1);
            multiply(slab + 1, As, Bs);
            __builtin_amdgcn_sched_barrier(0);
            GSTAMP(2);
            __syncthreads();
            GSTAMP(3);
            if (slab + 2 < slab1) {
#ifdef CAPHN_GEMM_PROFILE
                if constexpr (F) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TileA::NV + TileB::NV)); GSTAMP(9); }
#endif
                        stage(ra0, rb0, slab + 2, fc, As, Bs);
                GSTAMP(4);
                __syncthreads();
                GSTAMP(5);
            }
        }
#ifdef CAPHN_GEMM_PROFILE
        if (gprof) { for (int i = 0; i < 6; ++i) atomicAdd(&d_gemm_prof[i], gpc[i]); atomicAdd(&d_gemm_prof[9], gpc[9]); atomicAdd(&d_gemm_prof[6], (unsigned long long)(slab1 - slab0)); atomicAdd(&d_gemm_prof[7], 1ull); }
#endif
    };
    // fused bias gradient from the planes: x = hi + mid + lo
    float csum8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (PL) {
        // pre-split operands: same two-slab look-ahead, registers hold 16-byte bf16 chunks of the three planes
        const __bf16* Ap = static_cast<const __bf16*>(g.Ap);
        const __bf16* Bp = static_cast<const __bf16*>(g.Bp);
        bf16x8 pa0[TileA::NVP][3], pb0[TileB::NVP][3], pa1[TileA::NVP][3], pb1[TileB::NVP][3];
        int ppa[TileA::NVP], ppb[TileB::NVP];
        TileA::phys_rows_p(ppa, g.M, m0, tid, rmapA);
        TileB::phys_rows_p(ppb, g.N, n0, tid, nullptr);
        auto pload = [&](bf16x8 (&ra)[TileA::NVP][3], bf16x8 (&rb)[TileB::NVP][3], int slab) {
            TileA::load_p(ra, Ap, g.ldap, g.psa, g.M, g.K, m0, slab * BK, tid, ppa, kl, kbase);
            TileB::load_p(rb, Bp, g.ldbp, g.psb, g.N, g.K, n0, slab * BK, tid, ppb, kl, kbase);
        };
        auto pstage = [&](const bf16x8 (&ra)[TileA::NVP][3], const bf16x8 (&rb)[TileB::NVP][3], int slab) {
            if constexpr (TA) {
                if (do_cs) {
#pragma unroll
                    for (int i = 0; i < TileA::NVP; ++i) {
                        const int k = slab * BK + (tid + 256 * i) / (BM / 8);
                        if (k < g.K) {
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) {
                                union { bf16x8 v; unsigned short u[8]; } c; c.v = ra[i][pl];
#pragma unroll
                                for (int e = 0; e < 8; ++e) csum8[e] += bf16_bits_to_f32(c.u[e]);
                            }
                        }
                    }
                }
            }
            TileA::store_p(ra, As, tid, slab * BK, g.K);
            TileB::store_p(rb, Bs, tid, slab * BK, g.K);
        };
        const int last = slab1 - 1;
        pload(pa0, pb0, slab0);
        pload(pa1, pb1, min(slab0 + 1, last));
        pstage(pa0, pb0, slab0);
        __syncthreads();
        for (int slab = slab0; slab < slab1; slab += 2) {
            pload(pa0, pb0, min(slab + 2, last));
            multiply(slab, As, Bs);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (slab + 1 >= slab1) break;
            pstage(pa1, pb1, slab + 1);
            __syncthreads();
            pload(pa1, pb1, min(slab + 3, last));
            multiply(slab + 1, As, Bs);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (slab + 2 < slab1) {
                pstage(pa0, pb0, slab + 2);
                __syncthreads();
            }
        }
    } else if constexpr (WS) {
        // Wave-specialised ping-pong: in iteration i the producer waves split and store slab i + 1 into image (i + 1) & 1 (its
        // registers were requested two iterations ago) and request slab i + 3, while the consumer waves multiply slab i out of
        // image i & 1; ONE workgroup barrier per slab orders both hand-overs.  The consumers' instruction stream is fragment reads
        // and MFMAs only -- the split arithmetic (4.5 vector instructions per element) and the global-load waits live in other
        // waves of the same SIMD, so the hardware overlaps them instead of the compiler's schedule inside one wave.
        auto loop = [&](auto fc) {
            constexpr bool F = decltype(fc)::value;
            const int last = slab1 - 1;
            if (producer) {
                gload(ra0, rb0, slab0, fc);
                if (F || slab0 + 1 < slab1) gload(ra1, rb1, min(slab0 + 1, last), fc);
                stage(ra0, rb0, slab0, fc, As, Bs);
                if (F || slab0 + 2 < slab1) gload(ra0, rb0, min(slab0 + 2, last), fc);
            }
            __syncthreads();
            for (int slab = slab0; slab < slab1; slab += 2) {
                if (producer) {
                    if (slab + 1 < slab1) stage(ra1, rb1, slab + 1, fc, As1, Bs1);
                    if (F || slab + 3 < slab1) gload(ra1, rb1, min(slab + 3, last), fc);
                } else multiply(slab, As, Bs);
                __syncthreads();
                if (slab + 1 >= slab1) break;
                if (producer) {
                    if (slab + 2 < slab1) stage(ra0, rb0, slab + 2, fc, As, Bs);
                    if (F || slab + 4 < slab1) gload(ra0, rb0, min(slab + 4, last), fc);
                } else multiply(slab + 1, As1, Bs1);
                __syncthreads();
            }
        };
        if (fast) loop(std::true_type{}); else loop(std::false_type{});
    } else if constexpr (DB) {
        // Ping-pong LDS: slab s is multiplied out of image s & 1 while slab s + 1 is split and stored into the other image -- no
        // barrier between a slab's MFMAs and the next slab's staging, ONE barrier per slab instead of two, and a wave that has
        // issued its MFMAs goes on to the split arithmetic of the next slab while the matrix pipe (its own and the other waves')
        // is still busy.  (Phase stamps of the single-image loop: per slab 1.8 k cycles of fragment reads + MFMAs, 1.9 k of wait +
        // split + LDS stores, back to back, plus two barriers.)
        auto loop = [&](auto fc) {
            constexpr bool F = decltype(fc)::value;
            const int last = slab1 - 1;
            gload(ra0, rb0, slab0, fc);
            if (F || slab0 + 1 < slab1) gload(ra1, rb1, min(slab0 + 1, last), fc);
            stage(ra0, rb0, slab0, fc, As, Bs);
            __syncthreads();
            for (int slab = slab0; slab < slab1; slab += 2) {
                if (F || slab + 2 < slab1) gload(ra0, rb0, min(slab + 2, last), fc);
                multiply(slab, As, Bs);
                __builtin_amdgcn_sched_barrier(0);
                if (slab + 1 < slab1) stage(ra1, rb1, slab + 1, fc, As1, Bs1);
                __syncthreads();
                if (slab + 1 >= slab1) break;
                if (F || slab + 3 < slab1) gload(ra1, rb1, min(slab + 3, last), fc);
                multiply(slab + 1, As1, Bs1);
                __builtin_amdgcn_sched_barrier(0);
                if (slab + 2 < slab1) stage(ra0, rb0, slab + 2, fc, As, Bs);
                __syncthreads();
            }
        };
        if (fast) loop(std::true_type{}); else loop(std::false_type{});
    } else {
        if (fast) mainloop(std::true_type{}); else mainloop(std::false_type{});
    }

    if constexpr (TA && PL) {
        if (do_cs) {        // workgroup-uniform; the main loop ended on a barrier, LDS is free
            constexpr int CG = BM / 8, RG = 256 / CG;      // column groups of 8, threads per group
            float* red = reinterpret_cast<float*>(lds);
            const int cg = tid % CG, rg = tid / CG;
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rg * BM + cg * 8 + e] = csum8[e];
            __syncthreads();
            if (tid < BM && m0 + tid < g.M) {
                float v = 0.f;
                for (int r = 0; r < RG; ++r) v += red[r * BM + tid];
                atomicAdd(g.colsum_a + m0 + tid, v);
            }
        }
    }
    if constexpr (TA && !PL) {
        if (do_cs) {        // workgroup-uniform; the main loop ended on a barrier, LDS is free
            constexpr int CG = BM / 4, RG = 256 / CG;      // column groups, threads per group
            float* red = reinterpret_cast<float*>(lds);
            const int cg = tid % CG, rg = tid / CG;
            const bool mine = !WS || producer;             // (WS: the staging waves hold the column sums)
            if (mine) {
#pragma unroll
                for (int e = 0; e < 4; ++e) red[rg * BM + cg * 4 + e] = csum[e];
            }
            __syncthreads();
            if (mine && tid < BM && m0 + tid < g.M) {
                float v = 0.f;
                for (int r = 0; r < RG; ++r) v += red[r * BM + tid];
                atomicAdd(g.colsum_a + m0 + tid, v);
            }
        }
    }

    if (WS && producer) return;              // (no barrier below this point)
    if constexpr (DUAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += acc2[0][0][r];
    }
    // epilogue: acc register r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
    const bool atomic = g.splitk > 1;
    const bool add_bias = (g.flags & CAPHN_GEMM_BIAS) && (!atomic || bz == 0);
    // Interior tiles (every tile but the last row / column of tiles): straight-line code -- the flags are tested once per
    // workgroup, the row pointers are formed once per 32-row block.  The general loop below (a bounds test, four flag tests and
    // a 64-bit multiply PER ELEMENT, under a changing EXEC mask) took 34 k of a vocabulary-logits workgroup's 72 k cycles, as
    // much as its whole main loop; this path takes 12 k.
    if (m0 + BM <= g.M && n0 + BN <= g.N) {
        const bool relu = (g.flags & CAPHN_GEMM_RELU) != 0, lrelu = (g.flags & CAPHN_GEMM_LRELU) != 0;
        const bool accum = (g.flags & CAPHN_GEMM_ACCUM) != 0, masked = (g.flags & CAPHN_GEMM_MASK) != 0;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int row0 = m0 + wm * WM + a * 32 + 4 * kh;        // register r holds row row0 + (r & 3) + 8 (r >> 2)
            float* crow[16];
            if (rmapA) {
#pragma unroll
                for (int r = 0; r < 16; ++r) crow[r] = g.C + (size_t)rmapA[row0 + (r & 3) + 8 * (r >> 2)] * g.ldc;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) crow[r] = g.C + (size_t)(row0 + (r & 3) + 8 * (r >> 2)) * g.ldc;
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = n0 + wn * WN + b * 32 + li;
                const float bv = add_bias ? g.bias[col] : 0.f;
                if (atomic) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) atomicAdd(crow[r] + col, acc[a][b][r] + bv);
                } else {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = acc[a][b][r] + bv;
                    if (accum) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] += crow[r][col];
                    }
                    if (relu) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if (lrelu) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.01f * v[r];
                    }
                    if (masked) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            v[r] = (g.mask[(size_t)(row0 + (r & 3) + 8 * (r >> 2)) * g.ldmask + col] > 0.f) ? v[r] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) crow[r][col] = v[r];
                }
            }
        }
    } else
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
                if (atomic) { atomicAdd(c, v); continue; }
                if (g.flags & CAPHN_GEMM_ACCUM) v += *c;
                if (g.flags & CAPHN_GEMM_RELU) v = fmaxf(v, 0.f);
                if (g.flags & CAPHN_GEMM_LRELU) v = v > 0.f ? v : 0.01f * v;
                if (g.flags & CAPHN_GEMM_MASK) v = (g.mask[(size_t)row * g.ldmask + col] > 0.f) ? v : 0.f;
                *c = v;
            }
        }
#ifdef CAPHN_GEMM_PROFILE
    if (tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);          // the epilogue's stores have left the wave
        if (gwg < 8192) d_gemm_wgt[2 * gwg + 1] = wall_clock64();
        if (gwg == 0) atomicAdd(&d_gemm_prof[8], clock64() - gstart);
    }
#endif
}

template <int BM, int BN, bool TA, bool TB, int MODE>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmArgs g) { gemm_bf16x3_body<BM, BN, TA, TB, MODE>(g); }
// occupancy experiment (caphn_tune key 25): the same body compiled for 5 / 6 waves per SIMD (<= 96 / 80 VGPRs instead of 112-122):
// the kernel is bound by per-slab latencies, which more resident workgroups per CU could hide
template <int BM, int BN, bool TA, bool TB, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void gemm_bf16x3_kernel_w5(GemmArgs g) { gemm_bf16x3_body<BM, BN, TA, TB, MODE>(g); }
template <int BM, int BN, bool TA, bool TB, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void gemm_bf16x3_kernel_w6(GemmArgs g) { gemm_bf16x3_body<BM, BN, TA, TB, MODE>(g); }

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(512) void gemm_bf16x3_kernel_ws(GemmArgs g) { gemm_bf16x3_body<BM, BN, TA, TB, 5>(g); }

template <int BM, int BN, bool TA, bool TB, int PL>
int launch_one(const GemmArgs& g, hipStream_t s) {
    using TileA = TileS<BM, !TA>;
    using TileB = TileS<BN, TB>;
    constexpr size_t lds_tiles = sizeof(__bf16) * (TileA::ELEMS + TileB::ELEMS) * ((PL == 3 || PL == 5) ? 2 : 1);
    size_t lds = lds_tiles;
    if (g.kmap_lds > 0) lds += sizeof(int) * (size_t)g.kmap_lds;
    static bool attr_set[64];              // one flag per instantiation AND device (the attribute is per device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CAPHN_ELAUNCH;
    if (lds_tiles + 16384 > 48 * 1024 && !attr_set[dev]) {
        const void* fn = PL == 5 ? reinterpret_cast<const void*>(gemm_bf16x3_kernel_ws<BM, BN, TA, TB>)
                                 : reinterpret_cast<const void*>(gemm_bf16x3_kernel<BM, BN, TA, TB, PL == 5 ? 0 : PL>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_tiles + 16384)) != hipSuccess) return CAPHN_ELAUNCH;
        attr_set[dev] = true;
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.splitk > 1 ? g.splitk : 1);
    if constexpr (PL == 5) {
        hipLaunchKernelGGL((gemm_bf16x3_kernel_ws<BM, BN, TA, TB>), grid, dim3(512), lds, s, g);
        return caphn_launch_status();
    } else {
    if constexpr (BM == 64 && PL == 0) {
        if (g_tune_gemm_waves == 5) { hipLaunchKernelGGL((gemm_bf16x3_kernel_w5<BM, BN, TA, TB, PL>), grid, dim3(256), lds, s, g); return caphn_launch_status(); }
        if (g_tune_gemm_waves == 6) { hipLaunchKernelGGL((gemm_bf16x3_kernel_w6<BM, BN, TA, TB, PL>), grid, dim3(256), lds, s, g); return caphn_launch_status(); }
    }
    hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, BN, TA, TB, PL>), grid, dim3(256), lds, s, g);
    return caphn_launch_status();
    }
}
template <int BM, int BN, int PL>
int launch_cfg(const GemmArgs& g, int ta, int tb, hipStream_t s) {
    if (!ta && tb) return launch_one<BM, BN, false, true, PL>(g, s);
    if (!ta && !tb) return launch_one<BM, BN, false, false, PL>(g, s);
    if (ta && !tb) return launch_one<BM, BN, true, false, PL>(g, s);
    if constexpr (PL != 0) return CAPHN_EINVAL;      // TT is never pre-split / reduced (no caller)
    else return launch_one<BM, BN, true, true, 0>(g, s);
}

}  // namespace

extern int g_tune_gemm_planes;
int g_tune_gemm_tile = 0;     // 0: automatic tile choice; 64 / 128: forced (experiments)
int g_tune_gemm_single = 0;   // reduced-precision side modes (caphn_tune key 11): 1 one bf16 product per contraction instead of six; 2 "bf16x2" -- operands
                              // as two bf16 planes (16 significand bits), three products: ~2e-5 relative, inside the north star's 1e-4 on logits
int g_tune_gemm_order = 1;    // tile walk inside an XCD: 0 n fastest always, 1 (default) m fastest when B outgrows L2 and A is the smaller, 2 m fastest always
int g_tune_gemm_xcd = 1;      // 1 (default): XCD-aware tile order
int g_tune_gemm_waves = 0;    // 0 (default): as the compiler allocates (4 waves per SIMD); 5 / 6: the 64x64 six-product kernel compiled for that occupancy
int g_tune_gemm_ws = 0;       // wave-specialised kernel (MODE 5; caphn_tune key 29): 0 off, 64 / 128: that tile for every layout
int g_tune_gemm_db = 0;       // layouts that run the 64x64 tile with ping-pong LDS images (MODE 3): bit 0 NT, bit 1 NN, bit 2 TN
int g_tune_gemm_fast = 1;     // 1 (default): branch-free loads (static vmcnt) where alignment allows
// Tile choice: 128x128 when that alone gives >= 512 workgroups and K >= 512, else 64x64 (four workgroups per CU:
// caphn_debug_gemm_occupancy).  Measured and dropped: 64x256 / 256x64 tiles spanning the whole short side of the N = 200 problems
// (the long operand is read once, but two workgroups per CU hide far less latency: 1.3-2x slower) and 128x64 (round 1: no change;
// round 2, after the epilogue fix: three workgroups per CU, +50 us per step).
// pre-split operands: usable when every 16-byte chunk is aligned and whole (K % 8 == 0; K-slow extents may end inside
// the padding of their leading dimension) and, with a K map, when its slice fits the LDS (the branch-free path)
bool caphn_gemm_planes_ok(const GemmArgs& g, int ta, int tb) {
    bool pl = g_tune_gemm_planes && g.Ap && g.Bp && !(ta && tb) && (g.K % 8) == 0 && g.K >= 8 && (g.ldap % 8) == 0 && (g.ldbp % 8) == 0 &&
              (g.psa % 8) == 0 && (g.psb % 8) == 0 && caphn_aligned16(g.Ap) && caphn_aligned16(g.Bp);
    if (pl && ta && (((g.M + 7) & ~7) > g.ldap || g.M < 8)) pl = false;
    if (pl && !tb && (((g.N + 7) & ~7) > g.ldbp || g.N < 8)) pl = false;
    if (pl && g.map_mode == 2) {
        const int nslab = (g.K + 31) / 32, sk = g.splitk > 1 ? g.splitk : 1;
        if (!(ta && !tb) || ((nslab + sk - 1) / sk) * 32 > 4096) pl = false;
    }
    return pl;
}
// resident workgroups per CU of the default-mode kernel for an operand layout (tools: how many rounds a launch takes)
template <int BM, int BN, bool TA, bool TB>
static int occ_one() {
    using TileA = TileS<BM, !TA>;
    using TileB = TileS<BN, TB>;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(gemm_bf16x3_kernel<BM, BN, TA, TB, 0>), 256,
                                                     sizeof(__bf16) * (TileA::ELEMS + TileB::ELEMS)) != hipSuccess) return -1;
    return n;
}
extern "C" int caphn_debug_gemm_occupancy(int ta, int tb, int tile) {
    if (tile == 128) return !ta && tb ? occ_one<128, 128, false, true>() : !ta ? occ_one<128, 128, false, false>() : !tb ? occ_one<128, 128, true, false>() : occ_one<128, 128, true, true>();
    return !ta && tb ? occ_one<64, 64, false, true>() : !ta ? occ_one<64, 64, false, false>() : !tb ? occ_one<64, 64, true, false>() : occ_one<64, 64, true, true>();
}
int caphn_gemm_bf16x3_launch(GemmArgs g, int ta, int tb, hipStream_t s) {
    if (g_tune_gemm_xcd) g.flags |= 1 << 20;
    g.kmap_lds = 0;
    if (g.map_mode == 2 && ta && !tb) {      // LDS slice of the K map: slabs per split (upper bound from the host-side K) x 32
        const int nslab = (g.K + 31) / 32, sk = g.splitk > 1 ? g.splitk : 1;
        const int ints = ((nslab + sk - 1) / sk) * 32;
        if (ints <= 4096) g.kmap_lds = ints;      // 16 KB at most next to the 30 KB of tiles (64x64 configuration)
    }
    if (!g_tune_gemm_fast) g.flags |= 1 << 21;
    {   // the branch-free path addresses with 32-bit byte offsets: operands of 4 GB or more take the general path
        const double ea = ta ? (double)g.K * g.lda : (double)g.M * g.lda, eb = tb ? (double)g.N * g.ldb : (double)g.K * g.ldb;
        if (ea * 4.0 >= 4.0e9 || eb * 4.0 >= 4.0e9) g.flags |= 1 << 21;
    }
    // Walk order inside an XCD's run of tiles.  n fastest re-reads every B panel once per tile row: fine while all of B
    // (N x K) stays in the XCD's 4 MB L2, but the vocabulary projection's B is 7.7 MB while its A is 2 MB -- every tile row
    // streamed B again from beyond L2 (160 MB of fills for a 99 MB output).  m fastest keeps A resident and reads B once.
    if (g_tune_gemm_order != 0 && g.splitk <= 1) {
        const double abytes = 4.0 * (double)g.M * g.K, bbytes = 4.0 * (double)g.N * g.K, l2 = 3.0 * 1024 * 1024;
        if (g_tune_gemm_order == 2 || (bbytes > l2 && abytes < bbytes)) g.flags |= 1 << 22;
    }
    long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * g.splitk;
    // K of a few slabs (the vocabulary projection, K = 200): the product is output-bound, a workgroup lives ~20 us and the
    // kernel time is rounds x lifetime -- 1140 128x128 workgroups on 512 slots are 2.2 rounds, the same tiles as 64x64 fill the
    // chip evenly (step 1.771/1.822 -> 1.762/1.811 ms).  The big tile pays off when K is long enough to amortise its epilogue.
    if (g.K < 512) tiles128 = 0;
    if (g_tune_gemm_tile == 64) tiles128 = 0; else if (g_tune_gemm_tile == 128) tiles128 = 1 << 20;      // A/B experiments (caphn_tune key 12)
    const bool pl = caphn_gemm_planes_ok(g, ta, tb);
    if (!pl && g_tune_gemm_single == 0 && g_tune_gemm_tile == 0 && g_tune_gemm_ws == 0) {
        // short-K NT products (vocabulary logits and the other K = 200 GEMMs of the forward): A's fragments resident in registers
        const int rc = caphn_gemm_kres_launch(g, ta, tb, s);
        if (rc != 1) return rc;
    }
    if (pl) {
        if (tiles128 >= 512) return launch_cfg<128, 128, 1>(g, ta, tb, s);
        return launch_cfg<64, 64, 1>(g, ta, tb, s);
    }
    if (g_tune_gemm_single == 1 && !(ta && tb)) {
        if (tiles128 >= 512) return launch_cfg<128, 128, 2>(g, ta, tb, s);
        return launch_cfg<64, 64, 2>(g, ta, tb, s);
    }
    if (g_tune_gemm_single == 2 && !(ta && tb)) {       // bf16x2: two planes, three products
        if (tiles128 >= 512) return launch_cfg<128, 128, 4>(g, ta, tb, s);
        return launch_cfg<64, 64, 4>(g, ta, tb, s);
    }
    if (g_tune_gemm_ws == 128 && !(ta && tb)) return launch_cfg<128, 128, 5>(g, ta, tb, s);
    if (g_tune_gemm_ws == 64 && !(ta && tb)) return launch_cfg<64, 64, 5>(g, ta, tb, s);
    if (tiles128 >= 512) return launch_cfg<128, 128, 0>(g, ta, tb, s);
    // ping-pong LDS images for the 64x64 tile (caphn_tune key 23: bit 0 NT, bit 1 NN, bit 2 TN layouts)
    const int lay = (!ta && tb) ? 1 : (!ta && !tb) ? 2 : (ta && !tb) ? 4 : 0;
    if (g_tune_gemm_db & lay) return launch_cfg<64, 64, 3>(g, ta, tb, s);
    return launch_cfg<64, 64, 0>(g, ta, tb, s);
}

// ---------------------------------------------------------------------------------------------- operand splitting
namespace {
struct SplitJobs { SplitJob j[8]; int n; long block0[9]; };
// one workgroup = 256 threads x 8 elements of one row-major matrix; a row's tail chunk is padded with zeros up to ldp
__global__ __launch_bounds__(256) void split3_kernel(SplitJobs jobs) {
    int ji = 0;
    for (int i = 1; i < jobs.n; ++i) if ((long)blockIdx.x >= jobs.block0[i]) ji = i;
    const SplitJob& J = jobs.j[ji];
    const int cpr = (J.cols + 7) >> 3;                        // chunks per row
    const long nchunk = (long)(J.rows + J.zero_rows) * cpr;
    const long stride = (jobs.block0[ji + 1] - jobs.block0[ji]) * 256;
    __bf16* dst = static_cast<__bf16*>(J.dst);
    const bool vec = (J.ld & 3) == 0 && caphn_aligned16_dev(J.src);
    for (long c = ((long)blockIdx.x - jobs.block0[ji]) * 256 + threadIdx.x; c < nchunk; c += stride) {
        const int row = (int)(c / cpr), col = (int)(c % cpr) * 8;
        const float* sp = J.src + (size_t)row * J.ld + col;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (row >= J.rows) { }                      // padding rows: zeros (a K extent rounded up to 8)
        else if (vec && col + 8 <= J.cols) { v0 = *reinterpret_cast<const f32x4*>(sp); v1 = *reinterpret_cast<const f32x4*>(sp + 4); }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (col + e < J.cols) v0[e] = sp[e]; if (col + 4 + e < J.cols) v1[e] = sp[4 + e]; }
        }
        const Split4 a = split3(v0), b = split3(v1);
        __bf16* dp = dst + (size_t)row * J.ldp + col;
        *reinterpret_cast<bf16x8*>(dp) = __builtin_shufflevector(a.hi, b.hi, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8*>(dp + J.ps) = __builtin_shufflevector(a.mid, b.mid, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8*>(dp + 2 * J.ps) = __builtin_shufflevector(a.lo, b.lo, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
}  // namespace
int g_tune_gemm_planes = 1;      // 1 (default): use pre-split operands where the caller provides them
int caphn_split3_launch(const SplitJob* jobs, int n, hipStream_t s) {
    if (n <= 0) return CAPHN_OK;
    if (n > 8 || !jobs) return CAPHN_EINVAL;
    SplitJobs J; J.n = n;
    long b = 0;
    for (int i = 0; i < n; ++i) {
        const SplitJob& j = jobs[i];
        if (!j.src || !j.dst || j.rows <= 0 || j.cols <= 0 || (j.ldp % 8) || (j.ps % 8) || j.ldp < ((j.cols + 7) & ~7) ||
            !caphn_aligned16(j.dst)) return CAPHN_EINVAL;
        J.j[i] = j;
        J.block0[i] = b;
        long nb = ((long)(j.rows + j.zero_rows) * ((j.cols + 7) / 8) + 255) / 256;
        if (nb > 2048) nb = 2048;
        b += nb;
    }
    J.block0[n] = b;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)b), dim3(256), 0, s, J);
    return caphn_launch_status();
}
