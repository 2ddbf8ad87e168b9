// The recurrent loops of recurrent_gru.hip with TWO workgroups per caption, so that B = 128 captions occupy all 256
// CUs of an MI355X instead of half of them.  One workgroup per CU is bound by what that CU can stream from L2 per time
// step (W_hh and U_a, 640 KB, at ~31 B/clk); a pair splits the hidden index k in two halves and each half
//   * streams only ITS rows of [U_a; W_hh] (forward) / its rows of the transposed product (backward): half the bytes,
//   * keeps only its columns of the caption's G slab in LDS (59 KB instead of 118 KB),
//   * evaluates the attention scores' tanh only for its k (the score of position p is a sum over k),
// and the two exchange, per time step, one partial vector of P values (scores forward, d alpha backward) and one
// half vector of H/2 values (h forward, the partial dh backward) through L2.
//
// Exchange protocol (MI355X_MICROARCH.md, "handoff-1to1": data-tagged granules): every value travels as ONE naturally
// aligned 8-byte {float, tag} written by a single agent-scope (sc1, write-through) store and polled with agent-scope
// loads by the lane that needs it; the tag is 2 t + 1 / 2 t + 2 for the two exchanges of time step t, unique within a
// launch, under a per-LAUNCH epoch in the tag's upper half (a host counter, 1 .. 32768: a second backward on the same workspace, or
// any relaunch without the clearing launch in between, never matches what an earlier launch left behind), and the exchange
// area is zero when a forward starts (pair_prep_kernel, which precedes every teacher-forced forward on the same stream, clears
// the forward's and the backward's area), so a stale or uninitialised granule never matches.  No fences: a granule
// is self-validating.  Nothing depends on where the two workgroups run; partners are blocks w and w ^ 8, which the
// dispatcher usually places on one XCD (a speed matter only) -- but both must be RESIDENT for either to advance (co-residency
// assumption, include/caphn.h caphn_device_error).  Every poll is bounded in wall-clock time: a partner that never answers
// makes the lane set the device's sticky failure word (the host sees CAPHN_ETIMEOUT at its next call) and continue with NaN,
// so the step's loss and gradients are NaN, not plausible garbage; it never hangs the GPU.
//
// The half that receives overlaps the hand-off with work that does not need it: the forward multiplies its rows with
// ITS OWN half of h while the partner's half is in flight; the backward computes the partner's columns of the
// transposed product first, sends them, and does its own columns while waiting.
#include "common.h"
#include "decoder_internal.h"
#include <atomic>

namespace {

// 512 threads per workgroup: eight waves, two per SIMD, so a lane may hold up to 256 VGPRs -- the batched independent loads
// below need ~150; at 1024 threads (128-VGPR cap) they spilled 500 bytes per lane to scratch and every phase got slower.
constexpr int NT = 512;
// per-phase shader-clock stamps of workgroup 0 (tools/rec_phase_profile.py): compile with -DCAPHN_REC_PROFILE
#ifdef CAPHN_REC_PROFILE
#define PSTAMP(i) do { if (prof_on) { unsigned long long _n = clock64(); pc[i] += _n - plast; plast = _n; } } while (0)
#define PDECL const bool prof_on = (w == 0 && tid == 0 && a.prof != nullptr); \
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = prof_on ? clock64() : 0
#define PFLUSH do { if (prof_on) for (int i = 0; i < 8; ++i) a.prof[i] = pc[i]; } while (0)
#else
#define PSTAMP(i) do { } while (0)
#define PDECL do { } while (0)
#define PFLUSH do { } while (0)
#endif

typedef unsigned long long u64;

// near = the partner runs on the SAME XCD (both read HW_REG_XCC_ID at kernel start and told each other through the safe form):
// the granule is then stored with workgroup scope (sc0: the line STAYS in the XCD's L2) and the partner's sc1 poll is served by
// that L2 in ~200 cycles.  An agent-scope (sc1) store writes through and DROPS the line, so the poll that finds the granule is a
// trip past the L2 (545-900 cycles idle, 2-3 k under load) -- that read, not waiting for the partner, was the ~3 k cycles of a
// hand-off (work placed between send and receive did not hide any of it).  Partners on different XCDs keep the sc1 form: an
// L2 is only coherent for its own XCD's CUs.  Placement decides speed, never correctness.
__device__ int d_pair_opts = 0;       // caphn_tune key 24 (A/B): bit 0 no same-XCD hand-off form
__device__ __forceinline__ void xsend(u64* slot, float v, unsigned tag, bool near = false) {
    const u64 bits = ((u64)tag << 32) | (u64)__float_as_uint(v);
    if (near) __hip_atomic_store(slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one safe hand-off at kernel start: do my partner and I share an XCD?  (thread 0 asks, the answer goes through LDS)
__device__ __forceinline__ bool partner_is_near(u64* ctl_mine, u64* ctl_part, unsigned tag, int* err, long long limit, int tid, int* flag_s);
// tag of exchange x (1: h / dh, 2: scores / d alpha) of time step t in the launch with epoch `ep` (already shifted)
__device__ __forceinline__ unsigned xtag(unsigned ep, int t, int x) { return ep | (2u * (unsigned)t + (unsigned)x); }
// caphn_tune key 10 -- experiments and tests only: 1 = do not wait for the partner (timing; WRONG results), 2 = half 1 never
// sends (its partner must time out: tests/test_gpu_pair_envelope.py)
__device__ int d_skip_xrecv = 0;
struct XCtl { int* err; long long limit; };
// slow path of xrecv, out of line: the first poll almost always misses (the partner is a few hundred cycles behind), so this is
// where every hand-off spends its time, but it must not cost the callers registers
__device__ __noinline__ float xrecv_wait(u64* slot, unsigned tag, int* err, long long limit) {
    const long long t0 = wall_clock64();
    for (int spin = 1;; ++spin) {
        __builtin_amdgcn_s_sleep(1);
        const u64 bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(bits >> 32) == tag) return __uint_as_float((unsigned)bits);
        if ((spin & 1023) == 0) {
            // somebody on this device already gave up (the word is sticky): nobody waits out a second bound
            if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
            if (wall_clock64() - t0 > limit) break;
        }
    }
    if (err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return __builtin_nanf("");
}
__device__ __forceinline__ float xrecv(u64* slot, unsigned tag, const XCtl& c) {
    if (d_skip_xrecv == 1) return 0.f;
    const u64 bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(bits >> 32) == tag) return __uint_as_float((unsigned)bits);
    return xrecv_wait(slot, tag, c.err, c.limit);
}

__device__ __forceinline__ bool partner_is_near(u64* ctl_mine, u64* ctl_part, unsigned tag, int* err, long long limit, int tid, int* flag_s) {
    if (tid == 0) {
        const unsigned mine = __builtin_amdgcn_s_getreg(6164);      // hwreg(HW_REG_XCC_ID, 0, 4)
        xsend(ctl_mine, __uint_as_float(mine + 1u), tag);
        const XCtl c{err, limit};
        const unsigned theirs = __float_as_uint(xrecv(ctl_part, tag, c));
        *flag_s = (theirs == mine + 1u && !(d_pair_opts & 1) && d_skip_xrecv == 0) ? 1 : 0;
    }
    __syncthreads();
    return *flag_s != 0;
}

// sums of EIGHT values over the 64 lanes in 10 shuffles instead of 48: three halving exchanges (after them lane l holds
// the partial of value (l >> 3) & 7), then three plain stages over the 8 lanes that share a value.  Returns, in every
// lane, the total of value index (lane >> 3).
// The halving exchanges run on the lane-permute instructions of gfx950, not through ds_bpermute: v_permlane32_swap exchanges the
// upper half of one register with the lower half of another -- give it (A = v[j], B = v[j + half]) and A' + B' is, in the lower
// lanes, A(mine) + A(partner) and, in the upper lanes, B(partner) + B(mine): one instruction replaces two selects and a
// shuffle; v_permlane16_swap does the same between the odd and even 16-lane rows; the last exchange (lanes l ^ 8, inside a
// row) is a DPP row rotation.
__device__ __forceinline__ float wave_sum8(float (&v)[8], int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v[j]), __builtin_bit_cast(int, v[j + 4]), false, false);
        v[j] = __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, v[j]), __builtin_bit_cast(int, v[j + 2]), false, false);
        v[j] = __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
    }
    {
        const bool hi = (lane & 8) != 0;
        const float send = hi ? v[0] : v[1];
        const float keep = hi ? v[1] : v[0];
        const int got = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128 /* row_ror:8 */, 0xF, 0xF, true);
        v[0] = keep + __builtin_bit_cast(float, got);
    }
    return group8_sum(v[0]);
}

struct HalfK { int k0, nk, k0p, nkp; };
__host__ __device__ __forceinline__ int half_a(int H) { int a = (((H + 1) >> 1) + 3) & ~3; return a < H ? a : H; }
__device__ __forceinline__ HalfK half_of(int H, int hh) {
    const int HA = half_a(H);
    HalfK h;
    if (!hh) { h.k0 = 0; h.nk = HA; h.k0p = HA; h.nkp = H - HA; }
    else { h.k0 = HA; h.nk = H - HA; h.k0p = 0; h.nkp = HA; }
    return h;
}
// thread -> (local column kk, group g) map used to split sums over p across thread groups
struct KG { int k, g, ng; };
__device__ __forceinline__ KG kg_map(int tid, int n) {
    KG m;
    m.ng = n >= NT ? 1 : NT / n;
    m.g = n >= NT ? 0 : tid / n;
    m.k = n >= NT ? tid : tid - m.g * n;
    if (m.g >= m.ng) m.g = -1;
    return m;
}
// granules per workgroup: [h / dh half | scores / d alpha | 8 control granules (0: XCC id)]
__host__ __device__ __forceinline__ int xch_stride(int P, int H) { return ((half_a(H) + P + 7) & ~7) + 8; }
__host__ __device__ __forceinline__ int xch_ctl(int P, int H) { return ((half_a(H) + P + 7) & ~7); }

// y[r] = row_r . x + bias_r for the rows r = grp, grp + NT/8, ... < NR of this half's [U_a; W_hh] (8 lanes per row).
// Everything in this kernel is bound by dependent latencies, not by bytes: a load-wait-multiply loop per 16-byte chunk
// cost the same 18 k cycles per time step for 400 rows as for 800.  Here a lane requests all its chunks of TWO rows
// (2 x JM dwordx4, clamped addresses, no branches) before it uses any of them; x comes from LDS once per call.
// Local row r = q' nk + kk: q' = 0 is U_a (-> uah_s[kk]), q' = 1 + q is gate block q of W_hh (-> gh_s[q nk + kk]).
constexpr int BWD_TCR = 0;     // (register-resident rows of the BPTT transposed mat-vec: at 16 / 24 / 32 rows the kernel spills 52 / 104 / 144
                               //  registers -- its other phases already hold ~240; the variant is compiled out)
constexpr int BWD_TCR_UNUSED = 32;    // register-resident rows per thread of the all-on-chip backward variant (128 VGPRs)
constexpr int FULL_RC = 5;     // register sweeps of the all-on-chip forward variant (5 x 64 rows x 28 VGPRs)
constexpr int JM = 7;          // chunks per lane per column block: 8 lanes x 7 chunks x 4 floats = 224 columns per block
// Rows come from the PACKED copy WP [(NG + 1) H][pitch] = [U_a; W_hh] with a row pitch of a multiple of 32 floats: a lane
// group's eight 16-byte chunks of a row are then exactly one 128-byte line.  In the parameters' own layout a row is 800 bytes
// (H = 200), every such 128-byte piece straddles two lines, and the mat-vec ran at half the L2 -> CU rate (31 B/clk).
// ... and PER HALF, in the half's own row order: local row r = q' nk + kk of half hh sits at WP[hh][r] (pair_prep_kernel), so a
// row address is base + r * pitch -- no (q', kk) bookkeeping in the inner loops of the mat-vec and of the transposed mat-vec.
__device__ __forceinline__ const float* pair_row(const float* __restrict__ WPh, int pitch, int r) {
    return WPh + (size_t)r * pitch;
}
__device__ __forceinline__ void pair_matvec(const float* __restrict__ WPh, int pitch, const float* bias_s,
                                            const float* x_s, float* out_s,
                                            int H, int NR, int vec, int grp, int s) {
    constexpr int RS = NT / 8;                  // rows per sweep
    if (vec) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x_s);
        const int n4 = H >> 2;
#pragma unroll 1
        for (int r = grp; r < NR; r += 2 * RS) {
            const bool has_b = r + RS < NR;
            const f32x4* ra = reinterpret_cast<const f32x4*>(pair_row(WPh, pitch, r));
            const f32x4* rb = has_b ? reinterpret_cast<const f32x4*>(pair_row(WPh, pitch, r + RS)) : ra;
            float sa = 0.f, sb = 0.f;
            for (int c0 = 0; c0 < n4; c0 += 8 * JM) {
                f32x4 va[JM], vb[JM], xv[JM];
#pragma unroll
                for (int j = 0; j < JM; ++j) {
                    const int c = c0 + s + 8 * j, cc = min(c, n4 - 1);
                    va[j] = ra[cc]; vb[j] = rb[cc];
                    xv[j] = c < n4 ? x4[cc] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < JM; ++j) {
                    sa += va[j][0] * xv[j][0] + va[j][1] * xv[j][1] + va[j][2] * xv[j][2] + va[j][3] * xv[j][3];
                    sb += vb[j][0] * xv[j][0] + vb[j][1] * xv[j][1] + vb[j][2] * xv[j][2] + vb[j][3] * xv[j][3];
                }
            }
            sa = group8_sum(sa); sb = group8_sum(sb);
            if (s == 0) {       // (biases from LDS: a global load here makes the wave wait for every load issued before it)
                out_s[r] = sa + bias_s[r];
                if (has_b) out_s[r + RS] = sb + bias_s[r + RS];
            }
        }
    } else {
        for (int r = grp; r < NR; r += RS) {
            const float* row = pair_row(WPh, pitch, r);
            float sum = 0.f;
            for (int c = s; c < H; c += 8) sum += row[c] * x_s[c];
            sum = group8_sum(sum);
            if (s == 0) out_s[r] = sum + bias_s[r];
        }
    }
}

// The same product with part of the rows ON CHIP for the whole kernel (the weights do not change over the time steps, and at
// ~26 B/clk per CU the 320 KB a half streams per step are 12 k of its 26 k cycles): the rows of the first RC sweeps live in the
// lane group's own registers (RC x JM dwordx4 per lane), the next `NL` rows (slot = row order after those sweeps) in LDS, the
// rest streams from L2 as above.  Needs H % 4 == 0 and H <= 32 JM (one column block).
// STREAM = false: the launcher found room for ALL rows (RC register sweeps + NL rows of LDS >= NR): no global load in the time loop
template <int RC, bool STREAM>
__device__ __forceinline__ void pair_matvec_cached(const float* __restrict__ WPh, int pitch, const float* bias_s,
                                                   const float* x_s, float* out_s,
                                                   int H, int NR, int grp, int s, const f32x4 (&wc)[RC][JM],
                                                   const float* Wc_s, int NL) {
    constexpr int RS = NT / 8;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x_s);
    const int n4 = H >> 2;
    f32x4 xv[JM];
#pragma unroll
    for (int j = 0; j < JM; ++j) { const int c = s + 8 * j; xv[j] = c < n4 ? x4[min(c, n4 - 1)] : f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto dot = [&](const f32x4 (&v)[JM]) {
        float sa = 0.f;
#pragma unroll
        for (int j = 0; j < JM; ++j) sa += v[j][0] * xv[j][0] + v[j][1] * xv[j][1] + v[j][2] * xv[j][2] + v[j][3] * xv[j][3];
        return group8_sum(sa);
    };
    // (out_s = [uah ; gh] in local row order: a result goes to out_s[r].  The (q', kk) form of this -- a divergent subtract
    //  loop and two address forms per row -- was ~100 of the ~150 instructions of a register sweep)
    auto emit = [&](int r, float sum) { if (s == 0 && r < NR) out_s[r] = sum + bias_s[r]; };
    // the first two streamed rows are requested BEFORE the on-chip rows are multiplied: their L2 round trip runs under that work
    f32x4 va[STREAM ? JM : 1], vb[STREAM ? JM : 1];
    auto request = [&](int i) {
        if constexpr (STREAM) {
            const f32x4* pa = reinterpret_cast<const f32x4*>(pair_row(WPh, pitch, min(RS * i + grp, NR - 1)));
            const f32x4* pb = reinterpret_cast<const f32x4*>(pair_row(WPh, pitch, min(RS * (i + 1) + grp, NR - 1)));
#pragma unroll
            for (int j = 0; j < JM; ++j) { va[j] = pa[s + 8 * j]; vb[j] = pb[s + 8 * j]; }     // pad columns are zeros
        }
    };
    int ig = RC + (NL > grp ? (NL - grp + RS - 1) / RS : 0);       // first streamed sweep of this lane group
    if constexpr (STREAM) {   // (one row only: a pair in flight on top of the resident rows does not fit the 256 registers)
        const f32x4* pa = reinterpret_cast<const f32x4*>(pair_row(WPh, pitch, min(RS * ig + grp, NR - 1)));
#pragma unroll
        for (int j = 0; j < JM; ++j) va[j] = pa[s + 8 * j];
    }
#pragma unroll
    for (int i = 0; i < RC; ++i) emit(RS * i + grp, dot(wc[i]));
    {
        int i = RC;
#pragma unroll 1
        for (int slot = grp; slot < NL; slot += RS, ++i) {
            const f32x4* row = reinterpret_cast<const f32x4*>(Wc_s + (size_t)slot * H);
            float sa = 0.f;
#pragma unroll
            for (int j = 0; j < JM; ++j) {
                const f32x4 v = row[min(s + 8 * j, n4 - 1)];
                sa += v[0] * xv[j][0] + v[1] * xv[j][1] + v[2] * xv[j][2] + v[3] * xv[j][3];
            }
            emit(RS * i + grp, group8_sum(sa));
        }
    }
    if constexpr (STREAM) {
        if (RS * ig + grp < NR) {
            const float sa = dot(va);
            const int ra = RS * ig + grp;
            ig += 1;
            if (RS * ig + grp < NR) request(ig);
            emit(ra, sa);
        }
#pragma unroll 1
        while (RS * ig + grp < NR) {
            const float sa = dot(va), sb = dot(vb);
            const int ra = RS * ig + grp;
            ig += 2;
            if (RS * ig + grp < NR) request(ig);
            emit(ra, sa); emit(ra + RS, sb);
        }
    }
}

// CACHED: part of the weight rows on chip (pair_matvec_cached); its own instantiation, because a kernel that carries both
// mat-vec variants (and ten positions of D1 in flight) on top of the register-resident rows spills
// CM: 0 all rows streamed, 1 two register sweeps + spare LDS + the rest streamed, 2 EVERYTHING on chip (five register sweeps = 320
// rows, the other rows in LDS: the canonical GRU half has 400) -- the time loop then issues no weight load at all
template <bool LSTM, int CM>
__global__ __launch_bounds__(NT) void rec_pair_fwd_kernel(RecFwdArgs a) {
    constexpr bool CACHED = CM != 0;
    constexpr int NG = LSTM ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (w & 7) + 8 * (w >> 4), hh = (w >> 3) & 1;
    if (b >= a.B) return;                       // (its partner has the same b and leaves too)
    const int P = a.P, H = a.H, GH = NG * a.H, RG = a.RG;
    const int Ppad = (P + 63) & ~63;
    const HalfK hk = half_of(H, hh);
    const int k0 = hk.k0, nk = hk.nk, nkm = half_a(H);
    const float* WPh = a.WP + (size_t)hh * (NG + 1) * nkm * a.wp_pitch;     // my half of the packed [U_a; W_hh], local row order
    float* G_s = lds;                           // [P][RG][nk]
    float* h_s = G_s + (size_t)P * RG * nkm;    // [H]  (16-byte aligned: every size below is a multiple of 4 floats)
    float* va_s = h_s + ((H + 3) & ~3);         // [nk]
    float* c_s = va_s + nkm;
    float* out_s = c_s + nkm;                   // [(NG + 1) nk] the mat-vec's result in local row order: [uah (nk) ; gh (NG x nk)]
    float* uah_s = out_s;
    float* gh_s = out_s + nk;                   // [NG][nk]  (pitch nk)
    float* e_s = out_s + (NG + 1) * nkm;        // [Ppad]
    float* part_s = e_s + Ppad;                 // [ng][NG][nk]
    float* bias_s = part_s + (size_t)kg_map(0, H - nkm > 0 ? H - nkm : 1).ng * NG * nkm;    // [(NG + 1) nkm] b_Ua | b_hh of my rows, row order
    float* Waf_s = bias_s + (size_t)(NG + 1) * nkm;                                          // [P][nkm] my columns of W_a f (waf_lds)
    float* Wc_s = Waf_s + (a.waf_lds ? (size_t)P * nkm : 0);                                 // [wc_rows][H] cached weight rows

    u64* xmine = a.xch + (size_t)w * xch_stride(P, H);
    u64* xpart = a.xch + (size_t)(w ^ 8) * xch_stride(P, H);
    u64* xh_mine = xmine; u64* xe_mine = xmine + nkm;
    u64* xh_part = xpart; u64* xe_part = xpart + nkm;
    const unsigned ep = a.epoch << 16;
    const XCtl xc{a.err, a.xlimit};
    const bool mute = d_skip_xrecv == 2 && hh == 1;
    // (the flag word borrows the first word of the score vector: free until the first time step, a barrier lies between)
    const bool near = partner_is_near(xmine + xch_ctl(P, H), xpart + xch_ctl(P, H), ep | 0xFFFFu, a.err, a.xlimit, tid, reinterpret_cast<int*>(e_s));

    const float* Gb = a.G + (size_t)b * P * GH;
    const int vecS = a.vecS && (nk % 4 == 0) && (k0 % 4 == 0);
    if (vecS) {
        const int n4 = nk >> 2;
        for (int i = tid; i < P * RG * n4; i += NT) {
            const int c = i % n4, pq = i / n4, q = pq % RG, p = pq / RG;
            reinterpret_cast<f32x4*>(G_s + (size_t)pq * nk)[c] = reinterpret_cast<const f32x4*>(Gb + (size_t)p * GH + q * H + k0)[c];
        }
    } else {
        for (int i = tid; i < P * RG * nk; i += NT) {
            const int kk = i % nk, pq = i / nk, q = pq % RG, p = pq / RG;
            G_s[(size_t)pq * nk + kk] = Gb[(size_t)p * GH + q * H + k0 + kk];
        }
    }
    const int t0 = a.t0, t1 = a.t1 > 0 ? a.t1 : a.T, T = a.T;
    for (int k = tid; k < H; k += NT)
        h_s[k] = t0 == 0 ? a.h0[(size_t)b * H + k] : (a.Hsrc ? a.Hsrc : a.Hs)[((size_t)b * T + t0 - 1) * H + k];
    for (int kk = tid; kk < nk; kk += NT) {
        va_s[kk] = a.v_a[k0 + kk];
        c_s[kk] = LSTM ? (t0 == 0 ? a.c0[(size_t)b * H + k0 + kk] : a.Cs[((size_t)b * T + t0 - 1) * H + k0 + kk]) : 0.f;
    }
    const float bva = hh == 0 ? a.b_va[0] : 0.f;          // added once: e = e(half 0) + e(half 1)
    const float* Waf_b = a.Waf + (size_t)b * P * H;
    const KG m = kg_map(tid, nk);
    // my columns of W_a f: t-invariant, 16 values a lane for the whole kernel.  They used to sit in registers; those now hold
    // weight rows (below), and the 20 KB of LDS this takes would only hold 24 of them
    constexpr int WP = 8, WK = 2;
    const bool waf_regs = a.waf_lds && (P <= WP * (NT / 64)) && (nk <= WK * 64);
    for (int r = tid; r < (NG + 1) * nk; r += NT) {
        const int q = r / nk, kk = r - q * nk;
        bias_s[r] = q == 0 ? a.b_Ua[k0 + kk] : a.b_hh[(q - 1) * H + k0 + kk];
    }
    if (a.waf_lds)
        for (int i = tid; i < P * nk; i += NT) { const int p = i / nk, kk = i - p * nk; Waf_s[(size_t)p * nkm + kk] = Waf_b[p * H + k0 + kk]; }
    const int NR = (NG + 1) * nk;               // rows of [U_a; W_hh] this half multiplies
    const int grp = tid >> 3, s8 = tid & 7;
    const int vecW = (H % 4) == 0;              // the packed weights are 128-byte aligned; half boundaries are multiples of 4
    // on-chip part of my rows (pair_matvec_cached): sweeps 0 .. WRC-1 in registers, the next wc_rows rows in LDS
    constexpr int WRC = CM == 2 ? FULL_RC : CACHED ? 2 : 1;
    constexpr bool cached = CACHED;             // the launcher checked H % 4 == 0 and H <= 32 JM
    const int NL = cached ? a.wc_rows : 0;
    f32x4 wc[WRC][JM];
    if (cached) {
        const int n4 = H >> 2;
#pragma unroll
        for (int i = 0; i < WRC; ++i) {
            const f32x4* row = reinterpret_cast<const f32x4*>(pair_row(WPh, a.wp_pitch, min((NT / 8) * i + grp, NR - 1)));
#pragma unroll
            for (int j = 0; j < JM; ++j) wc[i][j] = row[s8 + 8 * j];
        }
        for (int idx = tid; idx < NL * n4; idx += NT) {
            const int slot = idx / n4, c = idx - slot * n4;
            const int r = min((NT / 8) * (WRC + slot / (NT / 8)) + slot % (NT / 8), NR - 1);
            reinterpret_cast<f32x4*>(Wc_s)[idx] = reinterpret_cast<const f32x4*>(pair_row(WPh, a.wp_pitch, r))[c];
        }
    }
    __syncthreads();
    PDECL;

    for (int t = t0; t < t1; ++t) {
        const size_t bt = (size_t)b * T + t;
        // the x-side gate pre-activations of this step: requested now, used in D2 (a global load, ~1 k cycles if waited for)
        float xgq[NG];
#pragma unroll
        for (int q = 0; q < NG; ++q) xgq[q] = tid < nk ? a.Xg[bt * GH + q * H + k0 + tid] : 0.f;
        // A: the partner's half of h arrives (sent at the end of its previous step), then my rows of U_a h + b_Ua and W_hh h + b_hh
        if (t > t0) {
            for (int j = tid; j < hk.nkp; j += NT) h_s[hk.k0p + j] = xrecv(xh_part + j, xtag(ep, t - 1, 1), xc);
            __syncthreads();
        }
        PSTAMP(1);
        if constexpr (cached) pair_matvec_cached<WRC, CM != 2>(WPh, a.wp_pitch, bias_s, h_s, out_s, H, NR, grp, s8, wc, Wc_s, NL);
        else pair_matvec(WPh, a.wp_pitch, bias_s, h_s, out_s, H, NR, vecW, grp, s8);
        __syncthreads();
        PSTAMP(2);
        // B: my part of e_p = v_a . tanh(Waf_p + uah) (+ b_va)
        if (waf_regs) {
            float sc[WP], uq[WK], vq[WK];
#pragma unroll
            for (int q = 0; q < WK; ++q) { const int kk = lane + 64 * q; uq[q] = kk < nk ? uah_s[kk] : 0.f; vq[q] = kk < nk ? va_s[kk] : 0.f; }
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                sc[i] = 0.f;
                const int pc = min(wave + i * (NT / 64), P - 1);                                   // (rows >= P are not stored)
#pragma unroll
                for (int q = 0; q < WK; ++q)
                    sc[i] += vq[q] * caphn_tanh(Waf_s[(size_t)pc * nkm + min(lane + 64 * q, nk - 1)] + uq[q]);   // v = 0 outside my columns
            }
            const float tot = wave_sum8(sc, lane);          // lane l: total of position wave + 8 (l >> 3)
            const int pw = wave + (lane >> 3) * (NT / 64);
            if ((lane & 7) == 0 && pw < P) e_s[pw] = tot + bva;
        } else {
            for (int p = wave; p < P; p += NT / 64) {
                float sc = 0.f;
                for (int kk = lane; kk < nk; kk += 64) sc += va_s[kk] * caphn_tanh(Waf_b[p * H + k0 + kk] + uah_s[kk]);
                sc = wave_sum(sc);
                if (lane == 0) e_s[p] = sc + bva;
            }
        }
        __syncthreads();
        PSTAMP(3);
        // exchange 1: e = e(half 0) + e(half 1), the same sum in the same order on both sides
        if (P <= 64) {
            // ... and the softmax right there, in the registers of wave 0 (both halves compute the same values)
            float mine = 0.f;
            if (wave == 0 && lane < P) {
                mine = e_s[lane];
                if (!mute) xsend(xe_mine + lane, mine, xtag(ep, t, 2), near);
            }
            if (wave == 0) {
                float e = -INFINITY;
                if (lane < P) {
                    const float theirs = xrecv(xe_part + lane, xtag(ep, t, 2), xc);
                    e = hh == 0 ? mine + theirs : theirs + mine;
                }
                const float mx = wave_max(e);
                const float ex = lane < P ? caphn_exp(e - mx) : 0.f;
                const float inv = 1.0f / wave_sum(ex);
                if (lane < P) { const float al = ex * inv; e_s[lane] = al; if (hh == 0) a.alphas[bt * P + lane] = al; }
            }
            PSTAMP(4);
        } else {
        for (int p = tid; p < P; p += NT) {
            const float mine = e_s[p];
            if (!mute) xsend(xe_mine + p, mine, xtag(ep, t, 2), near);
            const float theirs = xrecv(xe_part + p, xtag(ep, t, 2), xc);
            e_s[p] = hh == 0 ? mine + theirs : theirs + mine;
        }
        __syncthreads();
        PSTAMP(4);
        // C: softmax over P (one wave; both halves compute the same values)
        if (wave == 0) {
            float mx = -INFINITY;
            for (int p = lane; p < P; p += 64) mx = fmaxf(mx, e_s[p]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int p = lane; p < P; p += 64) { const float ex = caphn_exp(e_s[p] - mx); e_s[p] = ex; sum += ex; }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            for (int p = lane; p < P; p += 64) { const float al = e_s[p] * inv; e_s[p] = al; if (hh == 0) a.alphas[bt * P + p] = al; }
        }
        }
        __syncthreads();
        PSTAMP(5);
        // D1: partial gi_ctx[q][kk] = sum_{p = g mod ng} alpha_p G_p over thread groups
        if (m.g >= 0) {
            for (int kk = m.k; kk < nk; kk += (m.ng == 1 ? NT : nk)) {
                float accg[NG];
#pragma unroll
                for (int q = 0; q < NG; ++q) accg[q] = 0.f;
                if (RG == NG && (P + m.ng - 1) / m.ng <= 10) {      // all loads of a batch of positions before any use
                    constexpr int DB = CACHED ? 5 : 10;             // (LDS reads: five in flight hide their latency as well)
#pragma unroll 1
                    for (int i0 = 0; i0 < 10; i0 += DB) {
                        float al[DB], gv[DB][NG];
#pragma unroll
                        for (int i = 0; i < DB; ++i) {
                            const int p = m.g + (i0 + i) * m.ng, pc = min(p, P - 1);
                            al[i] = p < P ? e_s[pc] : 0.f;
#pragma unroll
                            for (int q = 0; q < NG; ++q) gv[i][q] = G_s[((size_t)pc * NG + q) * nk + kk];
                        }
#pragma unroll
                        for (int i = 0; i < DB; ++i)
#pragma unroll
                            for (int q = 0; q < NG; ++q) accg[q] += al[i] * gv[i][q];
                    }
                } else
                for (int p = m.g; p < P; p += m.ng) {
                    const float al = e_s[p];
#pragma unroll
                    for (int q = 0; q < NG; ++q)
                        accg[q] += al * (q < RG ? G_s[((size_t)p * RG + q) * nk + kk] : Gb[(size_t)p * GH + q * H + k0 + kk]);
                }
#pragma unroll
                for (int q = 0; q < NG; ++q) part_s[((size_t)m.g * NG + q) * nk + kk] = accg[q];
            }
        }
        __syncthreads();
        PSTAMP(6);
        // D2: gates and h' for my k; h' goes to the partner
        for (int kk = tid; kk < nk; kk += NT) {
            const int k = k0 + kk;
            float pre[NG];
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                float sg = 0.f;
                for (int g = 0; g < m.ng; ++g) sg += part_s[((size_t)g * NG + q) * nk + kk];
                pre[q] = (kk == tid ? xgq[q] : a.Xg[bt * GH + q * H + k]) + sg;
            }
            const float hp = h_s[k];
            float hnew;
            if (LSTM) {
                const float gi = caphn_sigmoid(pre[0] + gh_s[kk]);
                const float gf = caphn_sigmoid(pre[1] + gh_s[nk + kk]);
                const float gg = caphn_tanh(pre[2] + gh_s[2 * nk + kk]);
                const float go = caphn_sigmoid(pre[3 % NG] + gh_s[(3 % NG) * nk + kk]);
                const float cp = c_s[kk];
                const float cn = gf * cp + gi * gg;
                hnew = go * caphn_tanh(cn);
                a.gates[bt * GH + k] = gi; a.gates[bt * GH + H + k] = gf; a.gates[bt * GH + 2 * H + k] = gg;
                a.gates[bt * GH + (3 % NG) * H + k] = go;
                a.Cprev[bt * H + k] = cp; a.Cs[bt * H + k] = cn;
                c_s[kk] = cn;
            } else {
                const float r = caphn_sigmoid(pre[0] + gh_s[kk]);
                const float z = caphn_sigmoid(pre[1] + gh_s[nk + kk]);
                const float hnv = gh_s[2 * nk + kk];
                const float n = caphn_tanh(pre[2] + r * hnv);
                hnew = (1.0f - z) * n + z * hp;
                a.gates[bt * GH + k] = r; a.gates[bt * GH + H + k] = z; a.gates[bt * GH + 2 * H + k] = n;
                a.hn[bt * H + k] = hnv;
            }
            if (a.drop_p > 0.f) hnew *= caphn_keep_scale(a.drop_seed, (unsigned long long)bt * H + k, a.drop_p, 1.0f / (1.0f - a.drop_p));
            // (the hand-off first: the saved activations below are seven more stores in this lane's memory queue)
            if (t + 1 < t1 && !mute) xsend(xh_mine + kk, hnew, xtag(ep, t, 1), near);
            a.Hprev[bt * H + k] = hp;
            a.Hs[bt * H + k] = hnew;
            a.uah[bt * H + k] = uah_s[kk];
            h_s[k] = hnew;
        }
        __syncthreads();
        PSTAMP(7);
    }
    PFLUSH;
}

// ------------------------------------------------------------------------------------------ BPTT, two workgroups per caption
// TCR > 0 (the launcher's choice when the shape allows: float4 column chunks, one chunk per thread, and every row a thread walks is
// either among the LDS-resident ones or among the next TCR): those TCR rows of the transposed mat-vec live in the thread's
// registers for the whole kernel -- the time loop then streams no weight at all (it was 320 rows x 800 B from L2 per step)
template <bool LSTM, int TCR>
__global__ __launch_bounds__(NT) void rec_pair_bwd_kernel(RecBwdArgs a) {
    constexpr int NG = LSTM ? 4 : 3;
    constexpr int PGM = 10;
    constexpr int TU = 8;                       // streamed rows in flight per thread in the transposed mat-vec
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (w & 7) + 8 * (w >> 4), hh = (w >> 3) & 1;
    if (b >= a.B) return;
    const int P = a.P, H = a.H, GH = NG * a.H, T = a.T, RG = a.RG;
    const int Ppad = (P + 63) & ~63;
    const HalfK hk = half_of(H, hh);
    const int k0 = hk.k0, nk = hk.nk, nkm = half_a(H);
    const int H4 = (H + 3) & ~3;
    const float* WPh = a.WP + (size_t)hh * (NG + 1) * nkm * a.wp_pitch;     // my half of the packed [U_a; W_hh], local row order
    float* G_s = lds;                           // [P][RG][nk]
    float* dh_s = G_s + (size_t)P * RG * nkm;   // [nk] carried dh of my k
    float* dc_s = dh_s + nkm;
    float* uah_s = dc_s + nkm;
    float* va_s = uah_s + nkm;
    float* duah_s = va_s + nkm;                 // [nk]      \ one vector [duah ; dgh] of NR entries: local row r = q' nk + kk of the
    float* dgh_s = duah_s + nk;                 // [NG][nk]  / packed [U_a; W_hh] (q' = 0: U_a) meets entry r -- note the pitch nk, not nkm
    float* dgi_s = duah_s + (NG + 1) * nkm;
    float* dal_s = dgi_s + NG * nkm;            // [Ppad]
    float* al_s = dal_s + Ppad;
    float* dhp_s = al_s + Ppad;                 // [H4] partial dh_{t-1} over my rows, all columns
    float* part_s = dhp_s + H4;                 // [max(nslices, ng)][H4]
    float* Wc_s = part_s + (size_t)a.part_rows * H4;     // [wc_rows][H] the first rows of my [U_a; W_hh], resident for the whole kernel

    u64* xmine = a.xch + (size_t)w * xch_stride(P, H);
    u64* xpart = a.xch + (size_t)(w ^ 8) * xch_stride(P, H);
    u64* xh_mine = xmine; u64* xe_mine = xmine + nkm;
    u64* xh_part = xpart; u64* xe_part = xpart + nkm;
    const unsigned ep = a.epoch << 16;
    const XCtl xc{a.err, a.xlimit};
    const bool mute = d_skip_xrecv == 2 && hh == 1;
    const bool near = partner_is_near(xmine + xch_ctl(P, H), xpart + xch_ctl(P, H), ep | 0xFFFFu, a.err, a.xlimit, tid, reinterpret_cast<int*>(dal_s));

    const float* Gb = a.G + (size_t)b * P * GH;
    const int vecS = a.vecS && (nk % 4 == 0) && (k0 % 4 == 0);
    if (vecS) {
        const int n4 = nk >> 2;
        for (int i = tid; i < P * RG * n4; i += NT) {
            const int c = i % n4, pq = i / n4, q = pq % RG, p = pq / RG;
            reinterpret_cast<f32x4*>(G_s + (size_t)pq * nk)[c] = reinterpret_cast<const f32x4*>(Gb + (size_t)p * GH + q * H + k0)[c];
        }
    } else {
        for (int i = tid; i < P * RG * nk; i += NT) {
            const int kk = i % nk, pq = i / nk, q = pq % RG, p = pq / RG;
            G_s[(size_t)pq * nk + kk] = Gb[(size_t)p * GH + q * H + k0 + kk];
        }
    }
    const int bt0 = a.t0, bt1 = a.t1 > 0 ? a.t1 : T;
    for (int kk = tid; kk < nk; kk += NT) {
        dh_s[kk] = 0.f; va_s[kk] = a.v_a[k0 + kk];
        dc_s[kk] = (LSTM && bt1 < T) ? a.dc0[(size_t)b * H + k0 + kk] : 0.f;
    }
    const float* Waf_b = a.Waf + (size_t)b * P * H;
    // transposed mat-vec thread map over ONE half of the columns at a time: chunk of CH columns x row slice
    const int CH = (H % 4) == 0 ? 4 : 1;
    const KG m = kg_map(tid, nk);
    const int NR = (NG + 1) * nk;               // my rows: [dgh (NG nk) ; duah (nk)]
    __syncthreads();
    const bool fuse = a.dWaf != nullptr && m.g >= 0;
    float dw[PGM];
#pragma unroll
    for (int i = 0; i < PGM; ++i) dw[i] = 0.f;
    // my (k, position group) entries of W_a f: t-invariant, ten values a thread for the whole kernel (they were ten global loads
    // per thread and time step in the d(U_a h) phase: this kernel's LDS is taken by the G slab and the weight rows)
    float wafr[PGM];
#pragma unroll
    for (int i = 0; i < PGM; ++i) { const int p = m.g + i * m.ng; wafr[i] = (fuse && p < P) ? Waf_b[p * H + k0 + m.k] : 0.f; }
    float dva = 0.f, dbva = 0.f;
    PDECL;

    // partial of dh_{t-1}[cb .. ce) over my rows -> dhp_s[cb .. ce).  Threads = (column chunk, row slice); a thread walks its
    // slice's rows four at a time (four independent 16-byte loads in flight, no division in the loop)
    // register-resident rows of the transposed mat-vec: row a.wc_rows + slice + u nsl, my column chunk
    f32x4 wreg[TCR > 0 ? TCR : 1];
    const int tc_nch = (H + 3) / 4, tc_nsl = NT / (tc_nch < NT ? tc_nch : NT);
    const int tc_chunk = tid % tc_nch, tc_slice = tid / tc_nch;
    if constexpr (TCR > 0) {
#pragma unroll
        for (int u = 0; u < TCR; ++u) {
            const int r = a.wc_rows + tc_slice + u * tc_nsl;
            wreg[u] = (tc_slice < tc_nsl && r < NR) ? *reinterpret_cast<const f32x4*>(WPh + (size_t)r * a.wp_pitch + tc_chunk * 4)
                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float* dvec_s = duah_s;
    auto tmatvec = [&](int cb, int ce) {
        if constexpr (TCR > 0) {        // (whole rows only: cb == 0, ce == H)
            if (tc_slice < tc_nsl) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int r = tc_slice; r < a.wc_rows;) {        // LDS-resident rows, four at a time (wc_rows is a multiple of 4 nsl)
                    f32x4 wv[4]; float dj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        wv[u] = *reinterpret_cast<const f32x4*>(Wc_s + (size_t)r * H + tc_chunk * 4);
                        dj[u] = dvec_s[r];
                        r += tc_nsl;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { acc[0] += wv[u][0] * dj[u]; acc[1] += wv[u][1] * dj[u]; acc[2] += wv[u][2] * dj[u]; acc[3] += wv[u][3] * dj[u]; }
                }
#pragma unroll
                for (int u0 = 0; u0 < TCR; u0 += 8) {           // register-resident rows: eight LDS reads of d in flight
                    float dj[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int r = a.wc_rows + tc_slice + (u0 + u) * tc_nsl; dj[u] = (u0 + u < TCR && r < NR) ? dvec_s[r] : 0.f; }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u0 + u < TCR) { const f32x4 w = wreg[u0 + u]; acc[0] += w[0] * dj[u]; acc[1] += w[1] * dj[u]; acc[2] += w[2] * dj[u]; acc[3] += w[3] * dj[u]; }
                }
                *reinterpret_cast<f32x4*>(part_s + (size_t)tc_slice * H4 + tc_chunk * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            }
            __syncthreads();
            for (int j = tid; j < H; j += NT) {
                float sum = 0.f;
                for (int sl = 0; sl < tc_nsl; ++sl) sum += part_s[(size_t)sl * H4 + j];
                dhp_s[j] = sum;
            }
            __syncthreads();
            return;
        }
        const int ncol = ce - cb;
        if (ncol <= 0) return;
        const int nch = (ncol + CH - 1) / CH;
        const int nch_eff = min(nch, NT);
        const int nsl = NT / nch_eff;
        const int chunk = tid % nch_eff, slice = tid / nch_eff;
        if (slice < nsl) {
            for (int c = chunk; c < nch; c += nch_eff) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                int r = slice;
                const float* col = WPh + cb + c * CH;        // row r of my half: col + r * pitch (per-half packed copy)
                // rows [0, NLb) come from LDS (NLb is a multiple of 4 nsl: a batch of four is on one side or the other)
                const int NLb = (CH == 4 && cb == 0) ? a.wc_rows : 0;
                while (r < NLb) {               // (NLb <= NR: all four rows exist)
                    f32x4 wv[4]; float dj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        wv[u] = *reinterpret_cast<const f32x4*>(Wc_s + (size_t)r * H + c * 4);
                        dj[u] = dvec_s[r];
                        r += nsl;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { acc[0] += wv[u][0] * dj[u]; acc[1] += wv[u][1] * dj[u]; acc[2] += wv[u][2] * dj[u]; acc[3] += wv[u][3] * dj[u]; }
                }
                // streamed rows, TU at a time: TU independent 16-byte loads in flight per thread (linear addresses: nothing but
                // the row counter lives across the batch -- with the (q', kk) form this loop held 245 VGPRs at four loads)
                while (r < NR) {
                    const float* rp[TU]; float dj[TU];
#pragma unroll
                    for (int u = 0; u < TU; ++u) {
                        const bool ok = r < NR;
                        rp[u] = col + (size_t)(ok ? r : 0) * a.wp_pitch;
                        dj[u] = ok ? dvec_s[r] : 0.f;
                        r += nsl;
                    }
                    if (CH == 4) {
                        f32x4 wv[TU];
#pragma unroll
                        for (int u = 0; u < TU; ++u) wv[u] = *reinterpret_cast<const f32x4*>(rp[u]);
#pragma unroll
                        for (int u = 0; u < TU; ++u) { acc[0] += wv[u][0] * dj[u]; acc[1] += wv[u][1] * dj[u]; acc[2] += wv[u][2] * dj[u]; acc[3] += wv[u][3] * dj[u]; }
                    } else {
#pragma unroll
                        for (int u = 0; u < TU; ++u) acc[0] += rp[u][0] * dj[u];
                    }
                }
                if (CH == 4) *reinterpret_cast<f32x4*>(part_s + (size_t)slice * H4 + c * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                else part_s[(size_t)slice * H4 + c] = acc[0];
            }
        }
        __syncthreads();
        for (int j = tid; j < ncol; j += NT) {
            float sum = 0.f;
            for (int sl = 0; sl < nsl; ++sl) sum += part_s[(size_t)sl * H4 + j];
            dhp_s[cb + j] = sum;
        }
        __syncthreads();
    };
    // the saved activations of a step are requested one step ahead (seven or eight global loads per k; waited for at the
    // top of a step they cost ~2 k cycles): thread kk < nk holds its k's values, thread p < P alpha_p
    if (a.wc_rows > 0) {            // (CH == 4 guaranteed by the launcher: H % 4 == 0)
        const int n4 = H >> 2;
        for (int idx = tid; idx < a.wc_rows * n4; idx += NT) {
            const int r = idx / n4, c = idx - r * n4;
            reinterpret_cast<f32x4*>(Wc_s)[idx] = reinterpret_cast<const f32x4*>(WPh + (size_t)r * a.wp_pitch)[c];
        }
        __syncthreads();
    }
    const bool pfk = nk <= NT && P <= NT;
    float pf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pfa = 0.f;
    auto prefetch = [&](int t) {
        const size_t bt = (size_t)b * T + t;
        if (tid < P) pfa = a.alphas[bt * P + tid];
        if (tid < nk) {
            const int k = k0 + tid;
            pf[0] = a.dHs[bt * H + k]; pf[1] = a.uah[bt * H + k];
            pf[2] = a.gates[bt * GH + k]; pf[3] = a.gates[bt * GH + H + k]; pf[4] = a.gates[bt * GH + 2 * H + k];
            if (LSTM) { pf[5] = a.gates[bt * GH + (3 % NG) * H + k]; pf[6] = a.Cprev[bt * H + k]; pf[7] = a.Cs[bt * H + k]; }
            else { pf[5] = a.hn[bt * H + k]; pf[6] = a.Hprev[bt * H + k]; }
        }
    };
    if (pfk) prefetch(bt1 - 1);

    for (int t = bt1 - 1; t >= bt0; --t) {
        const size_t bt = (size_t)b * T + t;
        if (pfk) { if (tid < P) al_s[tid] = pfa; }
        else for (int p = tid; p < P; p += NT) al_s[p] = a.alphas[bt * P + p];
        // cell backward (pointwise) for my k
        for (int kk = tid; kk < nk; kk += NT) {
            const int k = k0 + kk;
            float dh = dh_s[kk] + (pfk ? pf[0] : a.dHs[bt * H + k]);
            if (a.drop_p > 0.f) dh *= caphn_keep_scale(a.drop_seed, (unsigned long long)bt * H + k, a.drop_p, 1.0f / (1.0f - a.drop_p));
            uah_s[kk] = pfk ? pf[1] : a.uah[bt * H + k];
            if (LSTM) {
                const float gi = pfk ? pf[2] : a.gates[bt * GH + k], gf = pfk ? pf[3] : a.gates[bt * GH + H + k];
                const float gg = pfk ? pf[4] : a.gates[bt * GH + 2 * H + k];
                const float go = pfk ? pf[5] : a.gates[bt * GH + (3 % NG) * H + k];
                const float cp = pfk ? pf[6] : a.Cprev[bt * H + k];
                const float tc = caphn_tanh(pfk ? pf[7] : a.Cs[bt * H + k]);
                const float d_o = dh * tc;
                const float dc = dc_s[kk] + dh * go * (1.0f - tc * tc);
                const float dip = dc * gg * gi * (1.0f - gi);
                const float dfp = dc * cp * gf * (1.0f - gf);
                const float dgp = dc * gi * (1.0f - gg * gg);
                const float dop = d_o * go * (1.0f - go);
                dc_s[kk] = dc * gf;
                dh_s[kk] = 0.f;
                dgi_s[kk] = dip; dgi_s[nk + kk] = dfp; dgi_s[2 * nk + kk] = dgp; dgi_s[(3 % NG) * nk + kk] = dop;
                dgh_s[kk] = dip; dgh_s[nk + kk] = dfp; dgh_s[2 * nk + kk] = dgp; dgh_s[(3 % NG) * nk + kk] = dop;
                a.dgi[bt * GH + k] = dip; a.dgi[bt * GH + H + k] = dfp; a.dgi[bt * GH + 2 * H + k] = dgp;
                a.dgi[bt * GH + (3 % NG) * H + k] = dop;
            } else {
                const float r = pfk ? pf[2] : a.gates[bt * GH + k], z = pfk ? pf[3] : a.gates[bt * GH + H + k];
                const float n = pfk ? pf[4] : a.gates[bt * GH + 2 * H + k];
                const float hnv = pfk ? pf[5] : a.hn[bt * H + k], hp = pfk ? pf[6] : a.Hprev[bt * H + k];
                const float dn = dh * (1.0f - z);
                const float dz = dh * (hp - n);
                const float dnp = dn * (1.0f - n * n);
                const float drp = dnp * hnv * r * (1.0f - r);
                const float dzp = dz * z * (1.0f - z);
                dh_s[kk] = dh * z;
                dgi_s[kk] = drp; dgi_s[nk + kk] = dzp; dgi_s[2 * nk + kk] = dnp;
                dgh_s[kk] = drp; dgh_s[nk + kk] = dzp; dgh_s[2 * nk + kk] = dnp * r;
                a.dgi[bt * GH + k] = drp; a.dgi[bt * GH + H + k] = dzp; a.dgi[bt * GH + 2 * H + k] = dnp;
                a.dgh[bt * GH + k] = drp; a.dgh[bt * GH + H + k] = dzp; a.dgh[bt * GH + 2 * H + k] = dnp * r;
            }
        }
        if (pfk && t > bt0) prefetch(t - 1);
        __syncthreads();
        PSTAMP(0);
        // my part of d alpha_p = G_p . dgi (sum over my k)
        if (RG == NG && P <= 8 * (NT / 64) && nk <= 128) {
            // wave w: positions w, w + 8, ..., all their partial sums first, then ONE eight-value reduction
            float sp[8], dg[NG][2];
#pragma unroll
            for (int q = 0; q < NG; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) { const int kk = lane + 64 * c; dg[q][c] = kk < nk ? dgi_s[q * nk + kk] : 0.f; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pc = min(wave + i * (NT / 64), P - 1);
                float acc1 = 0.f;
#pragma unroll
                for (int q = 0; q < NG; ++q)
#pragma unroll
                    for (int c = 0; c < 2; ++c) { const int kk = min(lane + 64 * c, nk - 1); acc1 += G_s[((size_t)pc * NG + q) * nk + kk] * dg[q][c]; }
                sp[i] = acc1;
            }
            const float tot = wave_sum8(sp, lane);
            const int pw = wave + (lane >> 3) * (NT / 64);
            if ((lane & 7) == 0 && pw < P) dal_s[pw] = tot + ((hh == 0 && a.dalphas) ? a.dalphas[bt * P + pw] : 0.f);
        } else
        for (int p = wave; p < P; p += NT / 64) {
            float sp = 0.f;
            for (int q = 0; q < NG; ++q) {
                if (q < RG) { for (int kk = lane; kk < nk; kk += 64) sp += G_s[((size_t)p * RG + q) * nk + kk] * dgi_s[q * nk + kk]; }
                else { for (int kk = lane; kk < nk; kk += 64) sp += Gb[(size_t)p * GH + q * H + k0 + kk] * dgi_s[q * nk + kk]; }
            }
            sp = wave_sum(sp);
            if (lane == 0) dal_s[p] = sp + ((hh == 0 && a.dalphas) ? a.dalphas[bt * P + p] : 0.f);
        }
        __syncthreads();
        PSTAMP(1);
        // exchange 1: d alpha = part(half 0) + part(half 1)
        if (P <= 64) {
            // ... and the softmax backward right there, in the registers of wave 0 (both halves, same values):
            // de_p = alpha_p (dalpha_p - sum_q alpha_q dalpha_q)
            float mine = 0.f;
            if (wave == 0 && lane < P) {
                mine = dal_s[lane];
                if (!mute) xsend(xe_mine + lane, mine, xtag(ep, t, 2), near);
            }
            if (wave == 0) {
                float da = 0.f, al = 0.f;
                if (lane < P) {
                    al = al_s[lane];
                    const float theirs = xrecv(xe_part + lane, xtag(ep, t, 2), xc);
                    da = hh == 0 ? mine + theirs : theirs + mine;
                }
                const float dot = wave_sum(al * da);
                if (lane < P) { const float de = al * (da - dot); dal_s[lane] = de; if (hh == 0) a.de[bt * P + lane] = de; }
            }
        } else {
        for (int p = tid; p < P; p += NT) {
            const float mine = dal_s[p];
            if (!mute) xsend(xe_mine + p, mine, xtag(ep, t, 2), near);
            const float theirs = xrecv(xe_part + p, xtag(ep, t, 2), xc);
            dal_s[p] = hh == 0 ? mine + theirs : theirs + mine;
        }
        __syncthreads();
        // softmax backward: de_p = alpha_p (dalpha_p - sum_q alpha_q dalpha_q)   (both halves, same values)
        if (wave == 0) {
            float dot = 0.f;
            for (int p = lane; p < P; p += 64) dot += al_s[p] * dal_s[p];
            dot = wave_sum(dot);
            for (int p = lane; p < P; p += 64) {
                const float de = al_s[p] * (dal_s[p] - dot);
                dal_s[p] = de;
                if (hh == 0) a.de[bt * P + p] = de;
            }
        }
        }
        __syncthreads();
        PSTAMP(2);
        // d(U_a h)[k] = v_k sum_p de_p (1 - tanh^2(Waf_pk + uah_k)) for my k; p split over thread groups
        if (fuse) {
            const int kk = m.k;
            const float u = uah_s[kk];
            float sd = 0.f;
#pragma unroll
            for (int i = 0; i < PGM; ++i) {
                const int p = m.g + i * m.ng;
                if (p < P) {
                    const float de = dal_s[p];
                    const float tv = caphn_tanh(wafr[i] + u);
                    const float wv = de * (1.0f - tv * tv);
                    sd += wv; dw[i] += wv; dva += de * tv;
                    if (hh == 0 && kk == 0) dbva += de;
                }
            }
            part_s[(size_t)m.g * H4 + kk] = sd;
        } else if (m.g >= 0) {
            for (int kk = m.k; kk < nk; kk += (m.ng == 1 ? NT : nk)) {
                const float u = uah_s[kk];
                float sd = 0.f;
                for (int p = m.g; p < P; p += m.ng) {
                    const float tv = caphn_tanh(Waf_b[p * H + k0 + kk] + u);
                    sd += dal_s[p] * (1.0f - tv * tv);
                }
                part_s[(size_t)m.g * H4 + kk] = sd;
            }
        }
        __syncthreads();
        PSTAMP(3);
        for (int kk = tid; kk < nk; kk += NT) {
            float sd = 0.f;
            for (int g = 0; g < m.ng; ++g) sd += part_s[(size_t)g * H4 + kk];
            const float du = sd * va_s[kk];
            duah_s[kk] = du;
            a.duah[bt * H + k0 + kk] = du;
        }
        __syncthreads();
        PSTAMP(4);
        // dh_{t-1}[j] = sum over ALL rows of W_hh^T dgh + U_a^T duah; I hold half of the rows.  The partner's columns
        // first (sent at once), then my own while its contribution to them travels
        // (one sweep over whole rows: 800 contiguous bytes per row and one reduction, instead of a sweep per column half --
        //  the hand-off it would have hidden costs ~1 k cycles, the second sweep cost 8 k)
        tmatvec(0, H);
        PSTAMP(5);
        if (!mute) for (int j = tid; j < hk.nkp; j += NT) xsend(xh_mine + j, dhp_s[hk.k0p + j], xtag(ep, t, 1), near);
        PSTAMP(6);
        for (int kk = tid; kk < nk; kk += NT) {
            const float mine = dhp_s[k0 + kk];
            const float theirs = xrecv(xh_part + kk, xtag(ep, t, 1), xc);
            dh_s[kk] += hh == 0 ? mine + theirs : theirs + mine;
        }
        __syncthreads();
        PSTAMP(7);
    }
    PFLUSH;
    for (int kk = tid; kk < nk; kk += NT) {
        a.dh0[(size_t)b * H + k0 + kk] = dh_s[kk];
        if (LSTM) a.dc0[(size_t)b * H + k0 + kk] = dc_s[kk];
    }
    // init_hidden's backward over my k (models/decoderlstm.py:122-135: h0 = init_h(mean_P f)): d mean_f[f] += sum_k dh0[k] W_inith[k][f]
    // (+ dc0 W_initc), eight rows' loads in flight; it was a 128-workgroup kernel of its own on the chain behind this one
    if (a.dmean_part != nullptr && bt0 == 0) {
        const int F = a.F;
        for (int f = tid; f < F; f += NT) {
            float s0 = 0.f;
            int kk = 0;
            for (; kk + 8 <= nk; kk += 8) {
                float wv[8], cv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    wv[u] = a.inith_w[(size_t)(k0 + kk + u) * F + f];
                    cv[u] = LSTM ? a.initc_w[(size_t)(k0 + kk + u) * F + f] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { s0 += dh_s[kk + u] * wv[u]; if (LSTM) s0 += dc_s[kk + u] * cv[u]; }
            }
            for (; kk < nk; ++kk) {
                s0 += dh_s[kk] * a.inith_w[(size_t)(k0 + kk) * F + f];
                if (LSTM) s0 += dc_s[kk] * a.initc_w[(size_t)(k0 + kk) * F + f];
            }
            a.dmean_part[((size_t)hh * a.B + b) * F + f] = s0;
        }
    }
    if (fuse) {
        const float vk = va_s[m.k];
#pragma unroll
        for (int i = 0; i < PGM; ++i) {
            const int p = m.g + i * m.ng;
            if (p < P) a.dWaf[((size_t)b * P + p) * H + k0 + m.k] = dw[i] * vk;
        }
        // d v_a / d b_va partial rows: [b][apart_rows][H + 1]; each half fills its columns of its first ng rows ...
        float* row = a.apart + ((size_t)b * a.apart_rows + m.g) * (H + 1);
        row[k0 + m.k] = dva;
        if (hh == 0 && m.k == 0) row[H] = dbva;
    }
    if (a.dWaf != nullptr) {      // ... and zeros in the rows its map does not reach (the column sum runs over all of them)
        for (int i = tid; i < (a.apart_rows - m.ng) * nk; i += NT) {
            const int g = m.ng + i / nk, kk = i % nk;
            a.apart[((size_t)b * a.apart_rows + g) * (H + 1) + k0 + kk] = 0.f;
        }
        if (hh == 0) for (int g = m.ng + tid; g < a.apart_rows; g += NT) a.apart[((size_t)b * a.apart_rows + g) * (H + 1) + H] = 0.f;
    }
}

constexpr size_t LDS_LIMIT = 160 * 1024;

// one launch in front of the pair forward: the exchange areas (forward and backward of this step) get tag 0 in every
// granule -- written through (agent scope), like every later store to them -- and [U_a; W_hh] is packed to the aligned pitch
__global__ __launch_bounds__(256) void pair_prep_kernel(u64* __restrict__ xch, size_t nxch, const float* __restrict__ U_a,
                                                        const float* __restrict__ W_hh, int H, int rows, int pitch, float* __restrict__ WP,
                                                        float* __restrict__ zbuf, size_t nz) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t j = i0; j < nxch; j += stride) __hip_atomic_store(xch + j, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (a buffer the backward wants zero-filled -- d Hs, accumulated by split-K atomics -- rides along: its own launch sat on the
    //  chain between the loss and the first backward GEMM)
    for (size_t j = i0; j < nz; j += stride) zbuf[j] = 0.f;
    // packed layout: WP[half][local row r = q' nk_half + kk][pitch], q' = 0 the rows of U_a, 1 + q gate block q of W_hh, restricted
    // to the half's k range [k0, k0 + nk); each half has room for (NG + 1) nkm rows (nkm = the wider half)
    const int HA = half_a(H), NGp1 = rows / H, hrows = NGp1 * HA;
    // (W_hh == nullptr -- caphn_decoder_pair_prep: the W_hh rows are written by the optimiser's rank-1 pass; only their pad columns
    //  are cleared here, the U_a rows are packed as usual)
    const bool lite = W_hh == nullptr;
    auto is_whh = [&](int hh, int r) { const int nk = hh ? H - HA : HA; return r >= nk && r < NGp1 * nk; };
    auto src_row = [&](int hh, int r) -> const float* {
        const int k0 = hh ? HA : 0, nk = hh ? H - HA : HA;
        if (r >= NGp1 * nk) return nullptr;
        const int q = r / nk, kk = r - q * nk;
        return q == 0 ? U_a + (size_t)(k0 + kk) * H : W_hh + (size_t)((q - 1) * H + k0 + kk) * H;
    };
    if ((H & 3) == 0 && caphn_aligned16_dev(U_a) && caphn_aligned16_dev(W_hh)) {
        // whole pitch written: the pad columns hold zeros, so a lane may read its chunk without a bounds clamp (one base address
        // and immediate offsets instead of an address pair per chunk)
        const int n4 = H >> 2, p4 = pitch >> 2;
        for (size_t j = i0; j < (size_t)2 * hrows * p4; j += stride) {
            const int rr = (int)(j / p4), c = (int)(j % p4);
            if (lite && is_whh(rr / hrows, rr % hrows)) {
                if (c >= n4) reinterpret_cast<f32x4*>(WP + (size_t)rr * pitch)[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const float* src = src_row(rr / hrows, rr % hrows);
            reinterpret_cast<f32x4*>(WP + (size_t)rr * pitch)[c] = (src && c < n4) ? reinterpret_cast<const f32x4*>(src)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else {
        for (size_t j = i0; j < (size_t)2 * hrows * pitch; j += stride) {
            const int rr = (int)(j / pitch), c = (int)(j % pitch);
            if (lite && is_whh(rr / hrows, rr % hrows)) {
                if (c >= H) WP[(size_t)rr * pitch + c] = 0.f;
                continue;
            }
            const float* src = src_row(rr / hrows, rr % hrows);
            WP[(size_t)rr * pitch + c] = (src && c < H) ? src[c] : 0.f;
        }
    }
}

}  // namespace
// (at least one full column block of the mat-vec, 32 JM floats: see the pad note in pair_prep_kernel)
int caphn_rec_pair_pitch(int H) { const int p = (H + 31) & ~31; return p < 32 * JM ? 32 * JM : p; }
int caphn_rec_pair_half_a(int H) { return half_a(H); }
size_t caphn_rec_pair_wp_floats(int H, int NG) { return (size_t)2 * (NG + 1) * half_a(H) * caphn_rec_pair_pitch(H); }
int caphn_launch_rec_pair_prep(unsigned long long* xch, size_t nxch, const float* U_a, const float* W_hh, int H, int NG, float* WP,
                               float* zbuf, size_t nz, hipStream_t s) {
    hipLaunchKernelGGL(pair_prep_kernel, dim3(256), dim3(256), 0, s, xch, nxch, U_a, W_hh, H, (NG + 1) * H, caphn_rec_pair_pitch(H), WP,
                       zbuf, zbuf ? nz : 0);
    return caphn_launch_status();
}
int caphn_rec_pair_debug_opts(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(d_pair_opts), &v, sizeof(int)) == hipSuccess ? CAPHN_OK : CAPHN_ELAUNCH; }
int caphn_rec_pair_debug_skip(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(d_skip_xrecv), &v, sizeof(int)) == hipSuccess ? CAPHN_OK : CAPHN_ELAUNCH; }

static int kgn(int n) { return n >= NT ? 1 : NT / n; }
int g_tune_rec_cache = 2;   // 2 (default): the forward keeps ALL of a half's [U_a; W_hh] on chip when it fits (GRU, H = 200), else as 1;
                            // 1: part of the recurrent weights on chip for the whole kernel (registers + spare LDS); 0: all streamed
size_t caphn_rec_pair_xch_bytes(int B, int P, int H) {
    const size_t nwg = 16 * (size_t)((B + 7) / 8);
    return nwg * xch_stride(P, H) * sizeof(u64);
}
size_t caphn_rec_pair_fwd_lds_bytes(int P, int H, int NG, int RG) {
    const size_t nkm = half_a(H), Ppad = (P + 63) & ~63;
    const int nkmin = H - half_a(H);
    if (nkmin < 1) return ~(size_t)0;
    return sizeof(float) * ((size_t)RG * P * nkm + ((H + 3) & ~3) + 3 * nkm + (size_t)NG * nkm + Ppad + (size_t)kgn(nkmin) * NG * nkm +
                            (size_t)(NG + 1) * nkm);
}
size_t caphn_rec_pair_bwd_lds_bytes(int P, int H, int NG, int RG) {
    const size_t nkm = half_a(H), Ppad = (P + 63) & ~63, H4 = (H + 3) & ~3;
    const int nkmin = H - half_a(H);
    if (nkmin < 1) return ~(size_t)0;
    // slices of the transposed product: NT / (chunks of the narrower half, scalar columns in the worst case)
    size_t nsl = NT / (size_t)(nkmin < NT ? nkmin : NT);
    const size_t nsl4 = NT / (size_t)(((nkmin + 3) / 4) < NT ? ((nkmin + 3) / 4) : NT);
    if (nsl4 > nsl) nsl = nsl4;
    if ((size_t)kgn(nkmin) > nsl) nsl = kgn(nkmin);
    return sizeof(float) * ((size_t)RG * P * nkm + 5 * nkm + 2 * (size_t)NG * nkm + 2 * Ppad + H4 + nsl * H4);
}
// usable: both halves non-empty, the forward's rows fit the registers kept across the hand-off, everything fits the LDS
bool caphn_rec_pair_ok(int P, int H, int NG, int RG) {
    if (H < 8 || RG < 0) return false;
    return caphn_rec_pair_fwd_lds_bytes(P, H, NG, RG) <= LDS_LIMIT && caphn_rec_pair_bwd_lds_bytes(P, H, NG, RG) <= LDS_LIMIT;
}
int caphn_rec_pair_resident_gates(int P, int H, int NG) {
    for (int rg = NG; rg >= 0; --rg) if (caphn_rec_pair_ok(P, H, NG, rg)) return rg;
    return -1;
}
int caphn_rec_pair_bwd_groups(int P, int H) {
    const int nkmin = H - half_a(H), nkmax = half_a(H);
    if (nkmin < 1 || nkmax > NT) return 0;
    const int ng = NT / nkmax;                  // the wider half has the fewest groups: both must carry P positions in PGM registers
    return (P + ng - 1) / ng <= 10 ? NT / nkmin : 0;      // rows of `apart` per caption: the narrower half has the most groups
}

static int set_attr(const void* f) {
    return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT) == hipSuccess ? CAPHN_OK : CAPHN_ELAUNCH;
}
// the 160 KB dynamic-LDS opt-in is a per-DEVICE function attribute: one guard per device, not per process
static int set_attrs_here() {
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CAPHN_ELAUNCH;
    if (done[dev]) return CAPHN_OK;
    if (set_attr(reinterpret_cast<const void*>(rec_pair_fwd_kernel<false, 0>)) || set_attr(reinterpret_cast<const void*>(rec_pair_fwd_kernel<true, 0>)) ||
        set_attr(reinterpret_cast<const void*>(rec_pair_fwd_kernel<false, 1>)) || set_attr(reinterpret_cast<const void*>(rec_pair_fwd_kernel<true, 1>)) ||
        set_attr(reinterpret_cast<const void*>(rec_pair_fwd_kernel<false, 2>)) ||
        set_attr(reinterpret_cast<const void*>(rec_pair_bwd_kernel<false, 0>)) || set_attr(reinterpret_cast<const void*>(rec_pair_bwd_kernel<true, 0>)) ||
        set_attr(reinterpret_cast<const void*>(rec_pair_bwd_kernel<false, BWD_TCR>)))
        return CAPHN_ELAUNCH;
    done[dev] = true;
    return CAPHN_OK;
}
#define RUN_ATTR() do { if (set_attrs_here() != CAPHN_OK) return CAPHN_ELAUNCH; } while (0)
// per-launch hand-off control: a fresh tag epoch (1 .. 32768 in the tag's upper 16 bits; the lower 16 carry 2 t + x, so T < 32767),
// the device's sticky failure word and the time bound
static int prepare_xch(int T, unsigned* epoch, int** err, long long* limit) {
    static std::atomic<unsigned> ctr{0};
    if (T >= 32767) return CAPHN_ELIMIT;
    *epoch = (ctr.fetch_add(1, std::memory_order_relaxed) & 0x7fffu) + 1u;
    *err = caphn_errword();
    *limit = g_tune_xch_timeout;
    return CAPHN_OK;
}
// rows of the half's [U_a; W_hh] the forward kernel keeps in the LDS left over (a multiple of 8: wave-uniform; at most what
// the two register-resident sweeps leave)
int caphn_rec_pair_fwd_cache_rows(int P, int H, int NG, int RG, int rc = 2) {
    size_t base = caphn_rec_pair_fwd_lds_bytes(P, H, NG, RG);
    if (base > LDS_LIMIT || g_tune_rec_cache == 0) return g_tune_rec_cache == 0 ? -1 : 0;
    if (base + sizeof(float) * (size_t)P * half_a(H) <= LDS_LIMIT) base += sizeof(float) * (size_t)P * half_a(H);      // W_a f columns
    long rows = (long)((LDS_LIMIT - base) / (sizeof(float) * (size_t)H)) & ~7L;
    const long nr = (long)(NG + 1) * half_a(H), left = nr - rc * (NT / 8);
    if (rows > left) rows = left > 0 ? (left + 7) & ~7L : 0;
    if ((size_t)rows * H * sizeof(float) + base > LDS_LIMIT) rows -= 8;
    return rows > 0 ? (int)rows : 0;
}
int caphn_launch_rec_pair_fwd(const RecFwdArgs& a_, bool lstm, hipStream_t s) {
    RecFwdArgs a = a_;
    size_t lds = caphn_rec_pair_fwd_lds_bytes(a.P, a.H, lstm ? 4 : 3, a.RG);
    if (lds > LDS_LIMIT || !a.xch || !a.WP) return CAPHN_ELIMIT;
    const size_t waf_bytes = sizeof(float) * (size_t)a.P * half_a(a.H);
    a.waf_lds = lds + waf_bytes <= LDS_LIMIT ? 1 : 0;
    if (a.waf_lds) lds += waf_bytes;
    a.wc_rows = ((a.H % 4) == 0 && a.H <= 32 * JM) ? caphn_rec_pair_fwd_cache_rows(a.P, a.H, lstm ? 4 : 3, a.RG) : -1;
    // everything on chip?  FULL_RC register sweeps + the LDS rows must cover the wider half's rows (GRU cell only: the LSTM's fourth
    // gate block does not fit).  caphn_tune(16, 1) keeps the partial cache for A/B runs; 2 (default) takes this when it fits
    bool full = false;
    if (!lstm && a.wc_rows >= 0 && g_tune_rec_cache >= 2) {
        const int nr = 4 * half_a(a.H);
        const int rows5 = caphn_rec_pair_fwd_cache_rows(a.P, a.H, 3, a.RG, FULL_RC);
        if (rows5 >= 0 && FULL_RC * (NT / 8) + rows5 >= nr) { full = true; a.wc_rows = rows5; }
    }
    if (a.wc_rows > 0) lds += sizeof(float) * (size_t)a.wc_rows * a.H;
    RUN_ATTR();
    if (prepare_xch(a.T, &a.epoch, &a.err, &a.xlimit) != CAPHN_OK) return CAPHN_ELIMIT;
    const unsigned nwg = 16u * (unsigned)((a.B + 7) / 8);
    const bool cached = a.wc_rows >= 0 && (a.H % 4) == 0 && a.H <= 32 * JM;
    if (cached && full) {
        hipLaunchKernelGGL((rec_pair_fwd_kernel<false, 2>), dim3(nwg), dim3(NT), lds, s, a);
    } else if (cached) {
        if (lstm) hipLaunchKernelGGL((rec_pair_fwd_kernel<true, 1>), dim3(nwg), dim3(NT), lds, s, a);
        else hipLaunchKernelGGL((rec_pair_fwd_kernel<false, 1>), dim3(nwg), dim3(NT), lds, s, a);
    } else {
        if (lstm) hipLaunchKernelGGL((rec_pair_fwd_kernel<true, 0>), dim3(nwg), dim3(NT), lds, s, a);
        else hipLaunchKernelGGL((rec_pair_fwd_kernel<false, 0>), dim3(nwg), dim3(NT), lds, s, a);
    }
    return caphn_launch_status();
}
// rows of part_s the backward kernel lays out (its LDS formula's last term), and the weight rows that fit behind them
static size_t bwd_part_rows(int H) {
    const int nkmin = H - half_a(H);
    size_t nsl = NT / (size_t)(nkmin < NT ? nkmin : NT);
    const size_t nsl4 = NT / (size_t)(((nkmin + 3) / 4) < NT ? ((nkmin + 3) / 4) : NT);
    if (nsl4 > nsl) nsl = nsl4;
    if ((size_t)kgn(nkmin) > nsl) nsl = kgn(nkmin);
    return nsl;
}
int caphn_rec_pair_bwd_cache_rows(int P, int H, int NG, int RG) {
    const size_t base = caphn_rec_pair_bwd_lds_bytes(P, H, NG, RG);
    if (base > LDS_LIMIT || g_tune_rec_cache == 0 || (H % 4) != 0 || H / 4 > NT) return 0;
    const long batch = 4 * (long)(NT / (H / 4));                   // rows a sweep of four loads per thread covers
    long rows = (long)((LDS_LIMIT - base) / (sizeof(float) * (size_t)H));
    const long nr = (long)(NG + 1) * (H - half_a(H));               // the narrower half's row count bounds both
    if (rows > nr) rows = nr;
    rows -= rows % batch;
    return rows > 0 ? (int)rows : 0;
}
int caphn_launch_rec_pair_bwd(const RecBwdArgs& a_, bool lstm, hipStream_t s) {
    RecBwdArgs a = a_;
    size_t lds = caphn_rec_pair_bwd_lds_bytes(a.P, a.H, lstm ? 4 : 3, a.RG);
    if (lds > LDS_LIMIT || !a.xch || !a.WP) return CAPHN_ELIMIT;
    a.part_rows = (int)bwd_part_rows(a.H);
    a.wc_rows = caphn_rec_pair_bwd_cache_rows(a.P, a.H, lstm ? 4 : 3, a.RG);
    lds += sizeof(float) * (size_t)a.wc_rows * a.H;
    if (a.dWaf && (!a.apart || caphn_rec_pair_bwd_groups(a.P, a.H) == 0 || a.apart_rows < caphn_rec_pair_bwd_groups(a.P, a.H))) return CAPHN_EINVAL;
    RUN_ATTR();
    if (prepare_xch(a.T, &a.epoch, &a.err, &a.xlimit) != CAPHN_OK) return CAPHN_ELIMIT;
    const unsigned nwg = 16u * (unsigned)((a.B + 7) / 8);
    // everything on chip?  float4 chunks, one per thread, and LDS rows + BWD_TCR register rows per slice cover the half's rows
    bool full = false;
    if (BWD_TCR > 0 && !lstm && g_tune_rec_cache >= 2 && (a.H % 4) == 0 && a.H / 4 <= NT && a.wc_rows > 0) {
        const int nsl = NT / (a.H / 4), nr = 4 * half_a(a.H);
        full = a.wc_rows + BWD_TCR * nsl >= nr;
    }
    if (full) hipLaunchKernelGGL((rec_pair_bwd_kernel<false, BWD_TCR>), dim3(nwg), dim3(NT), lds, s, a);
    else if (lstm) hipLaunchKernelGGL((rec_pair_bwd_kernel<true, 0>), dim3(nwg), dim3(NT), lds, s, a);
    else hipLaunchKernelGGL((rec_pair_bwd_kernel<false, 0>), dim3(nwg), dim3(NT), lds, s, a);
    return caphn_launch_status();
}
