// Teacher-forced recurrent loop of AttentionGru (models/decoderlstm.py:78-108) fused with
// BahdanauAttention (models/attention.py:21-46) and the GRUCell arithmetic: one persistent
// workgroup per caption walks all T steps (no per-step launches).
//
// MI355X mapping (160 KB LDS per CU, one workgroup per CU):
//  * t-invariant projections are hoisted out of the loop as MFMA GEMMs (decoder.hip):
//      Waf = W_a f + b_Wa  [B,P,H]   and   G = f W_ih[:,E:]^T  [B,P,3H]
//    so the context never has to be multiplied by W_ih inside the loop:
//      gi_ctx = (sum_p alpha_p f_p) W_ih_ctx^T = sum_p alpha_p G_p.
//    A caption's Waf (39 KB) and G (118 KB) slabs are read from HBM once, coalesced, and stay in
//    LDS for all T steps; h, U_a h and the gate pre-activations live in LDS too.
//  * per step the only global traffic is W_hh and U_a (640 KB, L2-resident, shared by all
//    workgroups) streamed with 16-byte loads, 8 lanes per weight row;
//  * attention scores: one wave per position p, lanes over H, v_a . tanh(.) reduced with wave
//    shuffles; the softmax over P = 49 (< 64 lanes) is a single-wave shuffle reduction.
#include "common.h"
#include "decoder_internal.h"

namespace {

constexpr int NT = 1024;
// phase stamp (workgroup 0, thread 0): adds the shader clocks since the previous stamp to counter i
// Per-phase shader-clock stamps of workgroup 0 (tools/rec_phase_profile.py): compile with -DCAPHN_REC_PROFILE.  Off by
// default: the eight 64-bit counters live in VGPRs of every thread (16 registers in kernels that sit at the 128-VGPR cap
// of a 1024-thread workgroup).
#ifdef CAPHN_REC_PROFILE
#define PSTAMP(i) do { if (prof_on) { unsigned long long _n = clock64(); pc[i] += _n - plast; plast = _n; } } while (0)
#define PDECL const bool prof_on = (b == 0 && tid == 0 && a.prof != nullptr); \
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = prof_on ? clock64() : 0
#define PFLUSH do { if (prof_on) for (int i = 0; i < 8; ++i) a.prof[i] = pc[i]; } while (0)
#else
#define PSTAMP(i) do { } while (0)
#define PDECL do { } while (0)
#define PFLUSH do { } while (0)
#endif
//            // 16 waves per workgroup: one workgroup per CU (LDS-bound), 4 waves per SIMD

__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int n, int vec, int tid) {
    if (vec) {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
        f32x4* d4 = reinterpret_cast<f32x4*>(dst);
        for (int i = tid; i < (n >> 2); i += NT) d4[i] = s4[i];
    } else {
        for (int i = tid; i < n; i += NT) dst[i] = src[i];
    }
}

// y[row] = dot(W[row,:], x_s) + bias[row] for `rows` rows; 8 lanes per row, result to out_s
// `rot` rotates the row order per workgroup: all workgroups stream the same L2-resident weights, and in
// lockstep they would hammer the same L2 channels at the same instant.
__device__ __forceinline__ void matvec_rows(const float* __restrict__ W, const float* __restrict__ bias,
                                            const float* x_s, float* out_s, int rows, int K, int vec, int tid, int rot) {
    const int grp = tid >> 3, s = tid & 7;
    for (int r0 = grp; r0 < rows; r0 += NT / 8) {
        int r = r0 + rot;
        if (r >= rows) r -= rows;
        const float* row = W + (size_t)r * K;
        float sum = 0.f;
        if (vec) {
            const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
            const f32x4* x4 = reinterpret_cast<const f32x4*>(x_s);
            for (int c = s; c < (K >> 2); c += 8) {
                f32x4 a = r4[c], b = x4[c];
                sum += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
            }
        } else {
            for (int c = s; c < K; c += 8) sum += row[c] * x_s[c];
        }
        sum += __shfl_xor(sum, 4, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 1, 64);
        if (s == 0) out_s[r] = sum + bias[r];
    }
}

constexpr int PGM = 10;     // positions per thread the BPTT kernel can carry attention-gradient accumulators for
// thread -> (column k, group g) map used to split sums over p across thread groups
struct KG { int k, g, ng; };
__device__ __forceinline__ KG kg_map(int tid, int H) {
    KG m;
    m.ng = H >= NT ? 1 : NT / H;
    m.g = H >= NT ? 0 : tid / H;
    m.k = H >= NT ? tid : tid - m.g * H;
    if (m.g >= m.ng) { m.g = -1; }      // spare threads
    return m;
}

// resident part of G: gate slabs [0, RG) of every position row
__device__ __forceinline__ void load_G_resident(float* G_s, const float* __restrict__ Gb, int P, int GH, int RGH,
                                                int vec, int tid) {
    if (RGH == GH) { copy_to_lds(G_s, Gb, P * GH, vec, tid); return; }
    if (vec) {
        const int r4 = RGH >> 2;
        for (int i = tid; i < P * r4; i += NT) {
            const int p = i / r4, c = i - p * r4;
            reinterpret_cast<f32x4*>(G_s + p * RGH)[c] = reinterpret_cast<const f32x4*>(Gb + (size_t)p * GH)[c];
        }
    } else {
        for (int i = tid; i < P * RGH; i += NT) { const int p = i / RGH, c = i - p * RGH; G_s[i] = Gb[(size_t)p * GH + c]; }
    }
}

template <bool LSTM>
__global__ __launch_bounds__(NT) void rec_attn_fwd_kernel(RecFwdArgs a) {
    constexpr int NG = LSTM ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, H = a.H, GH = NG * a.H, T = a.T, RG = a.RG, RGH = a.RG * a.H;
    const int Ppad = (P + 63) & ~63;
    float* G_s = lds;
    float* h_s = G_s + P * RGH;
    float* uah_s = h_s + H;
    float* va_s = uah_s + H;
    float* c_s = va_s + H;
    float* gh_s = c_s + H;
    float* e_s = gh_s + GH;
    float* part_s = e_s + Ppad;          // [ng][GH]

    const int sb = a.slab_div > 1 ? b / a.slab_div : b;
    const float* Gb = a.G + (size_t)sb * P * GH;
    load_G_resident(G_s, Gb, P, GH, RGH, a.vecS, tid);
    // time-step window [t0, t1) of this launch (t1 == 0: all T steps); t0 > 0 continues from the saved h_{t0-1} (c_{t0-1})
    const int t0 = a.t0, t1 = a.t1 > 0 ? a.t1 : a.T;
    for (int k = tid; k < H; k += NT) {
        h_s[k] = t0 == 0 ? a.h0[(size_t)b * H + k] : (a.Hsrc ? a.Hsrc : a.Hs)[((size_t)b * T + t0 - 1) * H + k];
        va_s[k] = a.v_a[k];
        c_s[k] = LSTM ? (t0 == 0 ? a.c0[(size_t)b * H + k] : a.Cs[((size_t)b * T + t0 - 1) * H + k]) : 0.f;
    }
    const float bva = a.b_va[0];
    const float* Waf_b = a.Waf + (size_t)sb * P * H;
    const KG m = kg_map(tid, H);
    const int rotU = a.rotate ? (int)(((unsigned)b * 13u) % (unsigned)H) : 0;
    const int rotW = a.rotate ? (int)(((unsigned)b * 37u) % (unsigned)GH) : 0;
    // W_a f is t-invariant and every wave scores the same positions each step: keep this lane's Waf[p][k]
    // (WP positions x WK column chunks) in registers for the whole loop when it fits (P <= 64, H <= 256)
    constexpr int WP = 4, WK = 4;
    const bool waf_regs = (P <= WP * (NT / 64)) && (H <= WK * 64);
    float wreg[WP][WK];
#pragma unroll
    for (int i = 0; i < WP; ++i)
#pragma unroll
        for (int q = 0; q < WK; ++q) {
            const int p = wave + i * (NT / 64), k = lane + 64 * q;
            wreg[i][q] = (waf_regs && p < P && k < H) ? Waf_b[p * H + k] : 0.f;
        }
    __syncthreads();
    PDECL;

    for (int t = t0; t < t1; ++t) {
        const size_t bt = (size_t)b * T + t;
        // A: U_a h + b_Ua -> uah_s ; W_hh h + b_hh -> gh_s   (W_hh, U_a streamed from L2)
        matvec_rows(a.U_a, a.b_Ua, h_s, uah_s, H, H, a.vecW, tid, rotU);
        matvec_rows(a.W_hh, a.b_hh, h_s, gh_s, GH, H, a.vecW, tid, rotW);
        __syncthreads();
        PSTAMP(0);
        // B: e_p = v_a . tanh(Waf_p + uah) + b_va   (one wave per position, shuffle reduction)
        if (waf_regs) {
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const int p = wave + i * (NT / 64);
                if (p < P) {
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < WK; ++q) {
                        const int k = lane + 64 * q;
                        if (k < H) s += va_s[k] * caphn_tanh(wreg[i][q] + uah_s[k]);
                    }
                    s = wave_sum(s);
                    if (lane == 0) e_s[p] = s + bva;
                }
            }
        } else {
            for (int p = wave; p < P; p += NT / 64) {
                float s = 0.f;
                for (int k = lane; k < H; k += 64) s += va_s[k] * caphn_tanh(Waf_b[p * H + k] + uah_s[k]);
                s = wave_sum(s);
                if (lane == 0) e_s[p] = s + bva;
            }
        }
        __syncthreads();
        PSTAMP(1);
        // C: softmax over P (one wave)
        if (wave == 0) {
            float mx = -INFINITY;
            for (int p = lane; p < P; p += 64) mx = fmaxf(mx, e_s[p]);
            mx = wave_max(mx);
            float sum = 0.f;
            for (int p = lane; p < P; p += 64) { float ex = caphn_exp(e_s[p] - mx); e_s[p] = ex; sum += ex; }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            for (int p = lane; p < P; p += 64) { float al = e_s[p] * inv; e_s[p] = al; a.alphas[bt * P + p] = al; }
        }
        __syncthreads();
        PSTAMP(2);
        // D1: partial gi_ctx = sum_{p = g mod ng} alpha_p G_p over thread groups
        if (m.g >= 0) {
            for (int k = m.k; k < H; k += (m.ng == 1 ? NT : H)) {
                float acc[NG];
#pragma unroll
                for (int q = 0; q < NG; ++q) acc[q] = 0.f;
                for (int p = m.g; p < P; p += m.ng) {
                    const float al = e_s[p];
#pragma unroll
                    for (int q = 0; q < NG; ++q)
                        acc[q] += al * (q < RG ? G_s[p * RGH + q * H + k] : Gb[(size_t)p * GH + q * H + k]);
                }
#pragma unroll
                for (int q = 0; q < NG; ++q) part_s[m.g * GH + q * H + k] = acc[q];
            }
        }
        __syncthreads();
        PSTAMP(3);
        // D2: gates and h'
        for (int k = tid; k < H; k += NT) {
            float pre[NG];
            const float* xg = a.Xg + bt * GH;
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                float s = 0.f;
                for (int g = 0; g < m.ng; ++g) s += part_s[g * GH + q * H + k];
                pre[q] = xg[q * H + k] + s;
            }
            const float hp = h_s[k];
            float hnew;
            if (LSTM) {
                const float gi = caphn_sigmoid(pre[0] + gh_s[k]);
                const float gf = caphn_sigmoid(pre[1] + gh_s[H + k]);
                const float gg = caphn_tanh(pre[2] + gh_s[2 * H + k]);
                const float go = caphn_sigmoid(pre[3 % NG] + gh_s[(3 % NG) * H + k]);
                const float cp = c_s[k];
                const float cn = gf * cp + gi * gg;
                hnew = go * caphn_tanh(cn);
                a.gates[bt * GH + k] = gi; a.gates[bt * GH + H + k] = gf; a.gates[bt * GH + 2 * H + k] = gg;
                a.gates[bt * GH + (3 % NG) * H + k] = go;
                a.Cprev[bt * H + k] = cp; a.Cs[bt * H + k] = cn;
                c_s[k] = cn;
            } else {
                const float r = caphn_sigmoid(pre[0] + gh_s[k]);
                const float z = caphn_sigmoid(pre[1] + gh_s[H + k]);
                const float hnv = gh_s[2 * H + k];
                const float n = caphn_tanh(pre[2] + r * hnv);
                hnew = (1.0f - z) * n + z * hp;
                a.gates[bt * GH + k] = r; a.gates[bt * GH + H + k] = z; a.gates[bt * GH + 2 * H + k] = n;
                a.hn[bt * H + k] = hnv;
            }
            if (a.drop_p > 0.f) hnew *= caphn_keep_scale(a.drop_seed, (unsigned long long)bt * H + k, a.drop_p, 1.0f / (1.0f - a.drop_p));
            a.Hprev[bt * H + k] = hp;
            a.Hs[bt * H + k] = hnew;
            a.uah[bt * H + k] = uah_s[k];
            h_s[k] = hnew;                  // only this thread reads h_s[k] in D2
        }
        __syncthreads();
        PSTAMP(4);
    }
    PFLUSH;
}

// ------------------------------------------------------------------------------------------
// BPTT.  Walks t = T-1 .. 0 with dh (and dc) carried in LDS.  Resident G slabs serve
// d alpha_p = G_p . dgi.  Emits per-step dgi, dgh, d(U_a h), d e -- the weight gradients are
// batched MFMA GEMMs afterwards.
template <int CH>   // CH = 4: dwordx4 column chunks, CH = 1: scalar columns (H % 4 != 0)
__device__ __forceinline__ void matvec_t_accum(const float* __restrict__ W, const float* d_s, int rows, int H,
                                               int chunk, int slice, int nslices, float (&acc)[CH], int rot) {
    for (int j0 = slice; j0 < rows; j0 += nslices) {
        int j = j0 + rot;
        if (j >= rows) j -= rows;
        const float dj = d_s[j];
        const float* row = W + (size_t)j * H + chunk * CH;
        if (CH == 4) {
            f32x4 w = *reinterpret_cast<const f32x4*>(row);
            acc[0] += w[0] * dj; acc[1 % CH] += w[1] * dj; acc[2 % CH] += w[2] * dj; acc[3 % CH] += w[3] * dj;
        } else {
            acc[0] += row[0] * dj;
        }
    }
}

template <bool LSTM>
__global__ __launch_bounds__(NT) void rec_attn_bwd_kernel(RecBwdArgs a) {
    constexpr int NG = LSTM ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, H = a.H, GH = NG * a.H, T = a.T, RG = a.RG, RGH = a.RG * a.H;
    const int Ppad = (P + 63) & ~63;
    float* G_s = lds;
    float* dh_s = G_s + P * RGH;
    float* dc_s = dh_s + H;
    float* duah_s = dc_s + H;
    float* uah_s = duah_s + H;
    float* va_s = uah_s + H;
    float* dgh_s = va_s + H;
    float* dgi_s = dgh_s + GH;
    float* dal_s = dgi_s + GH;
    float* al_s = dal_s + Ppad;
    float* part_s = al_s + Ppad;     // [max(nslices, ng)][H]

    const float* Gb = a.G + (size_t)b * P * GH;
    load_G_resident(G_s, Gb, P, GH, RGH, a.vecS, tid);
    const int bt0 = a.t0, bt1 = a.t1 > 0 ? a.t1 : T;
    for (int k = tid; k < H; k += NT) {
        dh_s[k] = 0.f; va_s[k] = a.v_a[k];
        dc_s[k] = (LSTM && bt1 < T) ? a.dc0[(size_t)b * H + k] : 0.f;
    }
    const float* Waf_b = a.Waf + (size_t)b * P * H;
    // transposed mat-vec thread map: chunk of CH columns x row slice
    const int CH = a.vecW ? 4 : 1;
    const int nch = (H + CH - 1) / CH;
    const int nch_eff = min(nch, NT);
    const int nslices = NT / nch_eff;
    const KG m = kg_map(tid, H);
    const int rotU = a.rotate ? (int)(((unsigned)b * 13u) % (unsigned)H) : 0;
    const int rotW = a.rotate ? (int)(((unsigned)b * 37u) % (unsigned)GH) : 0;
    __syncthreads();
    PDECL;
    // fused attention parameter gradients: this thread owns column m.k and positions m.g, m.g + ng, ... (<= PGM of them)
    const bool fuse = a.dWaf != nullptr && m.g >= 0;
    float dw[PGM];
#pragma unroll
    for (int i = 0; i < PGM; ++i) dw[i] = 0.f;
    float dva = 0.f, dbva = 0.f;

    for (int t = bt1 - 1; t >= bt0; --t) {
        const size_t bt = (size_t)b * T + t;
        for (int p = tid; p < P; p += NT) al_s[p] = a.alphas[bt * P + p];
        // cell backward (pointwise), thread k
        for (int k = tid; k < H; k += NT) {
            float dh = dh_s[k] + a.dHs[bt * H + k];
            if (a.drop_p > 0.f) dh *= caphn_keep_scale(a.drop_seed, (unsigned long long)bt * H + k, a.drop_p, 1.0f / (1.0f - a.drop_p));
            uah_s[k] = a.uah[bt * H + k];
            if (LSTM) {
                const float gi = a.gates[bt * GH + k], gf = a.gates[bt * GH + H + k], gg = a.gates[bt * GH + 2 * H + k];
                const float go = a.gates[bt * GH + (3 % NG) * H + k];
                const float cp = a.Cprev[bt * H + k];
                const float tc = caphn_tanh(a.Cs[bt * H + k]);
                const float d_o = dh * tc;
                const float dc = dc_s[k] + dh * go * (1.0f - tc * tc);
                const float dip = dc * gg * gi * (1.0f - gi);
                const float dfp = dc * cp * gf * (1.0f - gf);
                const float dgp = dc * gi * (1.0f - gg * gg);
                const float dop = d_o * go * (1.0f - go);
                dc_s[k] = dc * gf;
                dh_s[k] = 0.f;                              // no direct h_{t-1} -> h_t path in an LSTM
                dgi_s[k] = dip; dgi_s[H + k] = dfp; dgi_s[2 * H + k] = dgp; dgi_s[(3 % NG) * H + k] = dop;
                dgh_s[k] = dip; dgh_s[H + k] = dfp; dgh_s[2 * H + k] = dgp; dgh_s[(3 % NG) * H + k] = dop;
                a.dgi[bt * GH + k] = dip; a.dgi[bt * GH + H + k] = dfp; a.dgi[bt * GH + 2 * H + k] = dgp;
                a.dgi[bt * GH + (3 % NG) * H + k] = dop;
            } else {
                const float r = a.gates[bt * GH + k], z = a.gates[bt * GH + H + k], n = a.gates[bt * GH + 2 * H + k];
                const float hnv = a.hn[bt * H + k], hp = a.Hprev[bt * H + k];
                const float dn = dh * (1.0f - z);
                const float dz = dh * (hp - n);
                const float dnp = dn * (1.0f - n * n);
                const float drp = dnp * hnv * r * (1.0f - r);
                const float dzp = dz * z * (1.0f - z);
                dh_s[k] = dh * z;                              // direct path h_{t-1} -> h_t
                dgi_s[k] = drp; dgi_s[H + k] = dzp; dgi_s[2 * H + k] = dnp;
                dgh_s[k] = drp; dgh_s[H + k] = dzp; dgh_s[2 * H + k] = dnp * r;
                a.dgi[bt * GH + k] = drp; a.dgi[bt * GH + H + k] = dzp; a.dgi[bt * GH + 2 * H + k] = dnp;
                a.dgh[bt * GH + k] = drp; a.dgh[bt * GH + H + k] = dzp; a.dgh[bt * GH + 2 * H + k] = dnp * r;
            }
        }
        __syncthreads();
        PSTAMP(0);
        // d alpha_p = G_p . dgi  (+ external gradient on the returned attention weights)
        for (int p = wave; p < P; p += NT / 64) {
            float s = 0.f;
            for (int j = lane; j < RGH; j += 64) s += G_s[p * RGH + j] * dgi_s[j];
            for (int j = RGH + lane; j < GH; j += 64) s += Gb[(size_t)p * GH + j] * dgi_s[j];
            s = wave_sum(s);
            if (lane == 0) dal_s[p] = s + (a.dalphas ? a.dalphas[bt * P + p] : 0.f);
        }
        __syncthreads();
        PSTAMP(1);
        // softmax backward: de_p = alpha_p (dalpha_p - sum_q alpha_q dalpha_q)
        if (wave == 0) {
            float dot = 0.f;
            for (int p = lane; p < P; p += 64) dot += al_s[p] * dal_s[p];
            dot = wave_sum(dot);
            for (int p = lane; p < P; p += 64) {
                const float de = al_s[p] * (dal_s[p] - dot);
                dal_s[p] = de;
                a.de[bt * P + p] = de;
            }
        }
        __syncthreads();
        PSTAMP(2);
        // d(U_a h)[k] = v_k sum_p de_p (1 - tanh^2(Waf_pk + uah_k)); p split over thread groups
        if (fuse) {
            // same sum, with dWaf[p][k] += de_p (1 - tanh^2) and d v_a[k] += de_p tanh accumulated over t in registers
            const int k = m.k;
            const float u = uah_s[k];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < PGM; ++i) {
                const int p = m.g + i * m.ng;
                if (p < P) {
                    const float de = dal_s[p];
                    const float tv = caphn_tanh(Waf_b[p * H + k] + u);
                    const float w = de * (1.0f - tv * tv);
                    s += w; dw[i] += w; dva += de * tv;
                    if (k == 0) dbva += de;
                }
                __builtin_amdgcn_sched_barrier(0);      // one position at a time: ten interleaved tanh chains spill
            }
            part_s[m.g * H + k] = s;
        } else if (m.g >= 0) {
            for (int k = m.k; k < H; k += (m.ng == 1 ? NT : H)) {
                const float u = uah_s[k];
                float s = 0.f;
                for (int p = m.g; p < P; p += m.ng) {
                    const float tv = caphn_tanh(Waf_b[p * H + k] + u);
                    s += dal_s[p] * (1.0f - tv * tv);
                }
                part_s[m.g * H + k] = s;
            }
        }
        __syncthreads();
        PSTAMP(3);
        for (int k = tid; k < H; k += NT) {
            float s = 0.f;
            for (int g = 0; g < m.ng; ++g) s += part_s[g * H + k];
            const float du = s * va_s[k];
            duah_s[k] = du;
            a.duah[bt * H + k] = du;
        }
        __syncthreads();
        PSTAMP(4);
        // dh_{t-1} += W_hh^T dgh + U_a^T duah
        {
            const int chunk = tid % nch_eff, slice = tid / nch_eff;
            if (slice < nslices) {
                for (int c = chunk; c < nch; c += nch_eff) {
                    if (CH == 4) {
                        float acc[4] = {0.f, 0.f, 0.f, 0.f};
                        matvec_t_accum<4>(a.W_hh, dgh_s, GH, H, c, slice, nslices, acc, rotW);
                        matvec_t_accum<4>(a.U_a, duah_s, H, H, c, slice, nslices, acc, rotU);
                        *reinterpret_cast<f32x4*>(part_s + slice * H + c * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                    } else {
                        float acc[1] = {0.f};
                        matvec_t_accum<1>(a.W_hh, dgh_s, GH, H, c, slice, nslices, acc, rotW);
                        matvec_t_accum<1>(a.U_a, duah_s, H, H, c, slice, nslices, acc, rotU);
                        part_s[slice * H + c] = acc[0];
                    }
                }
            }
        }
        __syncthreads();
        PSTAMP(5);
        for (int k = tid; k < H; k += NT) {
            float s = dh_s[k];
            for (int sl = 0; sl < nslices; ++sl) s += part_s[sl * H + k];
            dh_s[k] = s;
        }
        __syncthreads();
        PSTAMP(6);
    }
    PFLUSH;
    for (int k = tid; k < H; k += NT) {
        a.dh0[(size_t)b * H + k] = dh_s[k];
        if (LSTM) a.dc0[(size_t)b * H + k] = dc_s[k];
    }
    if (fuse) {
        const float vk = va_s[m.k];
#pragma unroll
        for (int i = 0; i < PGM; ++i) {
            const int p = m.g + i * m.ng;
            if (p < P) a.dWaf[((size_t)b * P + p) * H + m.k] = dw[i] * vk;
        }
        float* row = a.apart + ((size_t)b * m.ng + m.g) * (H + 1);
        row[m.k] = dva;
        if (m.k == 0) row[H] = dbva;
    }
}

// dWaf[b,p,k] = v_k sum_t de[b,t,p] (1 - tanh^2(Waf[b,p,k] + uah[b,t,k]))
// part[(b,pc)][k] = sum_{t, p in chunk} de tanh(.)   (-> d v_a) ; part[..][H] = sum de (-> d b_va)
// One workgroup per (b, chunk of `pchunk` positions); threads over k.  pchunk = 1 gives B*P workgroups.
__global__ void attn_param_grads_kernel(AttnGradArgs a) {
    const int b = blockIdx.x, pc = blockIdx.y;
    const int P = a.P, H = a.H, T = a.T;
    const int p0 = pc * a.pchunk, p1 = min(P, p0 + a.pchunk);
    const size_t blk = (size_t)b * gridDim.y + pc;
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        const float vk = a.v_a[k];
        float dv = 0.f;
        for (int p = p0; p < p1; ++p) {
            const float w = a.Waf[((size_t)b * P + p) * H + k];
            float s = 0.f;
#pragma unroll 4
            for (int t = 0; t < T; ++t) {
                const float de = a.de[((size_t)b * T + t) * P + p];
                const float tv = caphn_tanh(w + a.uah[((size_t)b * T + t) * H + k]);
                s += de * (1.0f - tv * tv);
                dv += de * tv;
            }
            a.dWaf[((size_t)b * P + p) * H + k] = s * vk;
        }
        a.part[blk * (H + 1) + k] = dv;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int p = p0; p < p1; ++p)
            for (int t = 0; t < T; ++t) s += a.de[((size_t)b * T + t) * P + p];
        a.part[blk * (H + 1) + H] = s;
    }
}

// dmean[b,k] = sum_j dh0[b,j] W_inith[j,k] (+ dc0 W_initc): the init_hidden backward.  As a 64x64-tile GEMM this M = B
// product is eight workgroups of seven K-slabs each -- 15-60 us on the post-BPTT chain; one workgroup per caption does it
// in a few microseconds.
__global__ __launch_bounds__(256) void dmean_kernel(int H, int F, const float* __restrict__ dh0, const float* __restrict__ Wh,
                                                    const float* __restrict__ dc0, const float* __restrict__ Wc, float* __restrict__ out) {
    extern __shared__ float dv[];       // [2][H]
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < H; j += 256) { dv[j] = dh0[(size_t)b * H + j]; dv[H + j] = dc0 ? dc0[(size_t)b * H + j] : 0.f; }
    __syncthreads();
    for (int k = threadIdx.x; k < F; k += 256) {
        float s0 = 0.f, s1 = 0.f;
        // eight rows' loads in flight (the plain loop waited for every load: H dependent L2 latencies on the post-BPTT chain)
        int j = 0;
        for (; j + 8 <= H; j += 8) {
            float w[8], c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { w[u] = Wh[(size_t)(j + u) * F + k]; c[u] = dc0 ? Wc[(size_t)(j + u) * F + k] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += dv[j + u] * w[u]; s1 += dv[H + j + u] * c[u]; }
        }
        for (; j < H; ++j) { s0 += dv[j] * Wh[(size_t)j * F + k]; if (dc0) s1 += dv[H + j] * Wc[(size_t)j * F + k]; }
        out[(size_t)b * F + k] = s0 + s1;
    }
}

// ctx[b,t,k] = sum_p alpha[b,t,p] f[b,p,k]
__global__ void ctx_kernel(int P, int F, int T, const float* __restrict__ alphas, const float* __restrict__ f, float* __restrict__ ctx,
                           int ldc) {
    const int b = blockIdx.x, t = blockIdx.y;
    const float* al = alphas + ((size_t)b * T + t) * P;
    for (int k = threadIdx.x; k < F; k += blockDim.x) {
        float s = 0.f;
        int p = 0;
        for (; p + 8 <= P; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = f[((size_t)b * P + p + u) * F + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += al[p + u] * v[u];
        }
        for (; p < P; ++p) s += al[p] * f[((size_t)b * P + p) * F + k];
        ctx[((size_t)b * T + t) * ldc + k] = s;
    }
}
// df[b,p,k] = sum_t alpha[b,t,p] dctx[b,t,k] + dmean[b,k] / P
__global__ void df_kernel(int P, int F, int T, const float* __restrict__ alphas, const float* __restrict__ dctx,
                          const float* __restrict__ dmean, const float* __restrict__ dmean2, float* __restrict__ df) {
    const int b = blockIdx.x, p = blockIdx.y;
    const float invP = 1.0f / (float)P;
    for (int k = threadIdx.x; k < F; k += blockDim.x) {
        float s = (dmean2 ? dmean[(size_t)b * F + k] + dmean2[(size_t)b * F + k] : dmean[(size_t)b * F + k]) * invP;
        int t = 0;
        for (; t + 8 <= T; t += 8) {
            float a[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u] = alphas[((size_t)b * T + t + u) * P + p]; v[u] = dctx[((size_t)b * T + t + u) * F + k]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += a[u] * v[u];
        }
        for (; t < T; ++t) s += alphas[((size_t)b * T + t) * P + p] * dctx[((size_t)b * T + t) * F + k];
        df[((size_t)b * P + p) * F + k] = s;
    }
}
// init_hidden (and init_c) in one launch per caption: mean over the P positions (4 position lanes per column, LDS
// reduction), then h0 = W_h mean + b_h (c0 likewise), one wave per output row.  Replaces mean_p_kernel + one or two
// M = B GEMMs of eight workgroups each (16 + 12 us on the front of the forward).
__global__ __launch_bounds__(1024) void init_state_kernel(int P, int F, int H, const float* __restrict__ f,
                                                          const float* __restrict__ Wh, const float* __restrict__ bh,
                                                          const float* __restrict__ Wc, const float* __restrict__ bc,
                                                          float* __restrict__ meanf, float* __restrict__ h0, float* __restrict__ c0) {
    extern __shared__ float sm[];           // [4][Fp] partial sums, then mean in sm[0..F)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Fp = (F + 255) & ~255;
    for (int k0 = 0; k0 < F; k0 += 256) {
        const int k = k0 + (tid & 255), pl = tid >> 8;
        float s = 0.f;
        if (k < F) for (int p = pl; p < P; p += 4) s += f[((size_t)b * P + p) * F + k];
        sm[pl * Fp + k0 + (tid & 255)] = s;
    }
    __syncthreads();
    const float invP = 1.0f / (float)P;
    for (int k = tid; k < F; k += 1024) {
        const float m = ((sm[k] + sm[Fp + k]) + (sm[2 * Fp + k] + sm[3 * Fp + k])) * invP;
        sm[4 * Fp + k] = m;
        meanf[(size_t)b * F + k] = m;
    }
    __syncthreads();
    const float* mean = sm + 4 * Fp;
    for (int j = wave; j < H; j += 16) {
        float sh = 0.f, sc = 0.f;
        for (int k = lane; k < F; k += 64) {
            const float m = mean[k];
            sh += Wh[(size_t)j * F + k] * m;
            if (Wc) sc += Wc[(size_t)j * F + k] * m;
        }
        sh = wave_sum(sh);
        if (Wc) sc = wave_sum(sc);
        if (lane == 0) {
            h0[(size_t)b * H + j] = sh + bh[j];
            if (Wc) c0[(size_t)b * H + j] = sc + bc[j];
        }
    }
}
// mean over positions
__global__ void mean_p_kernel(int P, int F, const float* __restrict__ f, float* __restrict__ out) {
    const int b = blockIdx.x;
    const float invP = 1.0f / (float)P;
    for (int k = threadIdx.x; k < F; k += blockDim.x) {
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += f[((size_t)b * P + p) * F + k];
        out[(size_t)b * F + k] = s * invP;
    }
}

constexpr size_t LDS_LIMIT = 160 * 1024;
bool g_attr_set[64];              // per device: the 160 KB dynamic-LDS opt-in is a per-device function attribute
int ensure_lds_attr() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return CAPHN_ELAUNCH;
    if (g_attr_set[dev]) return CAPHN_OK;
    const void* fns[4] = {reinterpret_cast<const void*>(rec_attn_fwd_kernel<false>), reinterpret_cast<const void*>(rec_attn_fwd_kernel<true>),
                          reinterpret_cast<const void*>(rec_attn_bwd_kernel<false>), reinterpret_cast<const void*>(rec_attn_bwd_kernel<true>)};
    for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT) != hipSuccess) return CAPHN_ELAUNCH;
    g_attr_set[dev] = true;
    return CAPHN_OK;
}

}  // namespace

static int kg_groups(int H) { return H >= NT ? 1 : NT / H; }
size_t caphn_rec_fwd_lds_bytes(int P, int H, int NG, int RG) {
    const size_t Ppad = (P + 63) & ~63;
    return sizeof(float) * ((size_t)RG * P * H + 4 * (size_t)H + (size_t)NG * H + Ppad + (size_t)kg_groups(H) * NG * H);
}
size_t caphn_rec_bwd_lds_bytes(int P, int H, int NG, int RG) {
    const size_t Ppad = (P + 63) & ~63;
    const int CH = (H % 4 == 0) ? 4 : 1;
    const int nch = (H + CH - 1) / CH;
    const int nch_eff = nch < NT ? nch : NT;
    int nsl = NT / nch_eff;
    if (kg_groups(H) > nsl) nsl = kg_groups(H);
    // the scalar-column map (CH = 1) may be chosen at run time for unaligned weights
    const int nsl1 = NT / (H < NT ? H : NT);
    if (nsl1 > nsl) nsl = nsl1;
    return sizeof(float) * ((size_t)RG * P * H + 5 * (size_t)H + 2 * (size_t)NG * H + 2 * Ppad + (size_t)nsl * H);
}
int caphn_rec_resident_gates(int P, int H, int NG) {
    for (int rg = NG; rg >= 0; --rg)
        if (caphn_rec_fwd_lds_bytes(P, H, NG, rg) <= LDS_LIMIT && caphn_rec_bwd_lds_bytes(P, H, NG, rg) <= LDS_LIMIT) return rg;
    return -1;
}

int caphn_launch_rec_fwd(const RecFwdArgs& a, bool lstm, hipStream_t s) {
    const size_t lds = caphn_rec_fwd_lds_bytes(a.P, a.H, lstm ? 4 : 3, a.RG);
    if (lds > LDS_LIMIT) return CAPHN_ELIMIT;
    int rc = ensure_lds_attr(); if (rc) return rc;
    if (lstm) hipLaunchKernelGGL(rec_attn_fwd_kernel<true>, dim3(a.B), dim3(NT), lds, s, a);
    else hipLaunchKernelGGL(rec_attn_fwd_kernel<false>, dim3(a.B), dim3(NT), lds, s, a);
    return caphn_launch_status();
}
int caphn_rec_bwd_groups(int P, int H) {
    if (H > NT) return 0;
    const int ng = NT / H;
    return (P + ng - 1) / ng <= PGM ? ng : 0;
}
int caphn_launch_rec_bwd(const RecBwdArgs& a, bool lstm, hipStream_t s) {
    if (a.dWaf && (!a.apart || caphn_rec_bwd_groups(a.P, a.H) == 0)) return CAPHN_EINVAL;
    const size_t lds = caphn_rec_bwd_lds_bytes(a.P, a.H, lstm ? 4 : 3, a.RG);
    if (lds > LDS_LIMIT) return CAPHN_ELIMIT;
    int rc = ensure_lds_attr(); if (rc) return rc;
    if (lstm) hipLaunchKernelGGL(rec_attn_bwd_kernel<true>, dim3(a.B), dim3(NT), lds, s, a);
    else hipLaunchKernelGGL(rec_attn_bwd_kernel<false>, dim3(a.B), dim3(NT), lds, s, a);
    return caphn_launch_status();
}
int caphn_launch_attn_param_grads(const AttnGradArgs& a, int B, int npc, hipStream_t s) {
    hipLaunchKernelGGL(attn_param_grads_kernel, dim3(B, npc), dim3(a.H >= 192 ? 256 : 64), 0, s, a);
    return caphn_launch_status();
}
int caphn_launch_dmean(int B, int H, int F, const float* dh0, const float* Wh, const float* dc0, const float* Wc, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dmean_kernel, dim3(B), dim3(256), sizeof(float) * 2 * H, s, H, F, dh0, Wh, dc0, Wc, out);
    return caphn_launch_status();
}
int caphn_launch_ctx(int B, int T, int P, int F, const float* alphas, const float* f, float* ctx, int ldc, hipStream_t s) {
    hipLaunchKernelGGL(ctx_kernel, dim3(B, T), dim3(F >= 192 ? 256 : 64), 0, s, P, F, T, alphas, f, ctx, ldc);
    return caphn_launch_status();
}
int caphn_launch_df(int B, int T, int P, int F, const float* alphas, const float* dctx, const float* dmean, float* df, hipStream_t s,
                    const float* dmean2) {
    hipLaunchKernelGGL(df_kernel, dim3(B, P), dim3(F >= 192 ? 256 : 64), 0, s, P, F, T, alphas, dctx, dmean, dmean2, df);
    return caphn_launch_status();
}
int caphn_launch_init_state(int B, int P, int F, int H, const float* f, const float* Wh, const float* bh, const float* Wc,
                            const float* bc, float* meanf, float* h0, float* c0, hipStream_t s) {
    const int Fp = (F + 255) & ~255;
    hipLaunchKernelGGL(init_state_kernel, dim3(B), dim3(1024), sizeof(float) * (4 * Fp + F), s, P, F, H, f, Wh, bh, Wc, bc, meanf, h0, c0);
    return caphn_launch_status();
}
int caphn_launch_mean_p(int B, int P, int F, const float* f, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mean_p_kernel, dim3(B), dim3(F >= 192 ? 256 : 64), 0, s, P, F, f, out);
    return caphn_launch_status();
}
