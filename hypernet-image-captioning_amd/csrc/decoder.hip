// Decoder composites: AttentionGru.forward with teacher forcing (models/decoderlstm.py:49-120)
// and its backward, as a fixed sequence of hand-written kernels on one stream.
//
//   forward : feature_fc (2 MFMA GEMMs, ReLU fused) -> mean/init_h -> W_a f (GEMM, hoisted out of the
//             loop; the reference recomputes it every step, models/attention.py:34) -> G = f W_ih[:,E:]^T
//             -> embedding gather (+ the x_0 = x_1 = 0 quirk) -> x-side gate GEMM for all T at once
//             -> persistent recurrent kernel -> vocab projection GEMM.
//   backward: vocab GEMMs (dW, dh) -> persistent BPTT kernel -> batched weight-gradient GEMMs
//             (K = B*T or B*P, split-K) -> attention / feature_fc chain.
#include "common.h"
#include "decoder_internal.h"
#include "gemm_internal.h"
#include <algorithm>
#include <initializer_list>
#include <mutex>

int g_tune_deterministic = 0;   // 1: bit-reproducible gradients -- no split-K (fp32 atomics), embedding gradient by a destination-major scan
int g_tune_rec_pair = 1;     // 1 (default): two workgroups per caption in the recurrent kernels (recurrent_pair.hip)
int g_tune_rec_rotate = 1;
int g_tune_vocab_order = 1;         // fork mode 2: 0 vocabulary gradients start beside the dHs GEMM, 1 after it (beside BPTT only)
int g_tune_branch_mask = 7;         // side branches of the composites in use (bit i: branch i; a cleared branch runs on the caller's stream)
int g_tune_splitk_target = 1280;   // workgroups a split-K GEMM of the composites aims at (caphn_tune 17): five 64x64 workgroups per CU.
                                   // Same-box A/B: 1024 1.907/1.911, 1152 1.909/1.904, 1280 1.897/1.901 ms (256, 512, 2048: slower)
int g_tune_chain_main = 0;  // 1: with the hypernet VJP hooked in, the chain to it runs on the caller's stream (see "after BPTT");
                            // measured 30 us per step WORSE than 0 (2.026/2.023 vs 1.996/1.994 ms, same box, alternating)
int g_tune_hops = 0;        // 1: cross-stream dependencies of the composites as stream memory operations (struct Side) -- 4 us per hop in isolation
                            // (tools/native/hop_latency.hip) but the STEP went from 1.68 to 2.02 ms with them (same box, alternating;
                            // profiles/r03_hops_by_value_ab.txt): off.  caphn_tune key 26
int g_tune_fork = 4;        // 0: one stream; 1: independent branches on side streams, the vocabulary weight gradient (dW_fc) starting
                            // after BPTT; 2: dW_fc beside BPTT; 3: as 1, the big leaves held back until df exists; 4 (default): 2 when
                            // the pair kernels run BPTT (512-thread workgroups leave wave slots, registers and 60 KB of LDS per CU
                            // for a GEMM workgroup: 1.987/2.004 -> 1.956/1.958 ms per step), else 1 (a 1024-thread BPTT workgroup
                            // and the GEMM fight for the same CUs: 1-2 % worse, round 1)
int g_tune_join_chain = 0;   // caphn_tune key 35: 1 = the backward's branches 0 and 2 end into branch 1, which alone joins the caller's stream (measured +2..+3 us: off)
int g_tune_sk_dhs = 0, g_tune_sk_vocab_w = 0;      // experiments (caphn_tune keys 33 / 34): split-K of the two live-row vocabulary GEMMs of the backward, 0 = automatic
#define RUN(x) do { int _rc = (x); if (_rc != CAPHN_OK) return _rc; } while (0)

namespace {

// Side streams for the independent branches of one composite call: ONE set per device (include/caphn.h, "Hidden
// state"), created on that device's first composite call (outside any graph capture); fork/join is expressed with
// events, so a capturing caller stream captures the branches too.  Creation is serialised by a mutex; USE is not
// thread-safe: one host thread drives one device (as the reference's training loop does).
struct Side {
    hipStream_t st[3];
    // A dependency slot: "everything enqueued on `from` so far" -> a later wait on another stream.  Two forms.  Events
    // (hipEventRecord + hipStreamWaitEvent): capturable into a graph, but the waiting queue is released ~10 us after the producer
    // ends on an idle chip and 20-27 us inside the step (tools/native/hop_latency.hip; the kernel trace shows it after every fork
    // whose producer has only just finished).  Stream memory operations (hipStreamWriteValue32 of a sequence number +
    // hipStreamWaitValue32 >= that number, on a word of device memory): 4 us idle -- the command processor polls the word itself --
    // but not capturable.  begin() picks per composite call: memory operations unless the caller's stream is capturing.
    enum { D_FORK = 0, D_JOIN = 1, D_X = 4, D_MS = 10, D_PRE_F = 14, D_PRE_ALL = 15, D_COUNT = 16 };
    struct Dep { hipEvent_t ev; unsigned seq = 0; bool by_value = false; };
    Dep deps[D_COUNT];
    unsigned* vals = nullptr;          // D_COUNT words of device memory (zero at start); slot values only grow
    unsigned counter = 0;
    bool values = false;               // form chosen by the current composite call
    bool pre_valid = false;
    const void* pre_ws = nullptr;      // workspace of that precompute: a forward that waits on pre_f / pre_all itself (bit 16) must own it
    bool ready = false, on = false;
    bool forked[3] = {false, false, false};
    hipStream_t main = nullptr;
    int init() {
        if (ready) return CAPHN_OK;
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (ready) return CAPHN_OK;
        for (auto& s : st) if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return CAPHN_ELAUNCH;
        for (auto& d : deps) if (hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) != hipSuccess) return CAPHN_ELAUNCH;
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(unsigned) * D_COUNT) == hipSuccess && hipMemset(p, 0, sizeof(unsigned) * D_COUNT) == hipSuccess)
            vals = static_cast<unsigned*>(p);
        else (void)hipGetLastError();          // no word memory: events only
        ready = true;
        return CAPHN_OK;
    }
    int post(int slot, hipStream_t from) {
        Dep& d = deps[slot];
        if (values) {
            d.seq = ++counter; d.by_value = true;
            return hipStreamWriteValue32(from, vals + slot, d.seq, 0) == hipSuccess ? CAPHN_OK : CAPHN_ELAUNCH;
        }
        d.by_value = false;
        return hipEventRecord(d.ev, from) == hipSuccess ? CAPHN_OK : CAPHN_ELAUNCH;
    }
    int await(int slot, hipStream_t to) {
        Dep& d = deps[slot];
        if (d.by_value)
            return hipStreamWaitValue32(to, vals + slot, d.seq, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess ? CAPHN_OK : CAPHN_ELAUNCH;
        return hipStreamWaitEvent(to, d.ev, 0) == hipSuccess ? CAPHN_OK : CAPHN_ELAUNCH;
    }
    // milestone k is reached once everything enqueued on `from` so far has run
    int milestone(int k, hipStream_t from) {
        if (init() != CAPHN_OK) return CAPHN_ELAUNCH;
        return post(D_MS + k, from);
    }
    int begin(hipStream_t m, bool enable) {
        main = m; on = enable;
        if (init() != CAPHN_OK) return CAPHN_ELAUNCH;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        values = g_tune_hops != 0 && vals != nullptr && hipStreamIsCapturing(m, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
        return CAPHN_OK;
    }
    hipStream_t s(int i) const { return (on && ((g_tune_branch_mask >> i) & 1)) ? st[i] : main; }
    // branch i starts after everything enqueued on main so far
    int forkto(int i) {
        if (!on || !((g_tune_branch_mask >> i) & 1)) return CAPHN_OK;
        if (post(D_FORK, main) != CAPHN_OK) return CAPHN_ELAUNCH;
        if (await(D_FORK, st[i]) != CAPHN_OK) return CAPHN_ELAUNCH;
        forked[i] = true;
        return CAPHN_OK;
    }
    // several branches start at the same point of main: ONE record (a record costs the recording stream ~5-7 us of
    // queue time -- three of them sat between the BPTT kernel and the first kernel of the chain behind it)
    int fork_many(std::initializer_list<int> ids) {
        if (!on) return CAPHN_OK;
        if (post(D_FORK, main) != CAPHN_OK) return CAPHN_ELAUNCH;
        for (int i : ids) {
            if (!((g_tune_branch_mask >> i) & 1)) continue;       // branch folded into the caller's stream
            if (await(D_FORK, st[i]) != CAPHN_OK) return CAPHN_ELAUNCH;
            forked[i] = true;
        }
        return CAPHN_OK;
    }
    // main continues only after branch i's work so far
    int jointo(int i) {
        if (!on || !((g_tune_branch_mask >> i) & 1)) return CAPHN_OK;
        forked[i] = false;
        if (post(D_JOIN + i, st[i]) != CAPHN_OK) return CAPHN_ELAUNCH;
        return await(D_JOIN + i, main);
    }
    // branch i ends INTO branch j (j then carries both to main: one wait on the caller's stream instead of two -- every event
    // operation on that stream is a ~7 us packet between two kernels that would otherwise follow each other without a gap)
    int join_into(int i, int j) {
        if (!on || !((g_tune_branch_mask >> i) & 1)) return CAPHN_OK;
        if (!((g_tune_branch_mask >> j) & 1)) return jointo(i);        // j is folded into the caller's stream
        forked[i] = false;
        if (post(D_JOIN + i, st[i]) != CAPHN_OK) return CAPHN_ELAUNCH;
        return await(D_JOIN + i, st[j]);
    }
    // A composite that returns early (a failed launch) must not leave a branch forked: inside a stream capture an
    // unjoined branch makes hipStreamEndCapture fail (hipErrorStreamCaptureUnjoined) -- every composite holds a Scope.
    struct Scope {
        Side& s;
        explicit Scope(Side& side) : s(side) {}
        ~Scope() { for (int i = 0; i < 3; ++i) if (s.on && s.forked[i]) (void)s.jointo(i); }
    };
    // cross-branch dependency in two halves: record slot k at the producer's current tail, wait later on the consumer
    // (a record remembers its stream: a wait on that same stream is dropped -- stream order already gives it)
    hipStream_t xfrom[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int record(int k, hipStream_t from) {
        if (!on) return CAPHN_OK;
        xfrom[k] = from;
        return post(D_X + k, from);
    }
    int wait(int k, hipStream_t to) {
        if (!on || xfrom[k] == to) return CAPHN_OK;
        return await(D_X + k, to);
    }
    // `to` waits for what `from` has enqueued so far (cross-branch dependency), via slot k
    int dep(hipStream_t from, hipStream_t to, int k) {
        if (!on || from == to) return CAPHN_OK;
        if (post(D_X + k, from) != CAPHN_OK) return CAPHN_ELAUNCH;
        return await(D_X + k, to);
    }
};
constexpr int MAX_DEVICES = 64;
Side g_sides[MAX_DEVICES];
// the current device's set (HIP's current device is per host thread; torch sets it before calling in)
inline Side* side_here() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    return &g_sides[dev];
}

struct Ws {   // float offsets into the workspace
    size_t Y1, f, meanf, h0, c0, Waf, G, Xe, Xg, Hs, Hprev, gates, hn, Cs, Cprev, uah, alphas, idx;
    size_t dHs, dgi, dgh, duah, de, dh0, dc0, ctx, dctx, dXe, dWaf, dmeanf, df, dY1, apart, vtmp, colws, colws_s[3], prof, rowmap;
    size_t xch, xch_floats;     // exchange areas of the pair kernels: forward, then backward (xch_floats each)
    size_t wp;                  // packed recurrent weights of the pair kernels
    // AttentionGru(num_layers > 1): attention cell output / its gradient [B,T,H], layered initial state, dh carry, GEMM temporaries,
    // and per extra layer the saved slots (decoder_layers.hip)
    size_t H0s, dH0, h0L, dcarry, tgi, tgh, tmpA, tmpB;
    size_t Lin[CAPHN_MAX_DEC_LAYERS - 1], Lgates[CAPHN_MAX_DEC_LAYERS - 1], Lhn[CAPHN_MAX_DEC_LAYERS - 1],
           Ldgi[CAPHN_MAX_DEC_LAYERS - 1], Ldgh[CAPHN_MAX_DEC_LAYERS - 1];
    int L1;                     // number of extra layers
    size_t total;
    int npc, pchunk, NG, arows;
};
// the recurrent kernels run two workgroups per caption when the shape allows it (teacher-forced launches only)
inline bool use_pair(const caphn_decoder_dims* d) {
    const int NG = d->cell == CAPHN_CELL_LSTM ? 4 : 3;
    return g_tune_rec_pair && d->T > 1 && d->layers <= 1 && caphn_rec_pair_resident_gates(d->P, d->H, NG) >= 0;
}

inline size_t up4(size_t v) { return (v + 3) & ~(size_t)3; }

inline Ws layout(const caphn_decoder_dims* d) {
    Ws w;
    const size_t B = d->B, T = d->T, P = d->P, F = d->F, E = d->E, H = d->H, V = d->V, D = d->D;
    const bool lstm = d->cell == CAPHN_CELL_LSTM, raw = d->raw_features != 0;
    const size_t NG = lstm ? 4 : 3;
    w.NG = (int)NG;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += up4(n); return r; };
    w.Y1 = take(raw ? 0 : B * P * F); w.f = take(raw ? 0 : B * P * F); w.meanf = take(B * F); w.h0 = take(B * H);
    w.c0 = take(lstm ? B * H : 0);
    w.Waf = take(B * P * H); w.G = take(B * P * NG * H); w.Xe = take(B * T * (E + F)); w.Xg = take(B * T * NG * H);     // Xc = [Xe | ctx] rows of E + F: the cell's input, one operand for dW_ih
    w.Hs = take(B * T * H); w.Hprev = take(B * T * H); w.gates = take(B * T * NG * H); w.hn = take(lstm ? 0 : B * T * H);
    w.Cs = take(lstm ? B * T * H : 0); w.Cprev = take(lstm ? B * T * H : 0);
    w.uah = take(B * T * H); w.alphas = take(B * T * P); w.idx = take(2 * B * T);   // int64
    w.dHs = take(B * T * H); w.dgi = take(B * T * NG * H); w.dgh = take(lstm ? 0 : B * T * NG * H); w.duah = take(B * T * H);
    w.de = take(B * T * P); w.dh0 = take(B * H); w.dc0 = take(lstm ? B * H : 0);
    w.ctx = w.Xe + E; w.dctx = take(raw ? 0 : B * T * F);
    w.dXe = take(B * T * E); w.dWaf = take(B * P * H); w.dmeanf = take(raw ? 0 : 2 * B * F); w.df = take(raw ? 0 : B * P * F);      // dmeanf: two partials (pair BPTT epilogue)
    w.dY1 = take(raw ? 0 : B * P * F);
    w.pchunk = 1; w.npc = (int)((P + w.pchunk - 1) / w.pchunk);     // one workgroup per (caption, position)
    // per caption: positions, or thread groups when the BPTT kernel fuses the attention parameter gradients
    const size_t arows = std::max<size_t>(std::max<size_t>(w.npc, (size_t)caphn_rec_bwd_groups((int)P, (int)H)),
                                          (size_t)caphn_rec_pair_bwd_groups((int)P, (int)H));
    w.arows = (int)arows;
    w.apart = take(B * arows * (H + 1)); w.vtmp = take(H + 1);
    size_t cs = 0;
    auto need = [&](size_t M, size_t N) { cs = std::max(cs, caphn_colsum_workspace_bytes((int)M, (int)N) / sizeof(float)); };
    need(B * T, V); need(B * T, NG * H); need(B * T, H); need(B * P, H); need(B * P, F); need(B, H);
    need(B * arows, H + 1);
    if (d->layers > 1) need(B * (T + 1), 3 * H);
    w.colws = take(cs);
    {   // side-stream branches run their own (small) column sums concurrently
        size_t c2 = 0;
        auto need2 = [&](size_t M, size_t N) { c2 = std::max(c2, caphn_colsum_workspace_bytes((int)M, (int)N) / sizeof(float)); };
        need2(B * T, NG * H); need2(B * T, H); need2(B * P, H); need2(B * P, F); need2(B, H); need2(B * arows, H + 1);
        if (d->layers > 1) need2(B * (T + 1), 3 * H);
        for (int i = 0; i < 3; ++i) w.colws_s[i] = take(c2);
    }
    w.prof = take(64);        // 2 x 8 uint64 phase counters (forward, backward) of the recurrent kernels
    w.rowmap = take(B * T + 4);   // int: [0] = number of valid rows, [4..] = their physical (b*T+t) indices
    w.xch_floats = caphn_rec_pair_xch_bytes((int)B, (int)P, (int)H) / sizeof(float);
    w.xch = take(2 * w.xch_floats);
    o = (o + 31) & ~(size_t)31;                                  // 128-byte aligned
    w.wp = take(caphn_rec_pair_wp_floats((int)H, (int)NG));     // [U_a; W_hh] at the aligned row pitch
    w.L1 = d->layers > 1 ? d->layers - 1 : 0;
    {
        const size_t on = w.L1 > 0 ? 1 : 0, S = T + 1;
        w.H0s = take(on * B * T * H); w.dH0 = take(on * B * T * H); w.h0L = take(on * B * H); w.dcarry = take(on * B * H);
        w.tgi = take(on * B * 3 * H); w.tgh = take(on * B * 3 * H); w.tmpA = take(on * B * H); w.tmpB = take(on * B * H);
        for (int l = 0; l < CAPHN_MAX_DEC_LAYERS - 1; ++l) {
            const size_t u = l < w.L1 ? 1 : 0;
            w.Lin[l] = take(u * B * S * H); w.Lgates[l] = take(u * B * S * 3 * H); w.Lhn[l] = take(u * B * S * H);
            w.Ldgi[l] = take(u * B * S * 3 * H); w.Ldgh[l] = take(u * B * S * 3 * H);
        }
    }
    w.total = o;
    (void)D;
    return w;
}

// stable compaction of the rows whose target is not ignored: map[0] = count, map[4 + j] = j-th valid row
__global__ __launch_bounds__(1024) void valid_rows_kernel(int n, const int64_t* __restrict__ tgt, int64_t ignore, int* __restrict__ map) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + tid;
        const int v = (i < n && tgt[i] != ignore) ? 1 : 0;
        const unsigned long long ball = __ballot(v);
        const int before = __popcll(ball & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(ball);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int q = 0; q < 16; ++q) { if (q < wave) woff += wsum[q]; tot += wsum[q]; }
        const int base = base_s;
        if (v) map[4 + base + woff + before] = i;
        __syncthreads();
        if (tid == 0) base_s = base + tot;
        __syncthreads();
    }
    if (tid == 0) map[0] = base_s;
}

// src[0..n) -> a, src[n] -> b
__global__ void copy2_kernel(const float* __restrict__ src, int n, float* __restrict__ a, float* __restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = src[i];
    else if (i == n) b[0] = src[n];
}

// the teacher-forced input rows in one launch: idx[i] (kept for the backward's scatter) and Xe row i = embed[idx[i]] or zeros
// x_0 = x_1 = 0 (in-place zero of a view, decoderlstm.py:82-84), x_t = embed[caps[:,t-1]] for t >= 2
__global__ __launch_bounds__(256) void teacher_inputs_kernel(int T, int E, const int64_t* __restrict__ caps, const float* __restrict__ table,
                                                            int64_t* __restrict__ idx, float* __restrict__ out, int ostride) {
    const int i = blockIdx.x, t = i % T;
    const int64_t id = t < 2 ? (int64_t)-1 : caps[i - 1];
    if (threadIdx.x == 0) idx[i] = id;
    for (int e = threadIdx.x; e < E; e += 256) out[(size_t)i * ostride + e] = id < 0 ? 0.f : table[(size_t)id * E + e];
}

inline bool dims_ok(const caphn_decoder_dims* d) {
    if (d && !(d->dropout_p >= 0.f && d->dropout_p < 1.f)) return false;
    if (!(d && d->B > 0 && d->T > 0 && d->P > 0 && d->D > 0 && d->F > 0 && d->E > 0 && d->H > 0 && d->V > 0)) return false;
    if (d->cell != CAPHN_CELL_GRU && d->cell != CAPHN_CELL_LSTM) return false;
    if (d->raw_features && d->F != d->D) return false;
    if (d->layers < 0 || d->layers > CAPHN_MAX_DEC_LAYERS) return false;
    if (d->layers > 1 && d->cell != CAPHN_CELL_GRU) return false;      // the reference's AttentionLstm has no extra layers
    if (d->logits_ld != 0 && d->logits_ld < d->V) return false;
    return true;
}
inline bool layer_params_ok(const caphn_decoder_dims* d, const caphn_decoder_params* p) {
    for (int l = 0; l + 1 < d->layers; ++l)
        if (!p->lw_ih[l] || !p->lw_hh[l] || !p->lb_ih[l] || !p->lb_hh[l]) return false;
    return true;
}

// split-K heuristic for the weight-gradient GEMMs (small MxN, long K): aim at >= 4 workgroups per CU
// with at least 8 K-slabs each (the kernel picks 64x64 tiles below 1024 128x128 tiles)
inline int pick_splitk(int M, int N, int K) {
    if (g_tune_deterministic) return 1;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (t128 >= 1024) return 1;
    const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    const int nslab = (K + 31) / 32;
    long s = (g_tune_splitk_target + t64 - 1) / t64;
    if (s > nslab / 8) s = nslab / 8;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}


// C = A^T-or-not . B with optional split-K (zero-fills C first when splitting)
inline int gemm_auto(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int flags, hipStream_t s,
                     const int* rowmap = nullptr, int map_mode = 0, bool prezeroed = false) {
    int sk = pick_splitk(M, N, K);
    if (rowmap) {      // row subset: rowmap[0] = count (device), rowmap + 4 = indices
        if (sk > 1 && !prezeroed) {
            if (ldc == N) { RUN(caphn_zero_f32(C, (size_t)M * N, s)); }
            else if (hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N, M, s) != hipSuccess) return CAPHN_ELAUNCH;
        }
        return caphn_gemm_mapped(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, sk, rowmap + 4, rowmap, map_mode, s);
    }
    if (sk > 1 && (flags & ~CAPHN_GEMM_BIAS) == 0) {
        if (prezeroed) {
        } else if (ldc == N) {
            RUN(caphn_zero_f32(C, (size_t)M * N, s));
        } else {
            if (hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N, M, s) != hipSuccess) return CAPHN_ELAUNCH;
        }
        return caphn_gemm_f32(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, 0, flags, sk, s);
    }
    return caphn_gemm_f32(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, 0, flags, 1, s);
}

// weight gradient dW = A^T B (A stored [K, M]) with its bias gradient db = column sums of A fused into the GEMM
inline int wgrad_bias(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* dW, int ldc, float* db,
                      const int* rowmap, void* cws, hipStream_t s, bool prezeroed) {
    int sk = pick_splitk(M, N, K);
    if (rowmap && g_tune_sk_vocab_w > 0 && !g_tune_deterministic) sk = g_tune_sk_vocab_w;      // (experiment: caphn_tune key 34)
    return caphn_gemm_tn_colsum(M, N, K, A, lda, B, ldb, dW, ldc, db, sk, rowmap, cws, prezeroed, s);
}

// feature_fc, init_hidden (+ init_c), W_a f, G = f W_ih[:,E:]^T  -- everything that does not depend on the captions.
// part 1 (theta-independent: feature_fc, init_hidden, W_a f) can be issued ahead of time by caphn_decoder_precompute
// (dims.precomputed = 1 then skips it here); part 2 is the G GEMM, which needs the generated W_ih.
static int decoder_precompute(const caphn_decoder_dims* d, const caphn_decoder_params* p, const Ws& w, float* ws,
                              const float* features, const float** f_out, hipStream_t s, int parts = 3, bool mark = false) {
    const int B = d->B, P = d->P, D = d->D, F = d->F, E = d->E, H = d->H;
    const bool lstm = d->cell == CAPHN_CELL_LSTM, raw = d->raw_features != 0;
    const int BP = B * P, GH = w.NG * H, EF = E + F;
    const float* f = raw ? features : ws + w.f;
    *f_out = f;
    if ((parts & 1) && !raw) {
        // feature_fc: Linear(D,F) + ReLU + Linear(F,F)      decoderlstm.py:22-26,61
        RUN(caphn_gemm_f32(0, 1, BP, F, D, features, D, p->fc0_w, D, ws + w.Y1, F, p->fc0_b, nullptr, 0,
                           CAPHN_GEMM_BIAS | CAPHN_GEMM_RELU, 1, s));
        RUN(caphn_gemm_f32(0, 1, BP, F, F, ws + w.Y1, F, p->fc2_w, F, ws + w.f, F, p->fc2_b, nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
    }
    Side* sdp = side_here();
    if (!sdp) return CAPHN_ELAUNCH;
    Side& sd = *sdp;
    // (mark: the call comes from caphn_decoder_precompute, beside the optimiser on the caller's side stream -- init_hidden and W_a f
    //  then run behind feature_fc on that stream: a fork and a join are two event hops of ~25 us each, more than the 27 + 16 us the
    //  two kernels take in a row, and the next recurrent kernel waits for the later of them)
    RUN(sd.begin(s, g_tune_fork != 0 && (parts & 1) && !mark));
    Side::Scope scope(sd);
    if (mark) RUN(sd.post(Side::D_PRE_F, s));      // f exists: a later forward's G GEMM may go
    if (parts & 1) {
        RUN(sd.fork_many({0, 1}));
        // branch 0 -- init_hidden: mean over positions -> Linear(F,H) (and init_c for the LSTM)      :122-135 / :255-260
        RUN(caphn_launch_init_state(B, P, F, H, f, p->inith_w, p->inith_b, lstm ? p->initc_w : nullptr, lstm ? p->initc_b : nullptr,
                                    ws + w.meanf, ws + w.h0, lstm ? ws + w.c0 : nullptr, sd.s(0)));
        // branch 1 -- t-invariant attention projection W_a f + b      attention.py:34
        RUN(caphn_gemm_f32(0, 1, BP, H, F, f, F, p->Wa_w, F, ws + w.Waf, H, p->Wa_b, nullptr, 0, CAPHN_GEMM_BIAS, 1, sd.s(1)));
    }
    // main -- G = f W_ih[:, E:]^T  (context side of the gate pre-activations, hoisted)
    if (parts & 2)
        RUN(caphn_gemm_f32(0, 1, BP, GH, F, f, F, p->w_ih + E, EF, ws + w.G, GH, nullptr, nullptr, 0, 0, 1, s));
    if (parts & 1) { RUN(sd.jointo(0)); RUN(sd.jointo(1)); }
    return CAPHN_OK;
}

// next-step input token of the free-running decode: mode 0 -> -1 (zero vector), 1 -> captions[b, col],
// 2 -> argmax_v logits[b, col, v] (lowest index on ties)
__global__ __launch_bounds__(256) void next_token_kernel(int B, int T, int V, int mode, int col,
                                                         const int64_t* __restrict__ caps, const float* __restrict__ logits,
                                                         int64_t* __restrict__ idx_out, int ostride) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    int64_t* idx = idx_out + (size_t)b * ostride - b;          // idx[b] below lands on idx_out[b * ostride]
    if (mode == 0) { if (tid == 0) idx[b] = -1; return; }
    if (mode == 1) { if (tid == 0) idx[b] = caps[(size_t)b * T + col]; return; }
    const float* row = logits + ((size_t)b * T + col) * V;
    float best = -INFINITY; int besti = 0x7fffffff;
    for (int v = tid; v < V; v += 256) { const float x = row[v]; if (x > best) { best = x; besti = v; } }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ob = __shfl_xor(best, m, 64); const int oi = __shfl_xor(besti, m, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 4; ++i) if (bv[i] > best || (bv[i] == best && bi[i] < besti)) { best = bv[i]; besti = bi[i]; }
        idx[b] = besti;
    }
}

// All extra GRUCells at one slot: h = layer(h, h) for every layer (decoderlstm.py:65-67, :101-103).  hin/hout are [B,H] rows
// at the given pitches; dropout (after the last layer) applies to slot > 0 only.
static int layers_forward(const caphn_decoder_dims* d, const caphn_decoder_params* p, const Ws& w, float* ws, int slot,
                          const float* hin, int hin_ld, float* hout, int hout_ld, hipStream_t s) {
    const int B = d->B, H = d->H, G3 = 3 * d->H;
    const float* in = hin; int in_ld = hin_ld;
    for (int l = 0; l < w.L1; ++l) {
        const bool last = l == w.L1 - 1;
        float* out = last ? hout : ws + ((l & 1) ? w.tmpB : w.tmpA);
        const int out_ld = last ? hout_ld : H;
        RUN(caphn_gemm_f32(0, 1, B, G3, H, in, in_ld, p->lw_ih[l], H, ws + w.tgi, G3, p->lb_ih[l], nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
        RUN(caphn_gemm_f32(0, 1, B, G3, H, in, in_ld, p->lw_hh[l], H, ws + w.tgh, G3, p->lb_hh[l], nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
        LayerFwdArgs a;
        a.B = B; a.H = H; a.S = d->T + 1; a.slot = slot;
        a.gi = ws + w.tgi; a.gh = ws + w.tgh; a.hin = in; a.hin_ld = in_ld; a.hout = out; a.hout_ld = out_ld;
        a.sin = ws + w.Lin[l]; a.sgates = ws + w.Lgates[l]; a.shn = ws + w.Lhn[l];
        if (last && slot > 0) { a.drop_p = d->dropout_p; a.drop_seed = d->dropout_seed; a.T = d->T; a.t = slot - 1; }
        RUN(caphn_launch_layer_gru_fwd(a, s));
        in = out; in_ld = out_ld;
    }
    return CAPHN_OK;
}

// Backward of layers_forward at one slot: dh of the last layer's output = d1 (+ d2) -> dout = gradient of the input.
static int layers_backward(const caphn_decoder_dims* d, const caphn_decoder_params* p, const Ws& w, float* ws, int slot,
                           const float* d1, int d1_ld, const float* d2, float* dout, int dout_ld, hipStream_t s) {
    const int B = d->B, H = d->H, G3 = 3 * d->H, S = d->T + 1;
    for (int l = w.L1 - 1; l >= 0; --l) {
        float* o = l == 0 ? dout : ws + ((l & 1) ? w.tmpB : w.tmpA);
        const int o_ld = l == 0 ? dout_ld : H;
        LayerBwdArgs a;
        a.B = B; a.H = H; a.S = S; a.slot = slot;
        a.d1 = d1; a.d1_ld = d1_ld; a.d2 = d2;
        a.sin = ws + w.Lin[l]; a.sgates = ws + w.Lgates[l]; a.shn = ws + w.Lhn[l];
        a.dgi = ws + w.Ldgi[l]; a.dgh = ws + w.Ldgh[l]; a.dout = o; a.dout_ld = o_ld;
        if (l == w.L1 - 1 && slot > 0) { a.drop_p = d->dropout_p; a.drop_seed = d->dropout_seed; a.T = d->T; a.t = slot - 1; }
        RUN(caphn_launch_layer_gru_bwd(a, s));
        RUN(caphn_gemm_f32(0, 0, B, H, G3, ws + w.Ldgi[l] + (size_t)slot * G3, S * G3, p->lw_ih[l], H, o, o_ld, nullptr, nullptr, 0,
                           CAPHN_GEMM_ACCUM, 1, s));
        RUN(caphn_gemm_f32(0, 0, B, H, G3, ws + w.Ldgh[l] + (size_t)slot * G3, S * G3, p->lw_hh[l], H, o, o_ld, nullptr, nullptr, 0,
                           CAPHN_GEMM_ACCUM, 1, s));
        d1 = o; d1_ld = o_ld; d2 = nullptr;
    }
    return CAPHN_OK;
}

}  // namespace

extern "C" size_t caphn_decoder_workspace_bytes(const caphn_decoder_dims* d) {
    if (!dims_ok(d)) return 0;
    return layout(d).total * sizeof(float);
}

extern "C" const int* caphn_decoder_rowcount_ptr(const caphn_decoder_dims* d, void* ws_) {
    if (!dims_ok(d) || !ws_) return nullptr;
    return reinterpret_cast<const int*>(static_cast<float*>(ws_) + layout(d).rowmap);
}

extern "C" unsigned long long* caphn_decoder_profile_ptr(const caphn_decoder_dims* d, void* ws_) {
    if (!dims_ok(d) || !ws_) return nullptr;
    return reinterpret_cast<unsigned long long*>(static_cast<float*>(ws_) + layout(d).prof);
}

extern "C" int caphn_decoder_prepare_rows(const caphn_decoder_dims* d, const int64_t* targets, int64_t ignore_index,
                                          void* ws_, caphn_stream_t stream) {
    if (!dims_ok(d) || !targets || !ws_) return CAPHN_EINVAL;
    const Ws w = layout(d);
    hipLaunchKernelGGL(valid_rows_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), d->B * d->T, targets,
                       ignore_index, reinterpret_cast<int*>(static_cast<float*>(ws_) + w.rowmap));
    return caphn_launch_status();
}

// x side of the forward that needs the captions and the generated W_ih: embedding lookup with the reference's zeroed
// first two inputs, then the gate pre-activations for all T
static int decoder_inputs(const caphn_decoder_dims* d, const caphn_decoder_params* p, const Ws& w, float* ws,
                          const int64_t* captions, hipStream_t s, int parts = 3) {
    const int BT = d->B * d->T, E = d->E, GH = w.NG * d->H, EF = d->E + d->F;
    int64_t* idx = reinterpret_cast<int64_t*>(ws + w.idx);
    // part 1: the lookup (needs the captions and the embedding table, not the generated weights); part 2: the gate GEMM
    if ((parts & 1) && !(d->precomputed & 64))
        hipLaunchKernelGGL(teacher_inputs_kernel, dim3(BT), dim3(256), 0, s, d->T, E, captions, p->embed_w, idx, ws + w.Xe, EF);
    if (parts & 2)
    RUN(caphn_gemm_f32(0, 1, BT, GH, E, ws + w.Xe, EF, p->w_ih, EF, ws + w.Xg, GH, p->b_ih, nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
    return CAPHN_OK;
}

extern "C" int caphn_decoder_precompute(const caphn_decoder_dims* d, const caphn_decoder_params* p, const float* features,
                                        const int64_t* captions, void* ws_, caphn_stream_t stream) {
    if (!dims_ok(d) || !p || !features || !ws_) return CAPHN_EINVAL;
    const bool raw = d->raw_features != 0;
    if (!raw && (!p->fc0_w || !p->fc0_b || !p->fc2_w || !p->fc2_b)) return CAPHN_EINVAL;
    if (d->cell == CAPHN_CELL_LSTM && (!p->initc_w || !p->initc_b)) return CAPHN_EINVAL;
    const Ws w = layout(d);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* f = nullptr;
    // captions given: the generated W_ih / b_ih are final too, so G and the x-side gates can be done as well
    // (dims.precomputed bit 1 on THIS call: the theta-independent part was issued by an earlier call -- only G is left)
    RUN(decoder_precompute(d, p, w, static_cast<float*>(ws_), features, &f, s, (captions ? 3 : 1) & ~(d->precomputed & 1), true));
    // (dims.precomputed bit 4 on THIS call: the x side is issued elsewhere -- caphn_decoder_inputs on another stream, as soon as
    //  b_ih exists -- so only G is added here)
    if (captions && !(d->precomputed & 4)) RUN(decoder_inputs(d, p, w, static_cast<float*>(ws_), captions, s));
    Side* sd = side_here();
    if (!sd || sd->post(Side::D_PRE_ALL, s) != CAPHN_OK) return CAPHN_ELAUNCH;
    sd->pre_valid = true;
    sd->pre_ws = ws_;
    return caphn_launch_status();
}

extern "C" int caphn_decoder_pair_prep(const caphn_decoder_dims* d, const caphn_decoder_params* p, void* ws_, caphn_stream_t stream) {
    if (!dims_ok(d) || !p || !ws_ || !p->Ua_w || !use_pair(d)) return CAPHN_EINVAL;
    const Ws w = layout(d);
    float* ws = static_cast<float*>(ws_);
    RUN(caphn_launch_rec_pair_prep(reinterpret_cast<unsigned long long*>(ws + w.xch), 2 * w.xch_floats * sizeof(float) / sizeof(unsigned long long),
                                   p->Ua_w, nullptr, d->H, w.NG, ws + w.wp, ws + w.dHs, (size_t)d->B * d->T * d->H,
                                   static_cast<hipStream_t>(stream)));
    return caphn_launch_status();
}
extern "C" int caphn_decoder_pair_pack_desc(const caphn_decoder_dims* d, void* ws_, caphn_pair_pack* out) {
    if (!dims_ok(d) || !ws_ || !out || !use_pair(d)) return CAPHN_EINVAL;
    const Ws w = layout(d);
    out->wp = static_cast<float*>(ws_) + w.wp;
    out->H = d->H; out->HA = caphn_rec_pair_half_a(d->H); out->pitch = caphn_rec_pair_pitch(d->H);
    out->hrows = (w.NG + 1) * out->HA;
    return CAPHN_OK;
}

extern "C" int caphn_decoder_inputs(const caphn_decoder_dims* d, const caphn_decoder_params* p, const int64_t* captions,
                                    void* ws_, caphn_stream_t stream) {
    if (!dims_ok(d) || !p || !captions || !ws_ || !p->embed_w || !p->w_ih || !p->b_ih) return CAPHN_EINVAL;
    const Ws w = layout(d);
    RUN(decoder_inputs(d, p, w, static_cast<float*>(ws_), captions, static_cast<hipStream_t>(stream)));
    return caphn_launch_status();
}
extern "C" int caphn_decoder_lookup(const caphn_decoder_dims* d, const caphn_decoder_params* p, const int64_t* captions,
                                    void* ws_, caphn_stream_t stream) {
    if (!dims_ok(d) || !p || !captions || !ws_ || !p->embed_w) return CAPHN_EINVAL;
    const Ws w = layout(d);
    RUN(decoder_inputs(d, p, w, static_cast<float*>(ws_), captions, static_cast<hipStream_t>(stream), 1));
    return caphn_launch_status();
}

extern "C" int caphn_decoder_forward(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                     const float* features, const int64_t* captions,
                                     float* logits, float* alphas, void* ws_, caphn_stream_t stream) {
    if (!dims_ok(d) || !p || !features || !captions || !logits || !ws_) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = layout(d);
    float* ws = static_cast<float*>(ws_);
    const int B = d->B, T = d->T, P = d->P, D = d->D, F = d->F, E = d->E, H = d->H, V = d->V;
    const bool lstm = d->cell == CAPHN_CELL_LSTM, raw = d->raw_features != 0;
    const int BP = B * P, BT = B * T, GH = w.NG * H, EF = E + F;
    const bool pair = use_pair(d);
    const int RG = pair ? caphn_rec_pair_resident_gates(P, H, w.NG) : caphn_rec_resident_gates(P, H, w.NG);
    if (RG < 0) return CAPHN_ELIMIT;
    if (lstm && (!p->initc_w || !p->initc_b)) return CAPHN_EINVAL;
    if (!raw && (!p->fc0_w || !p->fc0_b || !p->fc2_w || !p->fc2_b)) return CAPHN_EINVAL;
    if (!layer_params_ok(d, p)) return CAPHN_EINVAL;

    const float* f = nullptr;
    // precomputed bits: 1 the theta-independent part, 2 G, 4 the x side (embedding lookup + gate pre-activations)
    const int pc = d->precomputed;
    if (!(pc & 4)) RUN(decoder_inputs(d, p, w, ws, captions, s));
    // bit 16: part 1 was issued by caphn_decoder_precompute on ANOTHER stream and the caller has not waited for it -- this
    // composite waits itself, and only where it must: for f before the G GEMM, for init_hidden / W_a f before the recurrent
    // kernel (the G GEMM then runs beside them instead of behind them)
    Side* sdw = (pc & 16) ? side_here() : nullptr;
    if (pc & 16) {
        // the events are per device: they belong to this call only if the last precompute issued on this device filled THIS workspace
        if (!sdw || !sdw->pre_valid || sdw->pre_ws != ws_) return CAPHN_EINVAL;
        RUN(sdw->await(Side::D_PRE_F, s));
    }
    RUN(decoder_precompute(d, p, w, ws, features, &f, s, ((pc & 1) ? 0 : 1) | ((pc & 2) ? 0 : 2)));
    if (pc & 16)
        RUN(sdw->await(Side::D_PRE_ALL, s));

    RecFwdArgs a;
    a.B = B; a.T = T; a.P = P; a.H = H; a.RG = RG;
    a.Waf = ws + w.Waf; a.G = ws + w.G; a.Xg = ws + w.Xg; a.h0 = ws + w.h0; a.c0 = ws + w.c0;
    a.W_hh = p->w_hh; a.b_hh = p->b_hh; a.U_a = p->Ua_w; a.b_Ua = p->Ua_b; a.v_a = p->va_w; a.b_va = p->va_b;
    a.Hs = ws + w.Hs; a.Hprev = ws + w.Hprev; a.alphas = ws + w.alphas; a.gates = ws + w.gates; a.hn = ws + w.hn;
    a.Cs = ws + w.Cs; a.Cprev = ws + w.Cprev; a.uah = ws + w.uah;
    a.vecW = (H % 4 == 0) && caphn_aligned16(p->w_hh) && caphn_aligned16(p->Ua_w);
    a.vecS = (H % 4 == 0) && caphn_aligned16(ws);
    a.prof = reinterpret_cast<unsigned long long*>(ws + w.prof);
    a.rotate = g_tune_rec_rotate;
    a.drop_p = d->dropout_p; a.drop_seed = d->dropout_seed;
    if (pair) {
        a.xch = reinterpret_cast<unsigned long long*>(ws + w.xch);
        a.WP = ws + w.wp; a.wp_pitch = caphn_rec_pair_pitch(H);
        if (!(pc & 128))       // (bit 128: caphn_decoder_pair_prep + the optimiser's pass left everything in place)
        RUN(caphn_launch_rec_pair_prep(a.xch, 2 * w.xch_floats * sizeof(float) / sizeof(unsigned long long), p->Ua_w, p->w_hh, H, w.NG,
                                       ws + w.wp, (pc & 8) ? ws + w.dHs : nullptr, (size_t)BT * H, s));
        RUN(caphn_launch_rec_pair_fwd(a, lstm, s));
    } else if (w.L1 > 0) {
        // num_layers > 1: one launch window per step.  The attention cell writes its h_t to H0s, the extra cells take it to
        // Hs[:, t] (the h the vocabulary projection and the next step see); the initial state passes through them too.
        RUN(layers_forward(d, p, w, ws, 0, ws + w.h0, H, ws + w.h0L, H, s));
        a.h0 = ws + w.h0L; a.Hs = ws + w.H0s; a.Hsrc = ws + w.Hs; a.drop_p = 0.f;
        for (int t = 0; t < T; ++t) {
            a.t0 = t; a.t1 = t + 1;
            RUN(caphn_launch_rec_fwd(a, lstm, s));
            RUN(layers_forward(d, p, w, ws, t + 1, ws + w.H0s + (size_t)t * H, T * H, ws + w.Hs + (size_t)t * H, T * H, s));
        }
    } else
    RUN(caphn_launch_rec_fwd(a, lstm, s));
    if ((pc & 8) && !pair) RUN(caphn_zero_f32(ws + w.dHs, (size_t)BT * H, s));      // (the pair path folded it into its prep kernel)

    // vocab projection for all (b,t) at once      decoderlstm.py:105
    const int LV = d->logits_ld > 0 ? d->logits_ld : V;       // row pitch of the caller's logits buffer
    if (d->row_subset)     // only rows with a live target (caphn_decoder_prepare_rows)
        RUN(caphn_gemm_mapped(0, 1, BT, V, H, ws + w.Hs, H, p->out_w, H, logits, LV, p->out_b, CAPHN_GEMM_BIAS, 1,
                              reinterpret_cast<const int*>(ws + w.rowmap) + 4, reinterpret_cast<const int*>(ws + w.rowmap), 1, s));
    else
        RUN(caphn_gemm_f32(0, 1, BT, V, H, ws + w.Hs, H, p->out_w, H, logits, LV, p->out_b, nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
    // bit 32 (given to the forward AND the backward of a training step): the context vectors ctx_t = sum_p alpha_tp f_p -- the
    // second operand of dW_ih, which the forward never forms (it uses G) -- are left beside the embeddings now, behind the logits
    // GEMM, instead of in front of the dW_ih GEMM on the backward's chain to d theta
    if (pc & 32) RUN(caphn_launch_ctx(B, T, P, F, ws + w.alphas, f, ws + w.ctx, EF, s));
    if (alphas)
        if (hipMemcpyAsync(alphas, ws + w.alphas, sizeof(float) * (size_t)BT * P, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return CAPHN_ELAUNCH;
    return caphn_launch_status();
}

struct HyperHook { const caphn_hyper_desc* hd; const float* acts; const caphn_hyper_grads* hg; void* ws; };
static int decoder_backward_impl(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                 const float* features, const int64_t* captions, float* dlogits, const float* dalphas,
                                 const caphn_decoder_grads* g, void* ws_, const HyperHook* hook, caphn_stream_t stream);
extern "C" int caphn_decoder_backward(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                      const float* features, const int64_t* captions,
                                      float* dlogits, const float* dalphas,
                                      const caphn_decoder_grads* g, void* ws_, caphn_stream_t stream) {
    return decoder_backward_impl(d, p, features, captions, dlogits, dalphas, g, ws_, nullptr, stream);
}
extern "C" int caphn_decoder_hyper_backward(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                            const float* features, const int64_t* captions,
                                            float* dlogits, const float* dalphas,
                                            const caphn_decoder_grads* g, void* ws_,
                                            const caphn_hyper_desc* hd, const float* acts, const caphn_hyper_grads* hg, void* hyper_ws,
                                            caphn_stream_t stream) {
    if (!hd || !acts || !hg || !hyper_ws || !g) return CAPHN_EINVAL;
    const size_t NG = d && d->cell == CAPHN_CELL_LSTM ? 4 : 3;
    if (d) {   // dtheta must be one contiguous block in theta order
        const size_t GH = NG * d->H, EF = (size_t)d->E + d->F;
        if (g->w_hh != g->w_ih + GH * EF || g->b_ih != g->w_hh + GH * d->H || g->b_hh != g->b_ih + GH) return CAPHN_EINVAL;
    }
    HyperHook hook{hd, acts, hg, hyper_ws};
    return decoder_backward_impl(d, p, features, captions, dlogits, dalphas, g, ws_, &hook, stream);
}
static int decoder_backward_impl(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                 const float* features, const int64_t* captions, float* dlogits, const float* dalphas,
                                 const caphn_decoder_grads* g, void* ws_, const HyperHook* hook, caphn_stream_t stream) {
    if (!dims_ok(d) || !p || !features || !captions || !dlogits || !g || !ws_) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = layout(d);
    float* ws = static_cast<float*>(ws_);
    const int B = d->B, T = d->T, P = d->P, D = d->D, F = d->F, E = d->E, H = d->H, V = d->V;
    const bool lstm = d->cell == CAPHN_CELL_LSTM, raw = d->raw_features != 0;
    const int BP = B * P, BT = B * T, GH = w.NG * H, EF = E + F;
    const bool pair = use_pair(d);
    const int RG = pair ? caphn_rec_pair_resident_gates(P, H, w.NG) : caphn_rec_resident_gates(P, H, w.NG);
    if (RG < 0) return CAPHN_ELIMIT;
    if (lstm && (!g->initc_w || !g->initc_b)) return CAPHN_EINVAL;
    if (!raw && (!g->fc0_w || !g->fc0_b || !g->fc2_w || !g->fc2_b)) return CAPHN_EINVAL;    // before any branch is forked
    if (!layer_params_ok(d, p)) return CAPHN_EINVAL;
    for (int l = 0; l < w.L1; ++l) if (!g->lw_ih[l] || !g->lw_hh[l] || !g->lb_ih[l] || !g->lb_hh[l]) return CAPHN_EINVAL;
    void* cws = ws + w.colws;
    const int64_t* idx = reinterpret_cast<const int64_t*>(ws + w.idx);
    const float* f = raw ? features : ws + w.f;
    float* dgi = ws + w.dgi;
    float* dgh = lstm ? dgi : ws + w.dgh;          // LSTM: d(gi) == d(gh)

    Side* sdp = side_here();
    if (!sdp) return CAPHN_ELAUNCH;
    Side& sd = *sdp;
    RUN(sd.begin(s, g_tune_fork != 0));
    Side::Scope scope(sd);
    void* cw0 = ws + w.colws_s[0]; void* cw1 = ws + w.colws_s[1]; void* cw2 = ws + w.colws_s[2];

    // vocab projection.  dHs = dlogits W feeds BPTT (main); dW = dlogits^T Hs and db = colsum(dlogits) are only
    // needed by the optimiser: branch 0 computes them beside the BPTT kernel, which occupies B of the 256 CUs.
    const int* rmap = d->row_subset ? reinterpret_cast<const int*>(ws + w.rowmap) : nullptr;
    const int LV = d->logits_ld > 0 ? d->logits_ld : V;       // row pitch of the caller's d logits buffer
    const int fork_mode = g_tune_fork == 4 ? (pair ? 2 : 1) : g_tune_fork;
    const bool late = fork_mode != 2;
    const bool gz = d->grads_zeroed != 0;
    const bool wg_first = !late && g_tune_vocab_order == 0;     // vocabulary gradients beside the dHs GEMM (and BPTT after it)
    if (wg_first) {
        RUN(sd.forkto(0));
        RUN(wgrad_bias(V, H, BT, dlogits, LV, ws + w.Hs, H, g->out_w, H, g->out_b, rmap, cws, sd.s(0), gz));
        RUN(sd.milestone(CAPHN_MS_VOCAB, sd.s(0)));
    }
    const bool dhs_zero = (d->precomputed & 8) != 0;      // the forward of this step left d Hs zero-filled (dims.precomputed bit 8)
    bool ctx_side = false;
    if (rmap) {   // rows of ignored targets have d logits == 0: dHs of those rows is zero, the others are gathered
        if (!dhs_zero) RUN(caphn_zero_f32(ws + w.dHs, (size_t)BT * H, s));
        RUN(caphn_gemm_mapped(0, 0, BT, H, V, dlogits, LV, p->out_w, H, ws + w.dHs, H, nullptr, 0,
                              (g_tune_sk_dhs > 0 && !g_tune_deterministic) ? g_tune_sk_dhs : pick_splitk(BT, H, V),
                              rmap + 4, rmap, 1, s));
    } else
    RUN(gemm_auto(0, 0, BT, H, V, dlogits, LV, p->out_w, H, ws + w.dHs, H, nullptr, 0, s, nullptr, 0, dhs_zero));
    if (!late && !wg_first) {
        // ... or only beside BPTT: dHs = dlogits W_fc sits on the chain into BPTT and ran 135 instead of ~105 us with the
        // vocabulary gradients streaming the same 99 MB of d logits next to it; BPTT alone is long enough to cover them
        // the context vectors ctx_t = sum_p alpha_tp f_p (second operand of dW_ih, which the forward never forms) go to a branch of
        // their own, beside BPTT: in front of the dW_ih GEMM they were 27 us on the chain to d theta, behind the logits GEMM in the
        // forward (dims.precomputed bit 32) 9 us + a launch gap between the loss and BPTT; here nothing waits for them (beside BPTT
        // the kernel is starved to ~180 us and still ends before BPTT does)
        if (!(d->precomputed & 32) && sd.s(0) != s && sd.s(2) != s) {
            RUN(sd.fork_many({0, 2}));
            RUN(caphn_launch_ctx(B, T, P, F, ws + w.alphas, f, ws + w.ctx, EF, sd.s(2)));
            RUN(sd.record(5, sd.s(2)));
            ctx_side = true;
        } else
        RUN(sd.forkto(0));
        RUN(wgrad_bias(V, H, BT, dlogits, LV, ws + w.Hs, H, g->out_w, H, g->out_b, rmap, cws, sd.s(0), gz));
        RUN(sd.milestone(CAPHN_MS_VOCAB, sd.s(0)));
    }

    RecBwdArgs a;
    a.B = B; a.T = T; a.P = P; a.H = H; a.RG = RG;
    a.Waf = ws + w.Waf; a.G = ws + w.G; a.W_hh = p->w_hh; a.U_a = p->Ua_w; a.v_a = p->va_w;
    a.Hprev = ws + w.Hprev; a.alphas = ws + w.alphas; a.gates = ws + w.gates; a.hn = ws + w.hn; a.uah = ws + w.uah;
    a.Cs = ws + w.Cs; a.Cprev = ws + w.Cprev;
    a.dHs = ws + w.dHs; a.dalphas = dalphas;
    a.dgi = dgi; a.dgh = dgh; a.duah = ws + w.duah; a.de = ws + w.de; a.dh0 = ws + w.dh0; a.dc0 = ws + w.dc0;
    a.vecW = (H % 4 == 0) && caphn_aligned16(p->w_hh) && caphn_aligned16(p->Ua_w);
    a.vecS = (H % 4 == 0) && caphn_aligned16(ws);
    a.prof = reinterpret_cast<unsigned long long*>(ws + w.prof) + 8;
    a.rotate = g_tune_rec_rotate;
    // attention parameter gradients (dWaf, partial d v_a / d b_va) come out of the BPTT kernel itself when its thread
    // map can carry them (it evaluates the same tanh for d(U_a h)): one ~50 us kernel less on the chain
    a.drop_p = d->dropout_p; a.drop_seed = d->dropout_seed;
    // (a windowed BPTT cannot keep the fused accumulators across launches: layered decoders use the separate kernel)
    const int ang = w.L1 > 0 ? 0 : pair ? caphn_rec_pair_bwd_groups(P, H) : caphn_rec_bwd_groups(P, H);
    if (ang > 0) { a.dWaf = ws + w.dWaf; a.apart = ws + w.apart; }
    if (w.L1 > 0) {
        // per step, top down: the extra cells' backward turns d h_t (vocab projection + what step t + 1 sent back) into the
        // gradient of the attention cell's output, one BPTT window consumes it and leaves d h_{t-1} in dcarry
        RUN(caphn_zero_f32(ws + w.dcarry, (size_t)B * H, s));
        a.dHs = ws + w.dH0; a.dh0 = ws + w.dcarry; a.drop_p = 0.f;
        for (int t = T - 1; t >= 0; --t) {
            RUN(layers_backward(d, p, w, ws, t + 1, ws + w.dHs + (size_t)t * H, T * H, ws + w.dcarry, ws + w.dH0 + (size_t)t * H, T * H, s));
            a.t0 = t; a.t1 = t + 1;
            RUN(caphn_launch_rec_bwd(a, lstm, s));
        }
        RUN(layers_backward(d, p, w, ws, 0, ws + w.dcarry, H, nullptr, ws + w.dh0, H, s));
    } else if (pair) {
        a.xch = reinterpret_cast<unsigned long long*>(ws + w.xch) + w.xch_floats * sizeof(float) / sizeof(unsigned long long);
        a.apart_rows = ang;
        a.WP = ws + w.wp; a.wp_pitch = caphn_rec_pair_pitch(H);
        if (!raw) { a.inith_w = p->inith_w; a.initc_w = lstm ? p->initc_w : nullptr; a.dmean_part = ws + w.dmeanf; a.F = F; }
        RUN(caphn_launch_rec_pair_bwd(a, lstm, s));
    } else
    RUN(caphn_launch_rec_bwd(a, lstm, s));

    // ---- after BPTT.  Two chains and their leaves:
    //   sT  ctx -> dW_ih (-> d theta complete once dW_hh is) -> hypernet VJP: the longest one, on branch 1 (caphn_tune(15, 1) puts it
    //       on the caller's stream instead: measured slower).
    //   sF  attn_param_grads -> df -> dY1 -> dW_fc0 (attention / feature_fc), on the caller's stream.
    //   b2  dW_hh, then the leaves of sF;  b0  vocabulary gradients (beside or after BPTT), the embedding gradient, BPTT's leaves.
    // Events: E2 dWaf ready (sF)   E3 dW_hh ready (b2)   E4 df ready (sF).  Column-sum scratch: cw0 sF, cw1 sT, cw2 b2, cws b0.
    hipStream_t b0 = sd.s(0), b1 = sd.s(1), b2 = sd.s(2);
    const bool t_main = hook != nullptr && g_tune_chain_main != 0;
    hipStream_t sT = t_main ? s : b1, sF = t_main ? b1 : s;
    const bool hold_big = late && !raw && fork_mode == 3;   // 3: release the two big leaves only once df exists (measured: no gain)
    RUN(sd.fork_many({0, 1, 2}));
    if (late && !hold_big) {   // the optimiser-only vocab gradients
        RUN(wgrad_bias(V, H, BT, dlogits, LV, ws + w.Hs, H, g->out_w, H, g->out_b, rmap, cws, b0, gz));
        RUN(sd.milestone(CAPHN_MS_VOCAB, b0));
    }
    // sT -- input weights dW_ih = dgi^T [Xe | ctx] (ctx lands beside the embeddings, so this is ONE GEMM: two back to back on the
    // chain to d theta cost 72 + 57 us)
    if (ctx_side) RUN(sd.wait(5, sT));
    else if (!(d->precomputed & 32)) RUN(caphn_launch_ctx(B, T, P, F, ws + w.alphas, f, ws + w.ctx, EF, sT));      // (else: left by the forward)
    if (lstm) RUN(gemm_auto(1, 0, GH, EF, BT, dgi, GH, ws + w.Xe, EF, g->w_ih, EF, nullptr, 0, sT, nullptr, 0, gz));
    else RUN(wgrad_bias(GH, EF, BT, dgi, GH, ws + w.Xe, EF, g->w_ih, EF, g->b_ih, nullptr, cw1, sT, gz));      // + db_ih
    // b2 -- recurrent weights dW_hh = dgh^T Hprev (+ db_hh), then the embedding gradient
    RUN(wgrad_bias(GH, H, BT, dgh, GH, ws + w.Hprev, H, g->w_hh, H, g->b_hh, nullptr, cw2, b2, gz));
    if (lstm) RUN(caphn_colsum_f32(BT, GH, dgi, GH, g->b_ih, cw2, b2));      // dgh aliases dgi: db_ih == db_hh
    RUN(sd.record(3, b2));
    // b0 -- the embedding gradient first (7.7 MB, the second-largest bucket of a data-parallel exchange: its milestone should
    // come as early as BPTT allows), on b2 when the vocabulary gradients occupy b0 after BPTT
    hipStream_t se = late ? b2 : b0;
    RUN(caphn_gemm_f32(0, 0, BT, E, GH, dgi, GH, p->w_ih, EF, ws + w.dXe, E, nullptr, nullptr, 0, 0, 1, se));
    if (!gz) RUN(caphn_zero_f32(g->embed_w, (size_t)V * E, se));
    RUN(caphn_embedding_scatter_add_v(BT, E, V, ws + w.dXe, idx, g->embed_w, se));
    RUN(sd.milestone(CAPHN_MS_EMBED, se));
    // dL/dtheta = [dW_ih | dW_hh | db_ih | db_hh] is complete once E3 has fired on top of sT's own work so far: a
    // data-parallel caller starts its all-gather of the rank-1 row factors here (CAPHN_MS_DTHETA)
    RUN(sd.wait(3, sT));
    RUN(sd.milestone(CAPHN_MS_DTHETA, sT));
    if (hook && !hold_big) {
        // the hypernet VJP (HBM-bound transposed GEMV over the 576 MB of second-layer weights), beside the sF chain
        RUN(caphn_hyper_backward(hook->hd, g->w_ih, hook->acts, hook->hg, hook->ws, sT));
    }
    if (!hold_big) RUN(sd.milestone(CAPHN_MS_HYPER, sT));
    // sF -- the chain's own small inputs stay on the chain's stream: waiting on a side-stream event here stalled the chain for
    // ~275 us in the kernel trace although the producers had long finished
    if (!raw) {
        RUN(caphn_gemm_f32(0, 0, BT, F, GH, dgi, GH, p->w_ih + E, EF, ws + w.dctx, F, nullptr, nullptr, 0, 0, 1, sF));
        if (!pair)      // (the pair BPTT kernel leaves d mean_f as two partials in its epilogue)
            RUN(caphn_launch_dmean(B, H, F, ws + w.dh0, p->inith_w, lstm ? ws + w.dc0 : nullptr, lstm ? p->initc_w : nullptr,
                                   ws + w.dmeanf, sF));
    }
    // attention parameter gradients (dWaf, partial d v_a) when the BPTT kernel did not fuse them
    AttnGradArgs ag;
    ag.T = T; ag.P = P; ag.H = H; ag.pchunk = w.pchunk;
    ag.Waf = ws + w.Waf; ag.uah = ws + w.uah; ag.de = ws + w.de; ag.v_a = p->va_w;
    ag.dWaf = ws + w.dWaf; ag.part = ws + w.apart;
    if (ang == 0) RUN(caphn_launch_attn_param_grads(ag, B, w.npc, sF));
    RUN(sd.record(2, sF));
    // b0 -- leaves of BPTT that nothing else waits for (behind the vocabulary gradients when those run after BPTT)
    RUN(wgrad_bias(H, H, BT, ws + w.duah, H, ws + w.Hprev, H, g->Ua_w, H, g->Ua_b, nullptr, cws, b0, gz));
    for (int l = 0; l < w.L1; ++l) {     // extra cells: one transposed GEMM over all B (T + 1) slots per weight
        const int BS = B * (T + 1), G3 = 3 * H;
        RUN(wgrad_bias(G3, H, BS, ws + w.Ldgi[l], G3, ws + w.Lin[l], H, g->lw_ih[l], H, g->lb_ih[l], nullptr, cws, b0, gz));
        RUN(wgrad_bias(G3, H, BS, ws + w.Ldgh[l], G3, ws + w.Lin[l], H, g->lw_hh[l], H, g->lb_hh[l], nullptr, cws, b0, gz));
    }
    RUN(wgrad_bias(H, F, B, ws + w.dh0, H, ws + w.meanf, F, g->inith_w, F, g->inith_b, nullptr, cws, b0, gz));
    if (lstm) RUN(wgrad_bias(H, F, B, ws + w.dc0, H, ws + w.meanf, F, g->initc_w, F, g->initc_b, nullptr, cws, b0, gz));
    // sF -- df = alpha^T dctx + dmean/P + dWaf W_a, then feature_fc backward
    if (!raw) {
        RUN(caphn_launch_df(B, T, P, F, ws + w.alphas, ws + w.dctx, ws + w.dmeanf, ws + w.df, sF, pair ? ws + w.dmeanf + (size_t)B * F : nullptr));
        RUN(caphn_gemm_f32(0, 0, BP, F, H, ws + w.dWaf, H, p->Wa_w, F, ws + w.df, F, nullptr, nullptr, 0, CAPHN_GEMM_ACCUM, 1, sF));
        RUN(sd.record(4, sF));
        if (hold_big) {
            // The chain's small kernels are starved while big leaf kernels fill the machine (a 160-275 us stall of
            // df_kernel in the kernel trace), so the latency-critical front of the chain runs first; the optimiser-only
            // vocab gradients (b0) and the hypernet VJP (sT) start here, beside the two remaining chain GEMMs.
            RUN(sd.wait(4, b0));
            RUN(wgrad_bias(V, H, BT, dlogits, LV, ws + w.Hs, H, g->out_w, H, g->out_b, rmap, cws, b0, gz));
            RUN(sd.milestone(CAPHN_MS_VOCAB, b0));
            if (hook) {
                RUN(sd.wait(4, sT));
                RUN(caphn_hyper_backward(hook->hd, g->w_ih, hook->acts, hook->hg, hook->ws, sT));
            }
            RUN(sd.milestone(CAPHN_MS_HYPER, sT));
        }
        RUN(caphn_gemm_f32(0, 0, BP, F, F, ws + w.df, F, p->fc2_w, F, ws + w.dY1, F, nullptr, ws + w.Y1, F, CAPHN_GEMM_MASK, 1, sF));
        RUN(wgrad_bias(F, D, BP, ws + w.dY1, F, features, D, g->fc0_w, D, g->fc0_b, nullptr, cw0, sF, gz));
    }
    // b2 (leaves of sF) -- d v_a, d b_va, dW_a, db_Wa once dWaf exists; fc2 gradients once df exists
    RUN(sd.wait(2, b2));
    RUN(caphn_colsum_f32(B * (ang > 0 ? ang : w.npc), H + 1, ws + w.apart, H + 1, ws + w.vtmp, cw2, b2));
    hipLaunchKernelGGL(copy2_kernel, dim3((H + 256) / 256), dim3(256), 0, b2, ws + w.vtmp, H, g->va_w, g->va_b);   // one launch, no runtime blits
    RUN(wgrad_bias(H, F, BP, ws + w.dWaf, H, f, F, g->Wa_w, F, g->Wa_b, nullptr, cw2, b2, gz));
    if (!raw) {
        RUN(sd.wait(4, b2));
        RUN(wgrad_bias(F, F, BP, ws + w.df, F, ws + w.Y1, F, g->fc2_w, F, g->fc2_b, nullptr, cw2, b2, gz));
    }
    if (g_tune_join_chain) { RUN(sd.join_into(0, 1)); RUN(sd.join_into(2, 1)); RUN(sd.jointo(1)); }
    else { RUN(sd.jointo(0)); RUN(sd.jointo(1)); RUN(sd.jointo(2)); }
    (void)captions;
    return caphn_launch_status();
}

extern "C" int caphn_decoder_backward_milestone(int which, caphn_stream_t waiter) {
    if (which < 0 || which >= CAPHN_MS_COUNT) return CAPHN_EINVAL;
    Side* sd = side_here();
    if (!sd) return CAPHN_ELAUNCH;
    if (!sd->ready) return CAPHN_OK;      // no backward composite has run on this device: nothing to wait for
    return sd->await(Side::D_MS + which, static_cast<hipStream_t>(waiter));
}

// Free-running / scheduled-sampling forward (no backward state kept: validation and inference).
//   GRU  (AttentionGru.forward, decoderlstm.py:78-96):  x_t = 0 for t = 0 (and t = 1 when not sampling: the zeroed
//        view), teacher embed[caps[:,t-1]] when step t does not sample, embed[argmax logits_{t-1}] when it does.
//   LSTM (AttentionLstm.forward, :236-251): the sampled embedding is produced AFTER fc at a sampling step and only
//        consumed by a later sampling step, so a sampling step t reuses whatever word_embed the previous
//        iteration left behind.
// use_sampling: HOST array of T flags (the per-step draws np.random.random() < sample_prob; entry 0 is ignored).
extern "C" int caphn_decoder_forward_sampled(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                             const float* features, const int64_t* captions,
                                             const unsigned char* use_sampling,
                                             float* logits, float* alphas, void* ws_, caphn_stream_t stream) {
    if (!dims_ok(d) || !p || !features || !captions || !use_sampling || !logits || !ws_ || d->logits_ld != 0) return CAPHN_EINVAL;
    if (d->dropout_p > 0.f) return CAPHN_EINVAL;      // forward only: evaluation, no dropout
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = layout(d);
    float* ws = static_cast<float*>(ws_);
    const int B = d->B, T = d->T, P = d->P, E = d->E, F = d->F, H = d->H, V = d->V;
    const bool lstm = d->cell == CAPHN_CELL_LSTM;
    const int GH = w.NG * H, EF = E + F;
    const int RG = caphn_rec_resident_gates(P, H, w.NG);
    if (RG < 0) return CAPHN_ELIMIT;
    if (lstm && (!p->initc_w || !p->initc_b)) return CAPHN_EINVAL;
    const float* f = nullptr;
    if (!layer_params_ok(d, p)) return CAPHN_EINVAL;
    RUN(decoder_precompute(d, p, w, ws, features, &f, s));
    if (w.L1 > 0) RUN(layers_forward(d, p, w, ws, 0, ws + w.h0, H, ws + w.h0L, H, s));
    int64_t* idx = reinterpret_cast<int64_t*>(ws + w.idx);
    // source of word_embed: 0 zero, 1 teacher column `col`, 2 sampled from logits column `col`
    int src_mode = 0, src_col = 0;
    for (int t = 0; t < T; ++t) {
        const bool samp = t > 0 && use_sampling[t] != 0;
        if (!samp) {
            // t == 0: zeroed view of embed[:,0,:]; t == 1 reads that same zeroed view (decoderlstm.py:82-88)
            if (t < 2) { src_mode = 0; } else { src_mode = 1; src_col = t - 1; }
        } else if (!lstm) {
            src_mode = 2; src_col = t - 1;              // GRU: argmax of the previous output, taken now
        }                                               // LSTM: keep what the previous iteration left
        hipLaunchKernelGGL(next_token_kernel, dim3(B), dim3(256), 0, s, B, T, V, src_mode, src_col, captions, logits, idx, 1);
        RUN(caphn_embedding_gather(B, E, p->embed_w, idx, ws + w.Xe, s));
        RUN(caphn_gemm_f32(0, 1, B, GH, E, ws + w.Xe, E, p->w_ih, EF, ws + w.Xg, GH, p->b_ih, nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
        RecFwdArgs a;
        a.B = B; a.T = 1; a.P = P; a.H = H; a.RG = RG;
        a.Waf = ws + w.Waf; a.G = ws + w.G; a.Xg = ws + w.Xg;
        a.h0 = t == 0 ? ws + (w.L1 > 0 ? w.h0L : w.h0) : ws + w.Hs + (size_t)(t - 1) * B * H;
        a.c0 = lstm ? (t == 0 ? ws + w.c0 : ws + w.Cs + (size_t)(t - 1) * B * H) : nullptr;
        a.W_hh = p->w_hh; a.b_hh = p->b_hh; a.U_a = p->Ua_w; a.b_Ua = p->Ua_b; a.v_a = p->va_w; a.b_va = p->va_b;
        a.Hs = ws + w.Hs + (size_t)t * B * H; a.Hprev = ws + w.Hprev; a.alphas = ws + w.alphas;
        a.gates = ws + w.gates; a.hn = ws + w.hn; a.Cs = ws + w.Cs + (size_t)t * B * H; a.Cprev = ws + w.Cprev; a.uah = ws + w.uah;
        a.prof = nullptr; a.rotate = 0;
        a.vecW = (H % 4 == 0) && caphn_aligned16(p->w_hh) && caphn_aligned16(p->Ua_w);
        a.vecS = (H % 4 == 0) && caphn_aligned16(ws);
        float* ht = a.Hs;
        if (w.L1 > 0) a.Hs = ws + w.H0s;
        RUN(caphn_launch_rec_fwd(a, lstm, s));
        if (w.L1 > 0) RUN(layers_forward(d, p, w, ws, t + 1, ws + w.H0s, H, ht, H, s));
        // logits[:, t, :] = h_t W_fc^T + b   (leading dimension T*V)
        RUN(caphn_gemm_f32(0, 1, B, V, H, ht, H, p->out_w, H, logits + (size_t)t * V, T * V, p->out_b, nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
        if (alphas)
            if (hipMemcpy2DAsync(alphas + (size_t)t * P, sizeof(float) * (size_t)T * P, ws + w.alphas, sizeof(float) * P,
                                 sizeof(float) * P, B, hipMemcpyDeviceToDevice, s) != hipSuccess) return CAPHN_ELAUNCH;
        if (samp && lstm) { src_mode = 2; src_col = t; }    // :247-251: sampled embedding of THIS output, used later
    }
    return caphn_launch_status();
}

// The same forward KEEPING the backward state (training with sample_prob > 0: train_gru.py:84 calls the captioner with 1.0
// inside training_step; cc_train_hypernet.py trains with 0.0).  The argmax is not differentiable, so autograd's graph of the
// reference is the teacher-forced one with the sampled token ids in place of the caption's: the recurrent kernel runs one
// time-step window per launch on the [B, T] workspace layout, the sampled ids are kept in the workspace's idx array (the
// backward scatters d x_t into THEIR embedding rows), and caphn_decoder_backward applies unchanged.
extern "C" int caphn_decoder_forward_sampled_train(const caphn_decoder_dims* d, const caphn_decoder_params* p,
                                                   const float* features, const int64_t* captions,
                                                   const unsigned char* use_sampling,
                                                   float* logits, float* alphas, void* ws_, caphn_stream_t stream) {
    if (!dims_ok(d) || !p || !features || !captions || !use_sampling || !logits || !ws_) return CAPHN_EINVAL;
    if (d->row_subset || d->precomputed || d->logits_ld != 0) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = layout(d);
    float* ws = static_cast<float*>(ws_);
    const int B = d->B, T = d->T, P = d->P, E = d->E, F = d->F, H = d->H, V = d->V;
    const bool lstm = d->cell == CAPHN_CELL_LSTM;
    const int GH = w.NG * H, EF = E + F;
    const bool pair = use_pair(d);
    const int RG = pair ? caphn_rec_pair_resident_gates(P, H, w.NG) : caphn_rec_resident_gates(P, H, w.NG);
    if (RG < 0) return CAPHN_ELIMIT;
    if (lstm && (!p->initc_w || !p->initc_b)) return CAPHN_EINVAL;
    if (!layer_params_ok(d, p)) return CAPHN_EINVAL;
    const float* f = nullptr;
    RUN(decoder_precompute(d, p, w, ws, features, &f, s));
    int64_t* idx = reinterpret_cast<int64_t*>(ws + w.idx);
    RecFwdArgs a;
    a.B = B; a.T = T; a.P = P; a.H = H; a.RG = RG;
    a.Waf = ws + w.Waf; a.G = ws + w.G; a.Xg = ws + w.Xg; a.h0 = ws + w.h0; a.c0 = ws + w.c0;
    a.W_hh = p->w_hh; a.b_hh = p->b_hh; a.U_a = p->Ua_w; a.b_Ua = p->Ua_b; a.v_a = p->va_w; a.b_va = p->va_b;
    a.Hs = ws + w.Hs; a.Hprev = ws + w.Hprev; a.alphas = ws + w.alphas; a.gates = ws + w.gates; a.hn = ws + w.hn;
    a.Cs = ws + w.Cs; a.Cprev = ws + w.Cprev; a.uah = ws + w.uah;
    a.vecW = (H % 4 == 0) && caphn_aligned16(p->w_hh) && caphn_aligned16(p->Ua_w);
    a.vecS = (H % 4 == 0) && caphn_aligned16(ws);
    a.prof = nullptr; a.rotate = 0;
    a.drop_p = d->dropout_p; a.drop_seed = d->dropout_seed;
    if (pair) {
        a.xch = reinterpret_cast<unsigned long long*>(ws + w.xch);
        a.WP = ws + w.wp; a.wp_pitch = caphn_rec_pair_pitch(H);
        RUN(caphn_launch_rec_pair_prep(a.xch, 2 * w.xch_floats * sizeof(float) / sizeof(unsigned long long), p->Ua_w, p->w_hh, H, w.NG,
                                       ws + w.wp, nullptr, 0, s));
    }
    if (w.L1 > 0) {     // as in caphn_decoder_forward
        RUN(layers_forward(d, p, w, ws, 0, ws + w.h0, H, ws + w.h0L, H, s));
        a.h0 = ws + w.h0L; a.Hs = ws + w.H0s; a.Hsrc = ws + w.Hs; a.drop_p = 0.f;
    }
    int src_mode = 0, src_col = 0;
    for (int t = 0; t < T; ++t) {
        const bool samp = t > 0 && use_sampling[t] != 0;
        if (!samp) { if (t < 2) { src_mode = 0; } else { src_mode = 1; src_col = t - 1; } }
        else if (!lstm) { src_mode = 2; src_col = t - 1; }
        // x_t: token id -> idx[b, t] -> Xe[b, t, :] -> x-side gate pre-activations of step t
        hipLaunchKernelGGL(next_token_kernel, dim3(B), dim3(256), 0, s, B, T, V, src_mode, src_col, captions, logits, idx + t, T);
        RUN(caphn_embedding_gather_strided(B, E, p->embed_w, idx + t, T, ws + w.Xe + (size_t)t * EF, T * EF, s));
        RUN(caphn_gemm_f32(0, 1, B, GH, E, ws + w.Xe + (size_t)t * EF, T * EF, p->w_ih, EF, ws + w.Xg + (size_t)t * GH, T * GH, p->b_ih,
                           nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
        a.t0 = t; a.t1 = t + 1;
        if (pair) RUN(caphn_launch_rec_pair_fwd(a, lstm, s)); else RUN(caphn_launch_rec_fwd(a, lstm, s));
        if (w.L1 > 0) RUN(layers_forward(d, p, w, ws, t + 1, ws + w.H0s + (size_t)t * H, T * H, ws + w.Hs + (size_t)t * H, T * H, s));
        RUN(caphn_gemm_f32(0, 1, B, V, H, ws + w.Hs + (size_t)t * H, T * H, p->out_w, H, logits + (size_t)t * V, T * V, p->out_b, nullptr, 0,
                           CAPHN_GEMM_BIAS, 1, s));
        if (samp && lstm) { src_mode = 2; src_col = t; }
    }
    if (alphas)
        if (hipMemcpyAsync(alphas, ws + w.alphas, sizeof(float) * (size_t)B * T * P, hipMemcpyDeviceToDevice, s) != hipSuccess) return CAPHN_ELAUNCH;
    return caphn_launch_status();
}

// ------------------------------------------------------------------------------------------------ search
// Beam search (hypernet_attention.py:251-306) and greedy search (models/decoderlstm.py:138-175) for a batch of
// images; the decode state lives in the search workspace (search.hip) and one step is six launches on `stream`.
namespace {
struct SearchWs {   // byte offsets
    size_t h_cur, logits, cand_val, cand_idx, score, idx, k_alive, n_comp, seqs0, seqs1, comp_seqs, comp_len, comp_score, total;
};
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
inline bool search_ok(const caphn_decoder_dims* d, const caphn_search_cfg* c) {
    return dims_ok(d) && c && d->cell == CAPHN_CELL_GRU && d->T == 1 && c->beam >= 1 && c->beam <= SEARCH_MAX_BEAM &&
           c->n_images >= 1 && d->B == c->n_images * c->beam && c->max_steps >= 1 && !d->row_subset;
}
inline SearchWs search_layout(const caphn_decoder_dims* d, const caphn_search_cfg* c) {
    SearchWs w;
    const size_t R = d->B, k = c->beam, N = c->n_images, L = (size_t)c->max_steps + 1;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += up256(bytes); return r; };
    w.h_cur = take(4 * R * d->H); w.logits = take(4 * R * d->V); w.cand_val = take(4 * R * k); w.cand_idx = take(4 * R * k);
    w.score = take(4 * R); w.idx = take(8 * R); w.k_alive = take(4 * N); w.n_comp = take(4 * N);
    w.seqs0 = take(8 * R * L); w.seqs1 = take(8 * R * L);
    w.comp_seqs = take(8 * N * k * L); w.comp_len = take(4 * N * k); w.comp_score = take(4 * N * k);
    w.total = o;
    return w;
}
inline SearchArgs search_args(const caphn_decoder_dims* d, const caphn_search_cfg* c, const Ws& w, float* ws, char* sw) {
    const SearchWs q = search_layout(d, c);
    SearchArgs a;
    a.n_images = c->n_images; a.beam = c->beam; a.max_steps = c->max_steps; a.H = d->H;
    a.zero_pad_rule = c->zero_pad_rule; a.end_token = c->end_token;
    a.h_cur = reinterpret_cast<float*>(sw + q.h_cur); a.h_new = ws + w.Hs;
    a.score = reinterpret_cast<float*>(sw + q.score); a.idx = reinterpret_cast<int64_t*>(sw + q.idx);
    a.cand_val = reinterpret_cast<float*>(sw + q.cand_val); a.cand_idx = reinterpret_cast<int*>(sw + q.cand_idx);
    a.k_alive = reinterpret_cast<int*>(sw + q.k_alive); a.n_comp = reinterpret_cast<int*>(sw + q.n_comp);
    a.seqs[0] = reinterpret_cast<int64_t*>(sw + q.seqs0); a.seqs[1] = reinterpret_cast<int64_t*>(sw + q.seqs1);
    a.comp_seqs = reinterpret_cast<int64_t*>(sw + q.comp_seqs); a.comp_len = reinterpret_cast<int*>(sw + q.comp_len);
    a.comp_score = reinterpret_cast<float*>(sw + q.comp_score);
    return a;
}
}  // namespace

extern "C" size_t caphn_decoder_search_workspace_bytes(const caphn_decoder_dims* d, const caphn_search_cfg* c) {
    if (!search_ok(d, c)) return 0;
    return search_layout(d, c).total;
}

extern "C" int caphn_decoder_search_begin(const caphn_decoder_dims* d, const caphn_decoder_params* p, const caphn_search_cfg* c,
                                          const float* features, void* ws_, void* sws_, caphn_stream_t stream) {
    if (!search_ok(d, c) || !p || !features || !ws_ || !sws_) return CAPHN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = layout(d);
    float* ws = static_cast<float*>(ws_);
    if (caphn_rec_resident_gates(d->P, d->H, w.NG) < 0) return CAPHN_ELIMIT;
    // feature_fc / init_hidden / W_a f / G once per IMAGE: the beams of an image share its slabs (slab_div)
    caphn_decoder_dims di = *d;
    di.B = c->n_images;
    const float* f = nullptr;
    if (!layer_params_ok(d, p)) return CAPHN_EINVAL;
    RUN(decoder_precompute(&di, p, w, ws, features, &f, s));
    if (w.L1 > 0) RUN(layers_forward(&di, p, w, ws, 0, ws + w.h0, d->H, ws + w.h0L, d->H, s));
    const SearchArgs a = search_args(d, c, w, ws, static_cast<char*>(sws_));
    return caphn_launch_search_init(a, ws + (w.L1 > 0 ? w.h0L : w.h0), c->first_token, c->lookup_first, s);
}

extern "C" int caphn_decoder_search_steps(const caphn_decoder_dims* d, const caphn_decoder_params* p, const caphn_search_cfg* c,
                                          int step0, int nsteps, float* alphas, void* ws_, void* sws_, caphn_stream_t stream) {
    if (!search_ok(d, c) || !p || !ws_ || !sws_ || step0 < 1 || nsteps < 0 || step0 + nsteps - 1 > c->max_steps) return CAPHN_EINVAL;
    if (!layer_params_ok(d, p)) return CAPHN_EINVAL;
    if (alphas && c->beam != 1) return CAPHN_EINVAL;      // attention maps are only tracked without beam re-ordering
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Ws w = layout(d);
    float* ws = static_cast<float*>(ws_);
    char* sw = static_cast<char*>(sws_);
    const SearchWs q = search_layout(d, c);
    const SearchArgs sa = search_args(d, c, w, ws, sw);
    const int R = d->B, P = d->P, E = d->E, F = d->F, H = d->H, V = d->V, GH = w.NG * H, EF = E + F;
    const int RG = caphn_rec_resident_gates(P, H, w.NG);
    if (RG < 0) return CAPHN_ELIMIT;
    float* logits = reinterpret_cast<float*>(sw + q.logits);
    for (int step = step0; step < step0 + nsteps; ++step) {
        RUN(caphn_embedding_gather(R, E, p->embed_w, sa.idx, ws + w.Xe, s));
        RUN(caphn_gemm_f32(0, 1, R, GH, E, ws + w.Xe, E, p->w_ih, EF, ws + w.Xg, GH, p->b_ih, nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
        RecFwdArgs a;
        a.B = R; a.T = 1; a.P = P; a.H = H; a.RG = RG; a.slab_div = c->beam;
        a.Waf = ws + w.Waf; a.G = ws + w.G; a.Xg = ws + w.Xg; a.h0 = sa.h_cur; a.c0 = nullptr;
        a.W_hh = p->w_hh; a.b_hh = p->b_hh; a.U_a = p->Ua_w; a.b_Ua = p->Ua_b; a.v_a = p->va_w; a.b_va = p->va_b;
        a.Hs = ws + w.Hs; a.Hprev = ws + w.Hprev; a.alphas = ws + w.alphas; a.gates = ws + w.gates; a.hn = ws + w.hn;
        a.Cs = nullptr; a.Cprev = nullptr; a.uah = ws + w.uah; a.prof = nullptr; a.rotate = 0;
        a.vecW = (H % 4 == 0) && caphn_aligned16(p->w_hh) && caphn_aligned16(p->Ua_w);
        a.vecS = (H % 4 == 0) && caphn_aligned16(ws);
        if (w.L1 > 0) a.Hs = ws + w.H0s;
        RUN(caphn_launch_rec_fwd(a, false, s));
        if (w.L1 > 0) RUN(layers_forward(d, p, w, ws, 1, ws + w.H0s, H, ws + w.Hs, H, s));
        RUN(caphn_gemm_f32(0, 1, R, V, H, ws + w.Hs, H, p->out_w, H, logits, V, p->out_b, nullptr, 0, CAPHN_GEMM_BIAS, 1, s));
        if (alphas)
            if (hipMemcpy2DAsync(alphas + (size_t)(step - 1) * P, sizeof(float) * (size_t)c->max_steps * P, ws + w.alphas,
                                 sizeof(float) * P, sizeof(float) * P, R, hipMemcpyDeviceToDevice, s) != hipSuccess) return CAPHN_ELAUNCH;
        RUN(caphn_launch_row_topk(R, V, c->beam, logits, sa.score, const_cast<float*>(sa.cand_val), const_cast<int*>(sa.cand_idx), s));
        RUN(caphn_launch_beam_merge(sa, step, s));
    }
    return caphn_launch_status();
}

extern "C" int caphn_decoder_search_result(const caphn_decoder_dims* d, const caphn_search_cfg* c, int steps_done, void* ws_, void* sws_,
                                           int64_t* seqs, int* lengths, float* scores, int* finished, int* n_active,
                                           caphn_stream_t stream) {
    if (!search_ok(d, c) || !ws_ || !sws_ || !seqs || !lengths || !scores || !finished || !n_active || steps_done < 0 ||
        steps_done > c->max_steps) return CAPHN_EINVAL;
    const Ws w = layout(d);
    const SearchArgs sa = search_args(d, c, w, static_cast<float*>(ws_), static_cast<char*>(sws_));
    return caphn_launch_search_result(sa, steps_done, seqs, lengths, scores, finished, n_active, static_cast<hipStream_t>(stream));
}

// nn.Linear's parameter gradients in one pass: dW [M, N] = dY^T X, db [M] = column sums of dY (dY [K, M], X [K, N]).  The bias
// gradient rides in the weight-gradient GEMM's n-tile-0 workgroups (it is the column sum of an operand that is staged anyway).
extern "C" int caphn_linear_wgrad_f32(int M, int N, int K, const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw,
                                      float* db, void* ws, caphn_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !dY || !X || !dW || !db || !ws) return CAPHN_EINVAL;
    return caphn_gemm_tn_colsum(M, N, K, dY, ldy, X, ldx, dW, ldw, db, pick_splitk(M, N, K), nullptr, ws, false,
                                static_cast<hipStream_t>(stream));
}
