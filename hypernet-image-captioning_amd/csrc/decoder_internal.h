// Internal (non-ABI) interfaces between the decoder composite and its kernels.
#pragma once
#include "common.h"

// Recurrent forward: GRU (3 gates r,z,n) or LSTM (4 gates i,f,g,o) cell with additive attention.
struct RecFwdArgs {
    int B, T, P, H;
    int RG;             // gate slabs of G kept LDS-resident (the rest stream from L2)
    const float* Waf;   // [B,P,H]    W_a f + b_Wa
    const float* G;     // [B,P,NG*H] f W_ih[:,E:]^T
    const float* Xg;    // [B,T,NG*H] x_t W_ih[:,:E]^T + b_ih
    const float* h0;    // [B,H]
    const float* c0;    // [B,H]      (LSTM)
    const float* W_hh; const float* b_hh;   // [NG*H,H],[NG*H]
    const float* U_a; const float* b_Ua;    // [H,H],[H]
    const float* v_a; const float* b_va;    // [H],[1]
    float* Hs; float* Hprev;                // [B,T,H] h_t, h_{t-1}
    float* alphas;                          // [B,T,P]
    float* gates;                           // [B,T,NG*H] post-activation gates
    float* hn;                              // [B,T,H]  W_hn h + b_hn          (GRU)
    float* Cs; float* Cprev;                // [B,T,H]  c_t, c_{t-1}           (LSTM)
    float* uah;                             // [B,T,H]  U_a h + b_Ua
    unsigned long long* prof;               // 8 counters: per-phase shader-clock sums of workgroup 0 (tuning aid)
    int vecW, vecS, rotate;
    int slab_div = 1;   // rows b share the image slab b / slab_div of Waf and G (beam search: beams of one image)
    // two-workgroups-per-caption kernels (recurrent_pair.hip)
    unsigned long long* xch = nullptr;   // exchange area, caphn_rec_pair_xch_bytes, zero at launch
    const float* WP = nullptr; int wp_pitch = 0;   // [U_a; W_hh] packed to a 128-byte-aligned row pitch (caphn_launch_rec_pair_prep)
    int t0 = 0, t1 = 0;                  // time-step window [t0, t1) of this launch (t1 == 0: T); t0 > 0 continues from Hs / Cs
    float drop_p = 0.f; unsigned long long drop_seed = 0;   // dropout on h_t: element (b, t, k) keeps by the hash of (seed, (b T + t) H + k)
    int waf_lds = 0;                     // pair forward kernel: its columns of W_a f live in LDS (set by the launcher)
    int wc_rows = 0;                     // pair forward kernel: weight rows cached in LDS (set by the launcher; -1: no on-chip rows)
    unsigned epoch = 0; int* err = nullptr; long long xlimit = 0;   // pair kernels (set by the launcher): tag epoch of this launch, the
                                         // device's sticky failure word, hand-off time bound in 100 MHz wall-clock ticks
    const float* Hsrc = nullptr;         // t0 > 0: h_{t0-1} comes from Hsrc [B,T,H] instead of Hs (multi-layer decoders: Hs receives the
                                         // attention cell's output, the next step continues from the LAST layer's output)
};
struct RecBwdArgs {
    int B, T, P, H;
    int RG;
    const float* Waf; const float* G;
    const float* W_hh; const float* U_a; const float* v_a;
    const float* Hprev; const float* alphas; const float* gates; const float* hn; const float* uah;
    const float* Cs; const float* Cprev;
    const float* dHs;       // [B,T,H] gradient arriving from the vocab projection
    const float* dalphas;   // [B,T,P] or null
    float* dgi; float* dgh; // [B,T,NG*H] (LSTM: dgh == dgi, one array)
    float* duah;            // [B,T,H]
    float* de;              // [B,T,P]
    float* dh0;             // [B,H]
    float* dc0;             // [B,H] (LSTM)
    unsigned long long* prof;
    int vecW, vecS, rotate;
    // attention parameter gradients accumulated across t inside the kernel (the tanh of the d(U_a h) phase is the same
    // one): dWaf [B,P,H] and the d v_a / d b_va partials part [B*ng, H+1] (ng = caphn_rec_bwd_groups(H)); null = not fused
    float* dWaf = nullptr; float* apart = nullptr;
    unsigned long long* xch = nullptr;   // pair kernels: exchange area (zero at launch)
    const float* WP = nullptr; int wp_pitch = 0;
    int apart_rows = 0;                  // pair kernels: rows of `apart` per caption
    float drop_p = 0.f; unsigned long long drop_seed = 0;
    unsigned epoch = 0; int* err = nullptr; long long xlimit = 0;   // pair kernels (set by the launcher), as in RecFwdArgs
    // pair kernels: init_hidden's backward in the kernel's epilogue -- each half leaves its part of d mean_f = dh0 W_inith (+ dc0
    // W_initc) over ITS k in dmean_part[half][B][F] (the two are added, half 0 first, by the df kernel); null = not wanted
    const float* inith_w = nullptr; const float* initc_w = nullptr; float* dmean_part = nullptr; int F = 0;
    int part_rows = 0, wc_rows = 0;      // pair backward kernel (set by the launcher): rows of its slice-partials array, weight rows kept in LDS
    int t0 = 0, t1 = 0;                  // time-step window [t0, t1), walked backwards (t1 == 0: T).  A window starts from dh = 0 (dc
                                         // from dc0 when t1 < T) and leaves dh_{t0-1} in dh0 (dc in dc0): the caller adds it to what
                                         // arrives at step t0 - 1
};
struct AttnGradArgs {
    int T, P, H, pchunk;
    const float* Waf; const float* uah; const float* de; const float* v_a;
    float* dWaf;            // [B,P,H]
    float* part;            // [B*npc, H+1]
};

// decoder_layers.hip: pointwise part of one extra GRUCell application h = layer(h, h) at one slot (0: initial state, t + 1: step t)
struct LayerFwdArgs {
    int B, H, S, slot;                   // S = T + 1 slots per caption in the saved arrays
    const float* gi; const float* gh;    // [B,3H] h W_ih^T + b_ih, h W_hh^T + b_hh
    const float* hin; int hin_ld;        // [B,H] rows at pitch hin_ld
    float* hout; int hout_ld;
    float* sin; float* sgates; float* shn;   // saved per slot: input [B,S,H], gates r,z,n [B,S,3H], W_hn h + b_hn [B,S,H]
    float drop_p = 0.f; unsigned long long drop_seed = 0; int T = 1, t = 0;   // dropout on the output (last layer, slot > 0)
};
struct LayerBwdArgs {
    int B, H, S, slot;
    const float* d1; int d1_ld; const float* d2;     // dh = d1 (+ d2 [B,H])
    const float* sin; const float* sgates; const float* shn;
    float* dgi; float* dgh;              // [B,S,3H]
    float* dout; int dout_ld;            // dh z (the caller accumulates d gi W_ih + d gh W_hh on top)
    float drop_p = 0.f; unsigned long long drop_seed = 0; int T = 1, t = 0;
};
int caphn_launch_layer_gru_fwd(const LayerFwdArgs& a, hipStream_t s);
int caphn_launch_layer_gru_bwd(const LayerBwdArgs& a, hipStream_t s);

size_t caphn_rec_fwd_lds_bytes(int P, int H, int NG, int RG);
size_t caphn_rec_bwd_lds_bytes(int P, int H, int NG, int RG);
int caphn_rec_resident_gates(int P, int H, int NG);       // largest RG whose fwd and bwd kernels fit 160 KB; -1 if none
int caphn_launch_rec_fwd(const RecFwdArgs& a, bool lstm, hipStream_t s);
int caphn_launch_rec_bwd(const RecBwdArgs& a, bool lstm, hipStream_t s);
// recurrent_pair.hip: the same loops with two workgroups per caption (B captions on 2 B CUs)
bool caphn_rec_pair_ok(int P, int H, int NG, int RG);
int caphn_rec_pair_resident_gates(int P, int H, int NG);
size_t caphn_rec_pair_xch_bytes(int B, int P, int H);
int caphn_rec_pair_bwd_groups(int P, int H);
int caphn_rec_pair_pitch(int H);
int caphn_rec_pair_half_a(int H);      // rows (and hidden indices) of the first half
size_t caphn_rec_pair_wp_floats(int H, int NG);
int caphn_launch_rec_pair_prep(unsigned long long* xch, size_t nxch, const float* U_a, const float* W_hh, int H, int NG, float* WP,
                               float* zbuf, size_t nz, hipStream_t s);      // zbuf (optional): nz floats zero-filled on the way
int caphn_launch_rec_pair_fwd(const RecFwdArgs& a, bool lstm, hipStream_t s);
int caphn_launch_rec_pair_bwd(const RecBwdArgs& a, bool lstm, hipStream_t s);
int caphn_rec_bwd_groups(int P, int H);     // > 0: the backward kernel can fuse the attention parameter gradients (rows of `apart` per caption)
int caphn_launch_attn_param_grads(const AttnGradArgs& a, int B, int npc, hipStream_t s);
int caphn_launch_ctx(int B, int T, int P, int F, const float* alphas, const float* f, float* ctx, int ldc, hipStream_t s);   // ctx rows at pitch ldc
int caphn_launch_df(int B, int T, int P, int F, const float* alphas, const float* dctx, const float* dmean, float* df, hipStream_t s,
                    const float* dmean2 = nullptr);      // dmean2: second partial of d mean_f (pair BPTT epilogue), added to the first
int caphn_launch_dmean(int B, int H, int F, const float* dh0, const float* Wh, const float* dc0, const float* Wc, float* out, hipStream_t s);
int caphn_launch_init_state(int B, int P, int F, int H, const float* f, const float* Wh, const float* bh, const float* Wc,
                            const float* bc, float* meanf, float* h0, float* c0, hipStream_t s);
int caphn_embedding_gather_strided(int rows, int E, const float* table, const int64_t* idx, int istride, float* out, int ostride, hipStream_t s);
int caphn_launch_mean_p(int B, int P, int F, const float* f, float* out, hipStream_t s);

// Beam / greedy search state (search.hip); all pointers are device memory inside the caller's search workspace.
constexpr int SEARCH_MAX_BEAM = 8;
struct SearchArgs {
    int n_images, beam, max_steps, H;
    int zero_pad_rule;          // hypernet_attention.py:263-264
    int64_t end_token;
    float* h_cur; const float* h_new;       // [n_images*beam, H]
    float* score;               // [rows] cumulative log-probability of each alive beam
    int64_t* idx;               // [rows] next input token (-1: zero vector)
    const float* cand_val; const int* cand_idx;   // [rows, beam]
    int* k_alive; int* n_comp;  // [n_images]
    int64_t* seqs[2];           // [rows, max_steps+1] ping-pong
    int64_t* comp_seqs; int* comp_len; float* comp_score;   // [n_images, beam, (max_steps+1)]
};
int caphn_launch_row_topk(int R, int V, int k, const float* logits, const float* score, float* cand_val, int* cand_idx, hipStream_t s);
int caphn_launch_beam_merge(const SearchArgs& a, int step, hipStream_t s);
int caphn_launch_search_init(const SearchArgs& a, const float* h0, int64_t first_token, int lookup_first, hipStream_t s);
int caphn_launch_search_result(const SearchArgs& a, int steps_done, int64_t* out_seq, int* out_len, float* out_score, int* finished,
                               int* n_active, hipStream_t s);
