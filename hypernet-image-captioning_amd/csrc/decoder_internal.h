// Internal (non-ABI) interfaces between the decoder composite and its kernels.
#pragma once
#include "common.h"

struct GruFwdArgs {
    int B, T, P, H;
    const float* Waf;   // [B,P,H]   W_a f + b_Wa
    const float* G;     // [B,P,3H]  f W_ih[:,E:]^T
    const float* Xg;    // [B,T,3H]  x_t W_ih[:,:E]^T + b_ih
    const float* h0;    // [B,H]
    const float* W_hh; const float* b_hh;   // [3H,H],[3H]
    const float* U_a; const float* b_Ua;    // [H,H],[H]
    const float* v_a; const float* b_va;    // [H],[1]
    float* Hs; float* Hprev;                // [B,T,H] h_t, h_{t-1}
    float* alphas;                          // [B,T,P]
    float* gates;                           // [B,T,3H] r,z,n
    float* hn;                              // [B,T,H]  W_hn h + b_hn
    float* uah;                             // [B,T,H]  U_a h + b_Ua
    int vecW, vecS;
};
struct GruBwdArgs {
    int B, T, P, H;
    const float* Waf; const float* G;
    const float* W_hh; const float* U_a; const float* v_a;
    const float* Hprev; const float* alphas; const float* gates; const float* hn; const float* uah;
    const float* dHs;       // [B,T,H] gradient arriving from the vocab projection
    const float* dalphas;   // [B,T,P] or null
    float* dgi; float* dgh; // [B,T,3H]
    float* duah;            // [B,T,H]
    float* de;              // [B,T,P]
    float* dh0;             // [B,H]
    int vecW, vecS;
};
struct AttnGradArgs {
    int T, P, H, pchunk;
    const float* Waf; const float* uah; const float* de; const float* v_a;
    float* dWaf;            // [B,P,H]
    float* part;            // [B*npc, H+1]
};

size_t caphn_gru_fwd_lds_bytes(int P, int H);
size_t caphn_gru_bwd_lds_bytes(int P, int H);
int caphn_launch_gru_fwd(const GruFwdArgs& a, hipStream_t s);
int caphn_launch_gru_bwd(const GruBwdArgs& a, hipStream_t s);
int caphn_launch_attn_param_grads(const AttnGradArgs& a, int B, int npc, hipStream_t s);
int caphn_launch_ctx(int B, int T, int P, int F, const float* alphas, const float* f, float* ctx, hipStream_t s);
int caphn_launch_df(int B, int T, int P, int F, const float* alphas, const float* dctx, const float* dmean, float* df, hipStream_t s);
int caphn_launch_mean_p(int B, int P, int F, const float* f, float* out, hipStream_t s);
