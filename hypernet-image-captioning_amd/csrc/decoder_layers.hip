// The extra GRUCells of AttentionGru(num_layers > 1)   (/root/reference/models/decoderlstm.py:34-36).
//
// Every application is h = layer(h, h) -- input and state are the same vector (:65-67 on the initial state, :101-103 at
// every time step) -- so one cell costs two [B,H] x [H,3H] GEMMs (launched by the composite in decoder.hip) and the
// pointwise kernel below.  Activations are kept per SLOT: slot 0 is the application to init_hidden's output, slot t + 1
// the one at time step t; a caption owns S = T + 1 consecutive slots ([B,S,.] arrays), so the layer weight gradients
// are one transposed GEMM over all B S rows after the time loop.
#include "decoder_internal.h"

namespace {

// PyTorch GRUCell: r = sig(gi_r + gh_r), z = sig(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h
__global__ __launch_bounds__(256) void layer_gru_fwd_kernel(LayerFwdArgs a) {
    const int H = a.H, G3 = 3 * a.H;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.B * H) return;
    const int b = i / H, k = i - b * H;
    const float* gi = a.gi + (size_t)b * G3;
    const float* gh = a.gh + (size_t)b * G3;
    const float hp = a.hin[(size_t)b * a.hin_ld + k];
    const float r = caphn_sigmoid(gi[k] + gh[k]);
    const float z = caphn_sigmoid(gi[H + k] + gh[H + k]);
    const float hnv = gh[2 * H + k];
    const float n = caphn_tanh(gi[2 * H + k] + r * hnv);
    float hnew = (1.0f - z) * n + z * hp;
    const size_t row = (size_t)b * a.S + a.slot;
    a.sgates[row * G3 + k] = r; a.sgates[row * G3 + H + k] = z; a.sgates[row * G3 + 2 * H + k] = n;
    a.shn[row * H + k] = hnv;
    a.sin[row * H + k] = hp;
    if (a.drop_p > 0.f)      // h = self.drop(h) after the LAST layer (:104): same element index as the one-layer kernels
        hnew *= caphn_keep_scale(a.drop_seed, ((unsigned long long)b * a.T + a.t) * H + k, a.drop_p, 1.0f / (1.0f - a.drop_p));
    a.hout[(size_t)b * a.hout_ld + k] = hnew;
}

// dh arrives as d1 (+ d2).  Emits d gi, d gh of this slot and the direct path dh z; the caller adds d gi W_ih + d gh W_hh.
__global__ __launch_bounds__(256) void layer_gru_bwd_kernel(LayerBwdArgs a) {
    const int H = a.H, G3 = 3 * a.H;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.B * H) return;
    const int b = i / H, k = i - b * H;
    float dh = a.d1[(size_t)b * a.d1_ld + k];
    if (a.d2) dh += a.d2[(size_t)b * H + k];
    if (a.drop_p > 0.f)
        dh *= caphn_keep_scale(a.drop_seed, ((unsigned long long)b * a.T + a.t) * H + k, a.drop_p, 1.0f / (1.0f - a.drop_p));
    const size_t row = (size_t)b * a.S + a.slot;
    const float r = a.sgates[row * G3 + k], z = a.sgates[row * G3 + H + k], n = a.sgates[row * G3 + 2 * H + k];
    const float hnv = a.shn[row * H + k], hp = a.sin[row * H + k];
    const float dn = dh * (1.0f - z) * (1.0f - n * n);
    const float dz = dh * (hp - n) * z * (1.0f - z);
    const float dr = dn * hnv * r * (1.0f - r);
    a.dgi[row * G3 + k] = dr; a.dgi[row * G3 + H + k] = dz; a.dgi[row * G3 + 2 * H + k] = dn;
    a.dgh[row * G3 + k] = dr; a.dgh[row * G3 + H + k] = dz; a.dgh[row * G3 + 2 * H + k] = dn * r;
    a.dout[(size_t)b * a.dout_ld + k] = dh * z;
}

}  // namespace

int caphn_launch_layer_gru_fwd(const LayerFwdArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.H <= 0 || a.slot < 0 || a.slot >= a.S) return CAPHN_EINVAL;
    hipLaunchKernelGGL(layer_gru_fwd_kernel, dim3((a.B * a.H + 255) / 256), dim3(256), 0, s, a);
    return CAPHN_OK;
}

int caphn_launch_layer_gru_bwd(const LayerBwdArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.H <= 0 || a.slot < 0 || a.slot >= a.S) return CAPHN_EINVAL;
    hipLaunchKernelGGL(layer_gru_bwd_kernel, dim3((a.B * a.H + 255) / 256), dim3(256), 0, s, a);
    return CAPHN_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Stand-alone BahdanauAttention.forward (/root/reference/models/attention.py:21-46) behind its two Linear layers, for callers
// that step a decoder by hand (inside AttentionGru's loops the same arithmetic is fused into the recurrent kernels).
//   e_p = v_a . tanh(Waf_p + uah) + b_va,  alpha = softmax_p(e),  ctx = sum_p alpha_p f_p
// One workgroup per caption: a wave per position for the scores, wave 0 the softmax, a thread per feature column the context.
namespace {
constexpr int BNT = 256;

__global__ __launch_bounds__(BNT) void bahdanau_fwd_kernel(int P, int F, int H, const float* __restrict__ f, const float* __restrict__ Waf,
                                                          const float* __restrict__ uah, const float* __restrict__ v_a,
                                                          const float* __restrict__ b_va, float* __restrict__ ctx,
                                                          float* __restrict__ alpha) {
    extern __shared__ float bl[];              // [P] scores / weights
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* u = uah + (size_t)b * H;
    for (int p = wave; p < P; p += BNT / 64) {
        const float* wr = Waf + ((size_t)b * P + p) * H;
        float s = 0.f;
        for (int k = lane; k < H; k += 64) s += v_a[k] * caphn_tanh(wr[k] + u[k]);
        s = wave_sum(s);
        if (lane == 0) bl[p] = s + b_va[0];
    }
    __syncthreads();
    if (wave == 0) {
        float mx = -INFINITY;
        for (int p = lane; p < P; p += 64) mx = fmaxf(mx, bl[p]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int p = lane; p < P; p += 64) { const float e = caphn_exp(bl[p] - mx); bl[p] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int p = lane; p < P; p += 64) { const float a = bl[p] * inv; bl[p] = a; alpha[(size_t)b * P + p] = a; }
    }
    __syncthreads();
    for (int k = tid; k < F; k += BNT) {
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += bl[p] * f[((size_t)b * P + p) * F + k];
        ctx[(size_t)b * F + k] = s;
    }
}

// Gradients of the above: dctx [B,F] and (optional) dalpha [B,P] in; dWaf [B,P,H], duah [B,H], the per-caption partials of
// d v_a / d b_va (part [B, H+1], column-summed by the caller) and the direct path df = alpha_p dctx [B,P,F] out.
__global__ __launch_bounds__(BNT) void bahdanau_bwd_kernel(int P, int F, int H, const float* __restrict__ f, const float* __restrict__ Waf,
                                                          const float* __restrict__ uah, const float* __restrict__ v_a,
                                                          const float* __restrict__ alpha, const float* __restrict__ dctx,
                                                          const float* __restrict__ dalpha, float* __restrict__ dWaf,
                                                          float* __restrict__ duah, float* __restrict__ part, float* __restrict__ df) {
    extern __shared__ float bl[];              // [P] d alpha -> d e, then [P] alpha
    float* de_s = bl;
    float* al_s = bl + P;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* dc = dctx + (size_t)b * F;
    for (int p = wave; p < P; p += BNT / 64) {
        const float* fr = f + ((size_t)b * P + p) * F;
        float s = 0.f;
        for (int k = lane; k < F; k += 64) s += fr[k] * dc[k];
        s = wave_sum(s);
        if (lane == 0) { de_s[p] = s + (dalpha ? dalpha[(size_t)b * P + p] : 0.f); al_s[p] = alpha[(size_t)b * P + p]; }
    }
    __syncthreads();
    if (wave == 0) {                            // softmax backward: de_p = alpha_p (dalpha_p - sum_q alpha_q dalpha_q)
        float dot = 0.f;
        for (int p = lane; p < P; p += 64) dot += al_s[p] * de_s[p];
        dot = wave_sum(dot);
        float sde = 0.f;
        for (int p = lane; p < P; p += 64) { const float de = al_s[p] * (de_s[p] - dot); de_s[p] = de; sde += de; }
        sde = wave_sum(sde);
        if (lane == 0) part[(size_t)b * (H + 1) + H] = sde;
    }
    __syncthreads();
    for (int k = tid; k < H; k += BNT) {
        const float u = uah[(size_t)b * H + k], vk = v_a[k];
        float su = 0.f, sv = 0.f;
        for (int p = 0; p < P; ++p) {
            const float tv = caphn_tanh(Waf[((size_t)b * P + p) * H + k] + u);
            const float w = de_s[p] * (1.0f - tv * tv) * vk;
            dWaf[((size_t)b * P + p) * H + k] = w;
            su += w; sv += de_s[p] * tv;
        }
        duah[(size_t)b * H + k] = su;
        part[(size_t)b * (H + 1) + k] = sv;
    }
    if (df)
        for (int i = tid; i < P * F; i += BNT) {
            const int p = i / F, k = i - p * F;
            df[(size_t)b * P * F + i] = al_s[p] * dc[k];
        }
}
}  // namespace

extern "C" int caphn_bahdanau_fwd(int B, int P, int F, int H, const float* f, const float* Waf, const float* uah, const float* v_a,
                                  const float* b_va, float* ctx, float* alpha, caphn_stream_t stream) {
    if (B <= 0 || P <= 0 || F <= 0 || H <= 0 || P > 8192 || !f || !Waf || !uah || !v_a || !b_va || !ctx || !alpha) return CAPHN_EINVAL;
    hipLaunchKernelGGL(bahdanau_fwd_kernel, dim3(B), dim3(BNT), sizeof(float) * P, static_cast<hipStream_t>(stream), P, F, H, f, Waf, uah,
                       v_a, b_va, ctx, alpha);
    return caphn_launch_status();
}

extern "C" int caphn_bahdanau_bwd(int B, int P, int F, int H, const float* f, const float* Waf, const float* uah, const float* v_a,
                                  const float* alpha, const float* dctx, const float* dalpha, float* dWaf, float* duah, float* part,
                                  float* df, caphn_stream_t stream) {
    if (B <= 0 || P <= 0 || F <= 0 || H <= 0 || P > 4096 || !f || !Waf || !uah || !v_a || !alpha || !dctx || !dWaf || !duah || !part)
        return CAPHN_EINVAL;
    hipLaunchKernelGGL(bahdanau_bwd_kernel, dim3(B), dim3(BNT), sizeof(float) * 2 * P, static_cast<hipStream_t>(stream), P, F, H, f, Waf,
                       uah, v_a, alpha, dctx, dalpha, dWaf, duah, part, df);
    return caphn_launch_status();
}
