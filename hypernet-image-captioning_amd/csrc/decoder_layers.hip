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
