// Shared device/host helpers for libcaphn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/caphn.h"

#define CAPHN_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// status.hip: the current device's sticky failure word (include/caphn.h, caphn_device_error)
int caphn_sticky_error();               // CAPHN_OK / CAPHN_ETIMEOUT, one plain host read
int* caphn_errword();                   // device-visible address of the word (allocated on first use; nullptr if that failed)
extern long long g_tune_xch_timeout;    // hand-off time bound of the pair recurrent kernels, in 100 MHz wall-clock ticks

static inline int caphn_launch_status() {
    if (hipGetLastError() != hipSuccess) return CAPHN_ELAUNCH;
    return caphn_sticky_error();
}
static inline bool caphn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__device__ __forceinline__ bool caphn_aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline size_t caphn_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device math: transcendental forms with absolute error ~1e-7 (v_exp_f32 / v_rcp_f32) ----
__device__ __forceinline__ float caphn_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float caphn_sigmoid(float x) {
    // 1/(1+e^-x); for x << 0 e^-x overflows to inf -> 0, for x >> 0 -> 1
    return __builtin_amdgcn_rcpf(1.0f + caphn_exp(-x));
}
__device__ __forceinline__ float caphn_tanh(float x) {
    // (1-e)/(1+e), e = exp(-2|x|) in (0,1]: no overflow, |err| ~ 1e-7
    float ax = __builtin_fabsf(x);
    float e = __builtin_amdgcn_exp2f(ax * -2.88539008177792681f);
    float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return __builtin_copysignf(t, x);
}

// ---- dropout: counter-based keep decision (splitmix64 of seed + index), so the backward pass recomputes the forward's mask
// from (seed, index) instead of storing it.  Returns 0 (dropped, probability p) or 1 / (1 - p).
__device__ __forceinline__ float caphn_keep_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);          // 24 uniform bits in [0, 1)
    return u < p ? 0.f : inv_keep;
}

// ---- sums over aligned groups of 8 lanes on the DPP path (no LDS round trip: __shfl_xor compiles to ds_bpermute_b32, ~100 cycles
// of dependent latency per stage -- three stages per row were the larger part of a register-resident mat-vec sweep).
// Pairing: lanes l ^ 1 (quad_perm [1,0,3,2]), l ^ 2 (quad_perm [2,3,0,1]), then l <-> 7 - l inside each half row
// (row_half_mirror: the partner sits in the other quad of the group); every lane of a group ends with the group's total.
__device__ __forceinline__ float caphn_dpp(float v, int ctrl_b1_4e_141) {
    // (ctrl must be a literal: dispatched below)
    const int x = __builtin_bit_cast(int, v);
    int y;
    if (ctrl_b1_4e_141 == 0xB1) y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
    else if (ctrl_b1_4e_141 == 0x4E) y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
    else y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);
    return __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float group8_sum(float v) {
    v += caphn_dpp(v, 0xB1);
    v += caphn_dpp(v, 0x4E);
    v += caphn_dpp(v, 0x141);
    return v;
}

// ---- wave reductions (64 lanes) ----
// On the DPP / lane-permute path (no ds_bpermute: six dependent LDS round trips per reduction were ~700 cycles): pairs and quads
// by quad_perm, the two quads of a half row by row_half_mirror, the two halves of a 16-lane row by row_mirror, odd/even rows by
// v_permlane16_swap, the wave halves by v_permlane32_swap (given the same register twice, the two results are "mine" and
// "the other side's" in every lane).  Every lane ends with the total.
__device__ __forceinline__ float caphn_dpp_row_mirror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = group8_sum(v);
    v += caphn_dpp_row_mirror(v);
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
        v = __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
        v = __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
    }
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, caphn_dpp(v, 0xB1));
    v = fmaxf(v, caphn_dpp(v, 0x4E));
    v = fmaxf(v, caphn_dpp(v, 0x141));
    v = fmaxf(v, caphn_dpp_row_mirror(v));
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
        v = fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
        v = fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
    }
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
