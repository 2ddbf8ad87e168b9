// fp32 -> three bf16 planes (hi + mid + lo == x exactly), shared by the split-bf16 GEMM and the kernels that emit planes.
#pragma once
#include "common.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// The three planes by TRUNCATION on the bit pattern (exact: 24 significand bits = 8 + 8 + 8, so hi + mid + lo == x
// with no rounding anywhere): hi = x & 0xffff0000, mid = (x - hi) & 0xffff0000, lo = x - hi - mid.  Per element pair
// that is 4 v_and, 2 v_pk_add_f32 and 3 v_perm_b32 (packing the upper halves of two dwords) -- 4.5 vector
// instructions per element instead of the 7.5 of the convert / convert-back / subtract formulation, in a kernel
// whose main loop is bound by vector-instruction issue, not by the MFMA pipe (rocprofv3 SQ counters).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct Split4 { bf16x4 hi, mid, lo; };
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) {      // {a[31:16], b[31:16]} -> one dword, a low
    return __builtin_amdgcn_perm(b, a, 0x07060302u);
}
__device__ __forceinline__ Split4 split3(f32x4 v) {
    union U2 { f32x2 f; u32x2 u; };
    union Out { unsigned u[2]; bf16x4 b; };
    Out hi, mid, lo;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        U2 x, a, r1, m, r2;
        x.f = f32x2{v[2 * h], v[2 * h + 1]};
        a.u = x.u & 0xffff0000u;
        r1.f = x.f - a.f;
        m.u = r1.u & 0xffff0000u;
        r2.f = r1.f - m.f;
        hi.u[h] = pack_hi16(x.u[0], x.u[1]);
        mid.u[h] = pack_hi16(r1.u[0], r1.u[1]);
        lo.u[h] = pack_hi16(r2.u[0], r2.u[1]);
    }
    Split4 s;
    s.hi = hi.b; s.mid = mid.b; s.lo = lo.b;
    return s;
}

