// Shared between the two GEMM back ends (fp32 MFMA, split-bf16 MFMA).
#pragma once
#include "common.h"

struct GemmArgs {
    int M, N, K;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    const float* bias;
    const float* mask; int ldmask;
    int flags;
    int splitk, slabs_per_split;      // slabs of the back end's BK
    int vecA, vecB;
};

// split-bf16 back end (gemm_bf16x3.hip): BK = 32
int caphn_gemm_bf16x3_launch(GemmArgs g, int ta, int tb, hipStream_t s);
extern int g_tune_gemm;     // 0: fp32 MFMA, 1: split-bf16 MFMA
