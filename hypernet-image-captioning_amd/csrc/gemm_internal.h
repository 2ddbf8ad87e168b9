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
    // optional row subset (the loss ignores <pad> targets: those rows of logits / d logits are dead work)
    //   map_mode 1: the M index is logical; A row and C row m live at physical row row_map[m]; M_eff = *dev_count
    //   map_mode 2: the K index is logical (weight gradients): both operands' row k lives at row_map[k]; K_eff = *dev_count
    const int* row_map; const int* dev_count; int map_mode;
};

// internal entry (decoder.hip): caphn_gemm_f32 plus the row subset
int caphn_gemm_mapped(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                      float* C, int ldc, const float* bias, int flags, int splitk,
                      const int* row_map, const int* dev_count, int map_mode, hipStream_t s);

// split-bf16 back end (gemm_bf16x3.hip): BK = 32
int caphn_gemm_bf16x3_launch(GemmArgs g, int ta, int tb, hipStream_t s);
extern int g_tune_gemm;     // 0: fp32 MFMA, 1: split-bf16 MFMA
