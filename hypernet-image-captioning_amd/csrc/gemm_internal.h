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
    // optional fused bias gradient of a weight-gradient GEMM (ta = 1, A stored [K, M]): colsum_a[m] += sum_k A[k][m],
    // accumulated by the n-tile-0 workgroups from the fp32 tiles they stage anyway (atomics; the caller zero-fills)
    float* colsum_a;
    int kmap_lds;        // map_mode 2: ints of LDS reserved behind the tiles for this workgroup's slice of the K map (0: none)
};

// internal entry (decoder.hip): caphn_gemm_f32 plus the row subset
int caphn_gemm_mapped(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                      float* C, int ldc, const float* bias, int flags, int splitk,
                      const int* row_map, const int* dev_count, int map_mode, hipStream_t s);

// C = A^T B (+ split-K) with db = column sums of A fused in when the split-bf16 back end is active; falls back to the
// separate column-sum kernel otherwise.  Zero-fills C (when splitting) and db itself.  cws: caphn_colsum workspace.
int caphn_gemm_tn_colsum(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         float* colsum_out, int splitk, const int* rowmap, void* cws, bool prezeroed, hipStream_t s);

// split-bf16 back end (gemm_bf16x3.hip): BK = 32
int caphn_gemm_bf16x3_launch(GemmArgs g, int ta, int tb, hipStream_t s);
extern int g_tune_gemm;     // 0: fp32 MFMA, 1: split-bf16 MFMA
