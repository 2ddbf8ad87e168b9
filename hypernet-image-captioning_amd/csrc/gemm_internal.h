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
    // optional PRE-SPLIT operands (both or neither): the three bf16 planes hi / mid / lo of A and B (x == hi + mid + lo
    // exactly, caphn_split3_launch), each plane a matrix with A's / B's logical layout, leading dimension ldap / ldbp
    // (elements, % 8 == 0) and psa / psb elements between planes.  The kernel then stages bf16 straight into LDS: no
    // split arithmetic in the main loop (it was 4.5 vector instructions per element per USE of a tile).
    const void* Ap; int ldap; size_t psa;
    const void* Bp; int ldbp; size_t psb;
};

// A GEMM operand as the decoder composites hand it around: the fp32 matrix and, when some producer made them, its planes.
struct Opnd {
    const float* f; int ld;
    const void* p; int ldp; size_t ps;
    Opnd() : f(nullptr), ld(0), p(nullptr), ldp(0), ps(0) {}
    Opnd(const float* f_, int ld_) : f(f_), ld(ld_), p(nullptr), ldp(0), ps(0) {}
    Opnd(const float* f_, int ld_, const void* p_, int ldp_, size_t ps_) : f(f_), ld(ld_), p(p_), ldp(ldp_), ps(ps_) {}
    // sub-matrix starting `cols` columns (and `rows` rows) in
    Opnd at(size_t rows, size_t cols) const {
        Opnd o = *this;
        o.f = f + rows * ld + cols;
        if (p) o.p = static_cast<const char*>(p) + 2 * (rows * ldp + cols);
        return o;
    }
};
// C = op(A) op(B) as caphn_gemm_f32 / caphn_gemm_mapped / caphn_gemm_tn_colsum, with operands that may carry planes
// (planes are used when BOTH have them and the shapes allow 16-byte plane loads; otherwise the fp32 matrices are).
struct GemmX {
    const float* bias = nullptr; const float* mask = nullptr; int ldmask = 0; int flags = 0; int splitk = 1;
    const int* rowmap = nullptr;    // caphn_decoder_prepare_rows map ([0] = count, +4 = indices) ...
    int map_mode = 0;               // ... applied to M (1) or K (2)
    float* colsum = nullptr;        // ta = 1 only: += column sums of A (bias gradient)
    int Kp = 0;                     // when the planes are used: K rounded up to 8, both operands' planes hold zeros there
};
int caphn_gemm_x(int ta, int tb, int M, int N, int K, const Opnd& A, const Opnd& B, float* C, int ldc, const GemmX& x, hipStream_t s);

// planes[0..3) = hi / mid / lo of src (rows x cols, leading dimensions ld / ldp; planes `ps` elements apart), several
// matrices per launch.  rows_dev (optional): device int, only that many leading rows are live (mapped GEMM operands)
struct SplitJob { const float* src; int ld; void* dst; int ldp; size_t ps; int rows, cols; int zero_rows; };   // zero_rows: rows of zeros appended
int caphn_split3_launch(const SplitJob* jobs, int n, hipStream_t s);

// internal entry (decoder.hip): caphn_gemm_f32 plus the row subset
int caphn_gemm_mapped(int ta, int tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                      float* C, int ldc, const float* bias, int flags, int splitk,
                      const int* row_map, const int* dev_count, int map_mode, hipStream_t s);

// C = A^T B (+ split-K) with db = column sums of A fused in when the split-bf16 back end is active; falls back to the
// separate column-sum kernel otherwise.  Zero-fills C (when splitting) and db itself.  cws: caphn_colsum workspace.
int caphn_gemm_tn_colsum(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                         float* colsum_out, int splitk, const int* rowmap, void* cws, bool prezeroed, hipStream_t s);

bool caphn_gemm_planes_ok(const GemmArgs& g, int ta, int tb);
// split-bf16 back end (gemm_bf16x3.hip): BK = 32
int caphn_gemm_bf16x3_launch(GemmArgs g, int ta, int tb, hipStream_t s);
// K-resident NT kernel for short contractions (gemm_kres.hip): CAPHN_OK = launched, 1 = not applicable, else an error
int caphn_gemm_kres_launch(const GemmArgs& g, int ta, int tb, hipStream_t s);
extern int g_tune_gemm;     // 0: fp32 MFMA, 1: split-bf16 MFMA
