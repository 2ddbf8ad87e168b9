// Device-resident beam / greedy decoding state machine (inference side of the hot path).
//
// Reference: beam search in HyperNet.test_step (hypernet_attention.py:251-306, k = 3, one image at a time, host
// loop with .item()/list bookkeeping every step) and AttentionGru.greedy_search (models/decoderlstm.py:138-175).
// Here every image of a batch carries `beam` rows; all bookkeeping (top-k over beam*V scores, beam re-ordering of
// h, completed-sequence lists, shrinking beam width) lives in HBM and is advanced by two small kernels per step,
// so the host never reads a token while decoding.  Greedy search is the beam = 1 case with the start token's
// embedding looked up instead of zeroed.
#include "common.h"
#include "decoder_internal.h"
#include <math.h>

namespace {

// one workgroup per row: log-softmax statistics and the row's `k` best (score + log p) candidates, best first;
// ties go to the lower vocabulary index
__global__ __launch_bounds__(256) void row_topk_kernel(int V, int k, const float* __restrict__ logits,
                                                       const float* __restrict__ score,
                                                       float* __restrict__ cand_val, int* __restrict__ cand_idx) {
    __shared__ float red[4];
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int chosen[SEARCH_MAX_BEAM];
    __shared__ float stat[2];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)r * V;
    float m = -INFINITY;
    for (int v = tid; v < V; v += 256) m = fmaxf(m, x[v]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int v = tid; v < V; v += 256) sum += expf(x[v] - m);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (tid == 0) { stat[0] = m; stat[1] = logf(red[0] + red[1] + red[2] + red[3]); }
    __syncthreads();
    const float lse = stat[1], base = score[r];
    for (int j = 0; j < k; ++j) {
        float best = -INFINITY; int besti = 0x7fffffff;
        for (int v = tid; v < V; v += 256) {
            bool taken = false;
            for (int q = 0; q < j; ++q) taken |= (chosen[q] == v);
            const float xv = x[v];
            if (!taken && (xv > best || (xv == best && v < besti))) { best = xv; besti = v; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(besti, o, 64);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
        }
        if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < 4; ++i) if (bv[i] > best || (bv[i] == best && bi[i] < besti)) { best = bv[i]; besti = bi[i]; }
            chosen[j] = besti;
            // scores = top_k_scores.expand_as(scores) + log_softmax(scores)        hypernet_attention.py:270-272
            cand_val[(size_t)r * k + j] = besti < V ? base + ((best - m) - lse) : -INFINITY;
            cand_idx[(size_t)r * k + j] = besti < V ? besti : 0;
        }
        __syncthreads();
    }
}

// one workgroup per image: k-way merge of the alive rows' candidate lists, then the reference's bookkeeping
// (hypernet_attention.py:274-299): extend sequences, set completed ones aside, shrink the beam, re-order h
__global__ __launch_bounds__(256) void beam_merge_kernel(SearchArgs a, int step, int first) {
    __shared__ int sel_row[SEARCH_MAX_BEAM], sel_word[SEARCH_MAX_BEAM], dst[SEARCH_MAX_BEAM];   // dst >= 0 alive slot, < 0: -(complete slot + 1)
    __shared__ float sel_val[SEARCH_MAX_BEAM];
    __shared__ int ka_s, ka_new_s, zero_s;
    const int n = blockIdx.x, tid = threadIdx.x, k = a.beam, L = a.max_steps + 1, H = a.H;
    const int base = n * k;
    if (tid == 0) {
        const int ka = a.k_alive[n];
        ka_s = ka;
        int heads[SEARCH_MAX_BEAM];
        for (int r = 0; r < k; ++r) heads[r] = 0;
        const int nrows = first ? 1 : ka;           // step 1: all rows are identical, only row 0 is ranked  (:274-275)
        int alive = 0, ncomp = a.n_comp[n];
        for (int j = 0; j < ka; ++j) {
            int br = 0; float bvv = -INFINITY; bool any = false;
            for (int r = 0; r < nrows; ++r) {
                if (heads[r] >= k) continue;
                const float v = a.cand_val[(size_t)(base + r) * k + heads[r]];
                if (!any || v > bvv) { bvv = v; br = r; any = true; }
            }
            sel_row[j] = br; sel_val[j] = bvv; sel_word[j] = a.cand_idx[(size_t)(base + br) * k + heads[br]];
            heads[br]++;
            if ((int64_t)sel_word[j] == a.end_token) {
                dst[j] = -(ncomp + 1);
                a.comp_score[(size_t)n * k + ncomp] = bvv; a.comp_len[(size_t)n * k + ncomp] = step + 1;
                ++ncomp;
            } else dst[j] = alive++;
        }
        a.n_comp[n] = ncomp;
        a.k_alive[n] = alive;
        ka_new_s = alive;
        // `if k_prev_words[0][0] == 0: embeddings[...] = 0` (:263-264): a <pad> at the head of the beam zeroes every input
        int w0 = -1;
        for (int j = 0; j < ka; ++j) if (dst[j] == 0) w0 = sel_word[j];
        zero_s = a.zero_pad_rule && w0 == 0;
    }
    __syncthreads();
    const int ka = ka_s, ka_new = ka_new_s;
    if (ka == 0) return;                             // image finished earlier: its rows idle
    const int64_t* sin = a.seqs[(step + 1) & 1];      // sequences of length `step` written by the previous step
    int64_t* sout = a.seqs[step & 1];
    for (int j = 0; j < ka; ++j) {
        const int64_t* src = sin + (size_t)(base + sel_row[j]) * L;
        int64_t* out = dst[j] >= 0 ? sout + (size_t)(base + dst[j]) * L
                                   : a.comp_seqs + ((size_t)n * k + (-dst[j] - 1)) * L;
        for (int i = tid; i < step; i += 256) out[i] = src[i];
        if (tid == 0) out[step] = sel_word[j];
        if (dst[j] >= 0) {
            const float* hs = a.h_new + (size_t)(base + sel_row[j]) * H;
            float* hd = a.h_cur + (size_t)(base + dst[j]) * H;
            for (int i = tid; i < H; i += 256) hd[i] = hs[i];
            if (tid == 0) {
                a.score[base + dst[j]] = sel_val[j];
                a.idx[base + dst[j]] = zero_s ? (int64_t)-1 : (int64_t)sel_word[j];
            }
        }
    }
    // rows that left the beam keep finite, inert inputs
    for (int r = ka_new; r < k; ++r) {
        float* hd = a.h_cur + (size_t)(base + r) * H;
        for (int i = tid; i < H; i += 256) hd[i] = 0.f;
        if (tid == 0) { a.score[base + r] = 0.f; a.idx[base + r] = -1; }
    }
}

__global__ __launch_bounds__(256) void search_init_kernel(SearchArgs a, const float* __restrict__ h0, int64_t first_token, int lookup_first) {
    const int n = blockIdx.x, tid = threadIdx.x, k = a.beam, L = a.max_steps + 1, H = a.H;
    for (int r = 0; r < k; ++r) {
        for (int i = tid; i < H; i += 256) a.h_cur[(size_t)(n * k + r) * H + i] = h0[(size_t)n * H + i];
        if (tid == 0) {
            a.score[n * k + r] = 0.f;
            a.idx[n * k + r] = lookup_first ? first_token : (int64_t)-1;
            a.seqs[0][(size_t)(n * k + r) * L] = first_token;
        }
    }
    if (tid == 0) { a.k_alive[n] = k; a.n_comp[n] = 0; }
}

// per image: the completed sequence with the best score (first maximum, as list.index(max(...)) :311), or -- when the
// beam never emptied -- the head of the beam as it stands (greedy_search's max_sentence stop)
__global__ __launch_bounds__(256) void search_result_kernel(SearchArgs a, int steps_done, int64_t* __restrict__ out_seq, int* __restrict__ out_len,
                                                            float* __restrict__ out_score, int* __restrict__ finished, int* __restrict__ n_active) {
    __shared__ int src_s, len_s;
    const int n = blockIdx.x, tid = threadIdx.x, k = a.beam, L = a.max_steps + 1;
    if (tid == 0) {
        const int ka = a.k_alive[n], nc = a.n_comp[n];
        if (ka > 0) atomicAdd(n_active, 1);
        finished[n] = ka == 0;
        if (ka == 0 && nc > 0) {
            int bi = 0; float bvv = a.comp_score[(size_t)n * k];
            for (int c = 1; c < nc; ++c) if (a.comp_score[(size_t)n * k + c] > bvv) { bvv = a.comp_score[(size_t)n * k + c]; bi = c; }
            src_s = -(bi + 1); len_s = a.comp_len[(size_t)n * k + bi]; out_score[n] = bvv;
        } else {
            src_s = 0; len_s = steps_done + 1; out_score[n] = a.score[n * k];
        }
        out_len[n] = len_s;
    }
    __syncthreads();
    const int64_t* src = src_s < 0 ? a.comp_seqs + ((size_t)n * k + (-src_s - 1)) * L
                                   : a.seqs[steps_done & 1] + (size_t)(n * k) * L;
    for (int i = tid; i < L; i += 256) out_seq[(size_t)n * L + i] = i < len_s ? src[i] : 0;
}

}  // namespace

int caphn_launch_row_topk(int R, int V, int k, const float* logits, const float* score, float* cand_val, int* cand_idx, hipStream_t s) {
    if (k < 1 || k > SEARCH_MAX_BEAM) return CAPHN_EINVAL;
    hipLaunchKernelGGL(row_topk_kernel, dim3(R), dim3(256), 0, s, V, k, logits, score, cand_val, cand_idx);
    return caphn_launch_status();
}
int caphn_launch_beam_merge(const SearchArgs& a, int step, hipStream_t s) {
    hipLaunchKernelGGL(beam_merge_kernel, dim3(a.n_images), dim3(256), 0, s, a, step, step == 1 ? 1 : 0);
    return caphn_launch_status();
}
int caphn_launch_search_init(const SearchArgs& a, const float* h0, int64_t first_token, int lookup_first, hipStream_t s) {
    hipLaunchKernelGGL(search_init_kernel, dim3(a.n_images), dim3(256), 0, s, a, h0, first_token, lookup_first);
    return caphn_launch_status();
}
int caphn_launch_search_result(const SearchArgs& a, int steps_done, int64_t* out_seq, int* out_len, float* out_score, int* finished,
                               int* n_active, hipStream_t s) {
    if (hipMemsetAsync(n_active, 0, sizeof(int), s) != hipSuccess) return CAPHN_ELAUNCH;
    hipLaunchKernelGGL(search_result_kernel, dim3(a.n_images), dim3(256), 0, s, a, steps_done, out_seq, out_len, out_score, finished, n_active);
    return caphn_launch_status();
}
