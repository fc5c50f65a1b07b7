// Text -> shape retrieval on the device: float64 similarities, top-k and the rank of the ground truth.
//
// Replaces the numpy path of /root/reference/tricolo/evaluation/eval_retrieval.py:70-82 (np.dot on a float64 text matrix,
// np.argsort ascending, last k flipped) and the first-hit search behind MRR (:184-187) for one query per workgroup:
//   sims[s]   = sum_d (double)text[q][d] * (double)shape[s][d]                       (fp64 FMA, wave-shuffle reduction)
//   topk[q][j] = j-th largest sims, ties broken towards the HIGHER shape index        (= a stable ascending sort, flipped)
//   first_hit[q] = number of shapes ranked before the query's own shape (label[q])   (MRR = mean 1 / (first_hit + 1))
// Embeddings never leave the GPU; only Nq x k indices / similarities and Nq ranks go back to the host metrics.
#include "common.h"
#include "../../include/tricolo_hip.h"

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// (sim, index) order of the reference ranking: larger sim first, equal sims -> larger index first
__device__ __forceinline__ bool ranks_before(double sa, int ia, double sb, int ib) { return sa > sb || (sa == sb && ia > ib); }

__global__ __launch_bounds__(256) void retrieval_topk_kernel(const float* __restrict__ text, const float* __restrict__ shape,
                                                             const int* __restrict__ label, int Ns, int D, int k,
                                                             int* __restrict__ topk_idx, double* __restrict__ topk_sim,
                                                             int* __restrict__ first_hit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sims = (double*)smem;                                  // [Ns]
    double* tq = sims + Ns;                                        // [D] the query in float64
    __shared__ double wbest[4];
    __shared__ int wbi[4], wcnt[4];
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int d = t; d < D; d += 256) tq[d] = (double)text[(size_t)q * D + d];
    __syncthreads();
    for (int s = wave; s < Ns; s += 4) {                           // one wave per shape row: coalesced along D
        const float* row = shape + (size_t)s * D;
        double acc = 0.0;
        for (int d = lane; d < D; d += 64) acc = fma(tq[d], (double)row[d], acc);
        acc = wave_sum_f64(acc);
        if (lane == 0) sims[s] = acc;
    }
    __syncthreads();
    // rank of the query's own shape in the full ordering
    if (first_hit) {
        const int c = label[q];
        const double sc = sims[c];
        int cnt = 0;
        for (int s = t; s < Ns; s += 256) cnt += (s != c && ranks_before(sims[s], s, sc, c)) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) wcnt[wave] = cnt;
        __syncthreads();
        if (t == 0) first_hit[q] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    // k rounds of block-wide arg-best; the winner is retired by setting it to -inf
    for (int j = 0; j < k; ++j) {
        double best = -INFINITY;
        int bi = -1;
        for (int s = t; s < Ns; s += 256) {
            const double v = sims[s];
            if (bi < 0 || ranks_before(v, s, best, bi)) { best = v; bi = s; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (oi >= 0 && (bi < 0 || ranks_before(ob, oi, best, bi))) { best = ob; bi = oi; }
        }
        if (lane == 0) { wbest[wave] = best; wbi[wave] = bi; }
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < 4; ++w)
                if (wbi[w] >= 0 && (wbi[0] < 0 || ranks_before(wbest[w], wbi[w], wbest[0], wbi[0]))) { wbest[0] = wbest[w]; wbi[0] = wbi[w]; }
            topk_idx[(size_t)q * k + j] = wbi[0];
            topk_sim[(size_t)q * k + j] = wbest[0];
            if (wbi[0] >= 0) sims[wbi[0]] = -INFINITY;
        }
        __syncthreads();
    }
}

extern "C" int tri_retrieval_topk(const float* text, const float* shape, const int* label, int Nq, int Ns, int D, int k, int* topk_idx,
                                  double* topk_sim, int* first_hit, void* stream) {
    if (Nq <= 0) return TRI_OK;
    if (k < 1 || k > Ns) { tri_set_error("tri_retrieval_topk: need 1 <= k <= number of shapes"); return TRI_ERR_ARG; }
    size_t smem = ((size_t)Ns + D) * sizeof(double);
    if (smem > 150 * 1024) { tri_set_error("tri_retrieval_topk: Ns + D > 19200 does not fit the per-query LDS ranking"); return TRI_ERR_UNSUPPORTED; }
    static size_t attr = 0;
    if (smem > attr) {
        hipFuncSetAttribute((const void*)retrieval_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr = smem;
    }
    retrieval_topk_kernel<<<Nq, 256, smem, (hipStream_t)stream>>>(text, shape, label, Ns, D, k, topk_idx, topk_sim, first_hit);
    return tri_check_launch("tri_retrieval_topk");
}
