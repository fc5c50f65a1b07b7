// HBM-bound passes of the TriCoLo towers on channels-last fp32 [M, C] tensors (gfx950): BatchNorm statistics /
// apply / backward, ReLU, residual add, the submanifold 2^3 max-pool with mask propagation, the ResNet stem
// 3x3/2 max-pool and the fused global-average-pool + per-shape view-max.
//
// Replaces: nn.BatchNorm1d + ReLU + spconv.SparseMaxPool3d (/root/reference/tricolo/model/module/voxel_encoder/
// sparse_cnn.py:13-35), torchvision BatchNorm2d / ReLU / MaxPool2d / AdaptiveAvgPool2d inside net_1
// (img_encoder/mv_cnn.py:20,29) and torch.max over views (mv_cnn.py:30-31).
// Every thread moves float4 (4 channels of one position): 16 B per lane, coalesced along C.
#include "common.h"
#include "../../include/tricolo_hip.h"

// ---------------------------------------------------------------------------------------------- BN statistics
// partial: [ntiles][2][C] (sum, sum of squares) written by the conv epilogue.  count from host (count_host > 0) or
// from a device counter (active voxel count).  Train-mode: biased variance for normalisation, unbiased for the
// running estimate, momentum 0.1, eps 1e-5 (torch.nn.BatchNorm semantics).  All reductions in double.
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int ntiles, int C, const int* __restrict__ count_dev,
                                   int count_host, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                                   float eps, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out) {
    // one channel per block, 256 record lanes: every load of the [ntiles][2][C] table is independent and in flight at
    // once (the 4-channel x 64-lane version spent 12-30 us in a dependent-load chain on the 1,536 - 6,144 record layers)
    __shared__ double ssum[4], ssq[4];
    const int c = blockIdx.x;
    // the finishing thread's operands are requested first: their latency then overlaps the record loads instead of
    // following the reduction (these 50 launches per step are pure latency on the step's critical path)
    float pg = 0.f, pb = 0.f, prm = 0.f, prv = 0.f;
    int pcount = count_host;
    if (threadIdx.x == 0) {
        pg = gamma[c]; pb = beta[c];
        if (running_mean) { prm = running_mean[c]; prv = running_var[c]; }
        if (count_dev) pcount = *count_dev;
    }
    double s = 0.0, q = 0.0;
#pragma unroll 8
    for (int tIdx = threadIdx.x; tIdx < ntiles; tIdx += 256) {
        s += (double)partial[((size_t)tIdx * 2 + 0) * C + c];
        q += (double)partial[((size_t)tIdx * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; ssq[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        q = (ssq[0] + ssq[1]) + (ssq[2] + ssq[3]);
        double n = (double)pcount;
        if (n < 1.0) {                       // SparseSequential skips BN when there is no active site
            mean_out[c] = 0.f; invstd_out[c] = 0.f; scale_out[c] = 0.f; shift_out[c] = 0.f;
            return;
        }
        double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        double invstd = 1.0 / sqrt(var + (double)eps);
        mean_out[c] = (float)mean;
        invstd_out[c] = (float)invstd;
        float sc = (float)((double)pg * invstd);
        scale_out[c] = sc;
        shift_out[c] = (float)((double)pb - mean * (double)pg * invstd);
        if (running_mean) {
            double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * prm + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * prv + momentum * unb);
        }
        if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
    }
}

extern "C" int tri_bn_finalize(const float* partial, int ntiles, int C, const int* count_dev, int count_host, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                               float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, void* stream) {
    bn_finalize_kernel<<<C, 256, 0, (hipStream_t)stream>>>(partial, ntiles, C, count_dev, count_host, gamma, beta,
                                                                      running_mean, running_var, num_batches_tracked, momentum,
                                                                      eps, mean, invstd, scale, shift);
    return tri_check_launch("tri_bn_finalize");
}

// Eval-mode scale/shift from running statistics.
__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* mean, float* invstd, float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float is = 1.0f / sqrtf(rv[c] + eps);
    mean[c] = rm[c]; invstd[c] = is; scale[c] = gamma[c] * is; shift[c] = beta[c] - rm[c] * gamma[c] * is;
}
extern "C" int tri_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                  float* mean, float* invstd, float* scale, float* shift, void* stream) {
    bn_eval_coeffs_kernel<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(C, gamma, beta, rm, rv, eps, mean, invstd, scale, shift);
    return tri_check_launch("tri_bn_eval_coeffs");
}

// ------------------------------------------------------------------------------------------ BN apply (+res, relu)
// out = act(y*scale + shift + residual),  residual = res (identity) or res*rscale + rshift (down-sample BN branch)
template <typename T>
__global__ void bn_act_kernel(const T* __restrict__ y, const float4* __restrict__ scale, const float4* __restrict__ shift,
                              const T* __restrict__ res, const float4* __restrict__ rscale, const float4* __restrict__ rshift,
                              T* __restrict__ out, long total4, int C4, int relu) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4);
        float4 v = Act<T>::ld4(y + i * 4), s = scale[c], b = shift[c];
        v.x = __fmaf_rn(v.x, s.x, b.x); v.y = __fmaf_rn(v.y, s.y, b.y); v.z = __fmaf_rn(v.z, s.z, b.z); v.w = __fmaf_rn(v.w, s.w, b.w);
        if (res) {
            float4 r = Act<T>::ld4(res + i * 4);
            if (rscale) {
                float4 rs = rscale[c], rb = rshift[c];
                r.x = r.x * rs.x + rb.x; r.y = r.y * rs.y + rb.y; r.z = r.z * rs.z + rb.z; r.w = r.w * rs.w + rb.w;
            }
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        Act<T>::st4(out + i * 4, v);
    }
}
static inline int ew_grid(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
extern "C" int tri_bn_act(const void* y, const float* scale, const float* shift, const void* res, const float* rscale,
                          const float* rshift, void* out, long M, int C, int relu, int act_fmt, void* stream) {
    if (C % 4) { tri_set_error("tri_bn_act: C must be a multiple of 4"); return TRI_ERR_ARG; }
    long total4 = M * (C / 4);
    TRI_ACT_DISPATCH(act_fmt, bn_act_kernel<T><<<ew_grid(total4), 256, 0, (hipStream_t)stream>>>(
        (const T*)y, (const float4*)scale, (const float4*)shift, (const T*)res, (const float4*)rscale, (const float4*)rshift, (T*)out,
        total4, C / 4, relu));
    return tri_check_launch("tri_bn_act");
}

// g = dout * (out > 0)   (ReLU backward from the saved output; may run in place on dout)
template <typename T>
__global__ void relu_bwd_kernel(const T* dout, const T* __restrict__ out, T* g, long total4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        float4 d = Act<T>::ld4(dout + i * 4), o = Act<T>::ld4(out + i * 4);
        d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
        Act<T>::st4(g + i * 4, d);
    }
}
extern "C" int tri_relu_bwd(const void* dout, const void* out, void* g, long n, int act_fmt, void* stream) {
    if (n % 4) { tri_set_error("tri_relu_bwd: n must be a multiple of 4"); return TRI_ERR_ARG; }
    TRI_ACT_DISPATCH(act_fmt, relu_bwd_kernel<T><<<ew_grid(n / 4), 256, 0, (hipStream_t)stream>>>((const T*)dout, (const T*)out, (T*)g, n / 4));
    return tri_check_launch("tri_relu_bwd");
}

// --------------------------------------------------------------------------------------------------- BN backward
// Pass 1: per-block partial sums of g and g*y per channel ([nblk][2][C]); pass 2 (tri_bn_bwd_finalize): dgamma, dbeta
// and the coefficients of dy = c1*g + c2 + c3*y; pass 3: apply (rows with row_mask == 0 stay zero).
// rows per block: 256 for large tensors, 64 for small ones (so that a 3,072-row layer still fills 48 CUs)
// ... and at most ~2,048 workgroups (= records for bn_bwd_finalize: a 2.4 M-row layer of the 224^2 configurations had 9,408)
static inline int bnb_rows(long M) {
    // small tensors are latency, not bandwidth: 16-row blocks put a 3,072-row layer on 192 CUs instead of 48 (7.3 -> ~3 us, round 4)
    if (M <= 16384) return 16;
    if (M <= 32768) return 32;
    if (M < 65536) return 64;
    long rows = (M + 2047) / 2048;
    rows = (rows + 63) / 64 * 64;
    return (int)(rows < 256 ? 256 : rows);
}
// MASK (compile time - a run-time test inside the row loop cost 1.6-2.5x on every launch): 0 g is final, 1 ReLU mask
// recomputed from y, 2 ReLU mask from the saved output `ro`
template <typename T, int MASK, bool ROWMASK>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ y, const T* __restrict__ g, long M, int C, float* __restrict__ partial,
                                     int BNB_ROWS, const float4* __restrict__ rs, const float4* __restrict__ rb,
                                     const T* __restrict__ ro, const uint8_t* __restrict__ row_mask) {
    extern __shared__ float sh[];                      // [rows_per_pass][C4*4][2]
    const int C4 = C >> 2;
    const int tpr = C4 < 256 ? C4 : 256;               // threads per row
    const int rpp = 256 / tpr;                         // rows per pass
    const int cpt = (C4 + tpr - 1) / tpr;              // float4 columns per thread
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const long r0 = (long)blockIdx.x * BNB_ROWS;
    const long r1 = r0 + BNB_ROWS < M ? r0 + BNB_ROWS : M;
    for (int cc = 0; cc < cpt; ++cc) {
        int c4 = tc + cc * tpr;
        float4 sg = make_float4(0, 0, 0, 0), sgy = make_float4(0, 0, 0, 0);
        float4 s4 = make_float4(0, 0, 0, 0), b4 = make_float4(1, 1, 1, 1);
        if (MASK == 1 && c4 < C4) { s4 = rs[c4]; b4 = rb[c4]; }
        if (c4 < C4 && tr < rpp)
#pragma unroll 4
            for (long r = r0 + tr; r < r1; r += rpp) {
                if (ROWMASK && !row_mask[r]) continue;           // inactive sites: y / g rows are not even written (compact conv rows)
                                                                 // (compile-time flag: a run-time test here halved the speed of every launch)
                float4 gv = Act<T>::ld4(g + r * C + c4 * 4), yv = Act<T>::ld4(y + r * C + c4 * 4);
                if (MASK == 1) {
                    gv.x = __fmaf_rn(yv.x, s4.x, b4.x) > 0.f ? gv.x : 0.f; gv.y = __fmaf_rn(yv.y, s4.y, b4.y) > 0.f ? gv.y : 0.f;
                    gv.z = __fmaf_rn(yv.z, s4.z, b4.z) > 0.f ? gv.z : 0.f; gv.w = __fmaf_rn(yv.w, s4.w, b4.w) > 0.f ? gv.w : 0.f;
                }
                if (MASK == 2) {                                    // ReLU taken after a residual add: mask from the saved output
                    const float4 ov = Act<T>::ld4(ro + r * C + c4 * 4);
                    gv.x = ov.x > 0.f ? gv.x : 0.f; gv.y = ov.y > 0.f ? gv.y : 0.f; gv.z = ov.z > 0.f ? gv.z : 0.f; gv.w = ov.w > 0.f ? gv.w : 0.f;
                }
                sg.x += gv.x; sg.y += gv.y; sg.z += gv.z; sg.w += gv.w;
                sgy.x += gv.x * yv.x; sgy.y += gv.y * yv.y; sgy.z += gv.z * yv.z; sgy.w += gv.w * yv.w;
            }
        __syncthreads();
        if (c4 < C4 && tr < rpp) {
            float* p = sh + ((size_t)tr * tpr + tc) * 8;
            p[0] = sg.x; p[1] = sg.y; p[2] = sg.z; p[3] = sg.w; p[4] = sgy.x; p[5] = sgy.y; p[6] = sgy.z; p[7] = sgy.w;
        }
        __syncthreads();
        if (tr == 0 && c4 < C4) {
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int rr = 0; rr < rpp; ++rr) {
                const float* p = sh + ((size_t)rr * tpr + tc) * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += p[k];
            }
            float* o = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[c4 * 4 + k] = a[k]; o[C + c4 * 4 + k] = a[4 + k]; }
        }
    }
}
extern "C" int tri_bn_bwd_num_blocks(long M) { return (int)((M + bnb_rows(M) - 1) / bnb_rows(M)); }
// relu_scale / relu_shift (optional, [C]): the BN output went through ReLU and `g` is the gradient w.r.t. the ReLU OUTPUT;
// the mask (y*scale + shift > 0, the forward's own expression) is recomputed instead of materialising relu_bwd's result.
// relu_out (optional, same shape as y): the ReLU came after a residual add (BasicBlock output); its mask is out > 0.
extern "C" int tri_bn_bwd_reduce(const void* y, const void* g, long M, int C, float* partial, const float* relu_scale,
                                 const float* relu_shift, const void* relu_out, const uint8_t* row_mask, int act_fmt, void* stream) {
    if (C % 4) { tri_set_error("tri_bn_bwd_reduce: C must be a multiple of 4"); return TRI_ERR_ARG; }
    int rows = bnb_rows(M);
    int nblk = (int)((M + rows - 1) / rows);
    int C4 = C / 4, tpr = C4 < 256 ? C4 : 256, rpp = 256 / tpr;
    size_t smem = (size_t)rpp * tpr * 8 * sizeof(float);
    if (relu_scale && relu_out) { tri_set_error("tri_bn_bwd_reduce: give either relu_scale/shift or relu_out"); return TRI_ERR_ARG; }
#define TRI_BNR(MASK_)                                                                                                              \
    do {                                                                                                                            \
        if (row_mask) TRI_ACT_DISPATCH(act_fmt, bn_bwd_reduce_kernel<T, MASK_, true><<<nblk, 256, smem, (hipStream_t)stream>>>(       \
            (const T*)y, (const T*)g, M, C, partial, rows, (const float4*)relu_scale, (const float4*)relu_shift, (const T*)relu_out, row_mask)); \
        else TRI_ACT_DISPATCH(act_fmt, bn_bwd_reduce_kernel<T, MASK_, false><<<nblk, 256, smem, (hipStream_t)stream>>>(               \
            (const T*)y, (const T*)g, M, C, partial, rows, (const float4*)relu_scale, (const float4*)relu_shift, (const T*)relu_out, row_mask)); \
    } while (0)
    if (relu_out) TRI_BNR(2);
    else if (relu_scale) TRI_BNR(1);
    else TRI_BNR(0);
#undef TRI_BNR
    return tri_check_launch("tri_bn_bwd_reduce");
}

__device__ __forceinline__ void bn_bwd_finalize_body(const float* __restrict__ partial, int nblk, int C, const int* __restrict__ count_dev,
                                                     int count_host, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ c1, float* __restrict__ c2, float* __restrict__ c3, float out_scale,
                                                     int c) {
    __shared__ double ssum[4], ssq[4];                              // one channel per block, see bn_finalize_kernel
    float pmu = 0.f, pis = 0.f, pga = 0.f;
    int pcount = count_host;
    if (threadIdx.x == 0) {                                         // requested up front, see bn_finalize_kernel
        pmu = mean[c]; pis = invstd[c]; pga = gamma[c];
        if (count_dev) pcount = *count_dev;
    }
    double s = 0.0, q = 0.0;
#pragma unroll 8
    for (int b = threadIdx.x; b < nblk; b += 256) {
        s += (double)partial[((size_t)b * 2 + 0) * C + c];
        q += (double)partial[((size_t)b * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = s; ssq[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        q = (ssq[0] + ssq[1]) + (ssq[2] + ssq[3]);
        double n = (double)pcount;
        if (n < 1.0) { dgamma[c] = 0.f; dbeta[c] = 0.f; c1[c] = 0.f; c2[c] = 0.f; c3[c] = 0.f; return; }
        double mu = pmu, is = pis, ga = pga;
        double dbe = s;                                   // sum g
        double dga = is * (q - mu * s);                   // sum g * xhat
        dgamma[c] = (float)(dga * (double)out_scale);     // parameter gradients leave the tower unscaled (f16 mode: g is scaled)
        dbeta[c] = (float)(dbe * (double)out_scale);
        double k1 = ga * is;
        double k3 = -ga * is * is * dga / n;
        double k2 = -k1 * dbe / n - k3 * mu;
        c1[c] = (float)k1; c2[c] = (float)k2; c3[c] = (float)k3;
    }
}
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, const int* __restrict__ count_dev,
                                       int count_host, const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ c1, float* __restrict__ c2, float* __restrict__ c3, float out_scale) {
    bn_bwd_finalize_body(partial, nblk, C, count_dev, count_host, gamma, mean, invstd, dgamma, dbeta, c1, c2, c3, out_scale, blockIdx.x);
}
// two BatchNorms of one shape in one launch (blocks [0, C): set a, [C, 2 C): set b); buf_* = [5][C] (dgamma, dbeta, c1, c2, c3)
__global__ void bn_bwd_finalize_pair_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int nblk, int C, int count_host,
                                            const float* __restrict__ ga, const float* __restrict__ ma, const float* __restrict__ ia,
                                            float* __restrict__ buf_a, const float* __restrict__ gb, const float* __restrict__ mb,
                                            const float* __restrict__ ib, float* __restrict__ buf_b, float out_scale) {
    const bool second = (int)blockIdx.x >= C;
    const int c = second ? blockIdx.x - C : blockIdx.x;
    float* buf = second ? buf_b : buf_a;
    bn_bwd_finalize_body(second ? pb : pa, nblk, C, nullptr, count_host, second ? gb : ga, second ? mb : ma, second ? ib : ia, buf, buf + C,
                         buf + 2 * C, buf + 3 * C, buf + 4 * C, out_scale, c);
}
extern "C" int tri_bn_bwd_finalize(const float* partial, int nblk, int C, const int* count_dev, int count_host, const float* gamma,
                                   const float* mean, const float* invstd, float* dgamma, float* dbeta, float* c1, float* c2,
                                   float* c3, float out_scale, void* stream) {
    bn_bwd_finalize_kernel<<<C, 256, 0, (hipStream_t)stream>>>(partial, nblk, C, count_dev, count_host, gamma, mean,
                                                                          invstd, dgamma, dbeta, c1, c2, c3, out_scale);
    return tri_check_launch("tri_bn_bwd_finalize");
}

template <typename T, int MASK>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ y, const T* g, const float4* __restrict__ c1,
                                    const float4* __restrict__ c2, const float4* __restrict__ c3,
                                    const uint8_t* __restrict__ row_mask, T* dy, long total4, int C4,
                                    const float4* __restrict__ rs, const float4* __restrict__ rb, const T* __restrict__ ro, T* gm,
                                    int keep_inactive) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        long row = i / C4;
        int c = (int)(i - row * C4);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (keep_inactive && row_mask && !row_mask[row]) continue;    // nobody reads dy there (a first layer: no data gradient): no zeros written
        if (!row_mask || row_mask[row]) {
            float4 yv = Act<T>::ld4(y + i * 4), gv = Act<T>::ld4(g + i * 4), a = c1[c], b = c2[c], d = c3[c];
            if (MASK == 1) {                                        // ReLU mask recomputed from y (see tri_bn_bwd_reduce)
                const float4 s4 = rs[c], b4 = rb[c];
                gv.x = __fmaf_rn(yv.x, s4.x, b4.x) > 0.f ? gv.x : 0.f; gv.y = __fmaf_rn(yv.y, s4.y, b4.y) > 0.f ? gv.y : 0.f;
                gv.z = __fmaf_rn(yv.z, s4.z, b4.z) > 0.f ? gv.z : 0.f; gv.w = __fmaf_rn(yv.w, s4.w, b4.w) > 0.f ? gv.w : 0.f;
            }
            if (MASK == 2) {
                const float4 ov = Act<T>::ld4(ro + i * 4);
                gv.x = ov.x > 0.f ? gv.x : 0.f; gv.y = ov.y > 0.f ? gv.y : 0.f; gv.z = ov.z > 0.f ? gv.z : 0.f; gv.w = ov.w > 0.f ? gv.w : 0.f;
                if (gm) Act<T>::st4(gm + i * 4, gv);                // the masked gradient also feeds the identity / down-sample branch
            }
            o.x = a.x * gv.x + b.x + d.x * yv.x; o.y = a.y * gv.y + b.y + d.y * yv.y;
            o.z = a.z * gv.z + b.z + d.z * yv.z; o.w = a.w * gv.w + b.w + d.w * yv.w;
        }
        Act<T>::st4(dy + i * 4, o);
    }
}
extern "C" int tri_bn_bwd_apply(const void* y, const void* g, const float* c1, const float* c2, const float* c3,
                                const uint8_t* row_mask, void* dy, long M, int C, const float* relu_scale, const float* relu_shift,
                                const void* relu_out, void* g_masked, int keep_inactive, int act_fmt, void* stream) {
    long total4 = M * (C / 4);
#define TRI_BNA(MASK_)                                                                                                              \
    TRI_ACT_DISPATCH(act_fmt, bn_bwd_apply_kernel<T, MASK_><<<ew_grid(total4), 256, 0, (hipStream_t)stream>>>(                      \
        (const T*)y, (const T*)g, (const float4*)c1, (const float4*)c2, (const float4*)c3, row_mask, (T*)dy, total4, C / 4,          \
        (const float4*)relu_scale, (const float4*)relu_shift, (const T*)relu_out, (T*)g_masked, keep_inactive))
    if (relu_out) TRI_BNA(2);
    else if (relu_scale) TRI_BNA(1);
    else TRI_BNA(0);
#undef TRI_BNA
    return tri_check_launch("tri_bn_bwd_apply");
}

// ---- BatchNorm backward of TWO tensors that share their upstream gradient (round 6): the last BatchNorm of a down-sampling
// BasicBlock (bn2 of y2) and the shortcut's BatchNorm (of yd) both receive g = dout * (out > 0) (mv_cnn.py:20: torchvision BasicBlock,
// out = relu(bn2(y2) + bn_d(yd))).  The three passes serve both tensors: g and the saved output are read once, the sums of g are shared,
// and the shortcut branch is left with its 1x1 / 2 data gradient alone.  Same per-thread summation order and expressions as the single forms.
template <typename T>
__global__ void bn_bwd_reduce_pair_kernel(const T* __restrict__ ya, const T* __restrict__ yb, const T* __restrict__ g, long M, int C,
                                          float* __restrict__ partial_a, float* __restrict__ partial_b, int BNB_ROWS,
                                          const T* __restrict__ ro) {
    extern __shared__ float sh[];                      // [rows_per_pass][C4][12]
    const int C4 = C >> 2;
    const int tpr = C4 < 256 ? C4 : 256;
    const int rpp = 256 / tpr;
    const int cpt = (C4 + tpr - 1) / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const long r0 = (long)blockIdx.x * BNB_ROWS;
    const long r1 = r0 + BNB_ROWS < M ? r0 + BNB_ROWS : M;
    for (int cc = 0; cc < cpt; ++cc) {
        int c4 = tc + cc * tpr;
        float4 sg = make_float4(0, 0, 0, 0), sga = make_float4(0, 0, 0, 0), sgb = make_float4(0, 0, 0, 0);
        if (c4 < C4 && tr < rpp)
#pragma unroll 4
            for (long r = r0 + tr; r < r1; r += rpp) {
                float4 gv = Act<T>::ld4(g + r * C + c4 * 4);
                const float4 av = Act<T>::ld4(ya + r * C + c4 * 4), bv = Act<T>::ld4(yb + r * C + c4 * 4), ov = Act<T>::ld4(ro + r * C + c4 * 4);
                gv.x = ov.x > 0.f ? gv.x : 0.f; gv.y = ov.y > 0.f ? gv.y : 0.f; gv.z = ov.z > 0.f ? gv.z : 0.f; gv.w = ov.w > 0.f ? gv.w : 0.f;
                sg.x += gv.x; sg.y += gv.y; sg.z += gv.z; sg.w += gv.w;
                sga.x += gv.x * av.x; sga.y += gv.y * av.y; sga.z += gv.z * av.z; sga.w += gv.w * av.w;
                sgb.x += gv.x * bv.x; sgb.y += gv.y * bv.y; sgb.z += gv.z * bv.z; sgb.w += gv.w * bv.w;
            }
        __syncthreads();
        if (c4 < C4 && tr < rpp) {
            float* p = sh + ((size_t)tr * tpr + tc) * 12;
            p[0] = sg.x; p[1] = sg.y; p[2] = sg.z; p[3] = sg.w; p[4] = sga.x; p[5] = sga.y; p[6] = sga.z; p[7] = sga.w;
            p[8] = sgb.x; p[9] = sgb.y; p[10] = sgb.z; p[11] = sgb.w;
        }
        __syncthreads();
        if (tr == 0 && c4 < C4) {
            float a[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int rr = 0; rr < rpp; ++rr) {
                const float* p = sh + ((size_t)rr * tpr + tc) * 12;
#pragma unroll
                for (int k = 0; k < 12; ++k) a[k] += p[k];
            }
            float* oa = partial_a + (size_t)blockIdx.x * 2 * C;
            float* ob = partial_b + (size_t)blockIdx.x * 2 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) { oa[c4 * 4 + k] = a[k]; oa[C + c4 * 4 + k] = a[4 + k]; ob[c4 * 4 + k] = a[k]; ob[C + c4 * 4 + k] = a[8 + k]; }
        }
    }
}
template <typename T>
__global__ void bn_bwd_apply_pair_kernel(const T* __restrict__ ya, const T* __restrict__ yb, const T* g, const float4* __restrict__ ca,
                                         const float4* __restrict__ cb, T* dya, T* dyb, long total4, int C4, const T* __restrict__ ro, T* gm) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        float4 gv = Act<T>::ld4(g + i * 4);
        const float4 av = Act<T>::ld4(ya + i * 4), bv = Act<T>::ld4(yb + i * 4), ov = Act<T>::ld4(ro + i * 4);
        gv.x = ov.x > 0.f ? gv.x : 0.f; gv.y = ov.y > 0.f ? gv.y : 0.f; gv.z = ov.z > 0.f ? gv.z : 0.f; gv.w = ov.w > 0.f ? gv.w : 0.f;
        if (gm) Act<T>::st4(gm + i * 4, gv);
        {
            const float4 a = ca[c], b = ca[C4 + c], d = ca[2 * C4 + c];
            float4 o;
            o.x = a.x * gv.x + b.x + d.x * av.x; o.y = a.y * gv.y + b.y + d.y * av.y;
            o.z = a.z * gv.z + b.z + d.z * av.z; o.w = a.w * gv.w + b.w + d.w * av.w;
            asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(o.w));      // (keeps the storage conversion a rounding of its own, as in bn_bwd_apply_kernel)
            Act<T>::st4(dya + i * 4, o);
        }
        {
            const float4 a = cb[c], b = cb[C4 + c], d = cb[2 * C4 + c];
            float4 o;
            o.x = a.x * gv.x + b.x + d.x * bv.x; o.y = a.y * gv.y + b.y + d.y * bv.y;
            o.z = a.z * gv.z + b.z + d.z * bv.z; o.w = a.w * gv.w + b.w + d.w * bv.w;
            // (no barrier here: the shortcut's single pass - bn_bwd_apply_kernel<T, 0> - lets the compiler fold its last FMA and the f16 conversion
            //  into v_fma_mixlo_f16, the <T, 2> form above does not; the bit-equality test of the two paths watches over both choices)
            Act<T>::st4(dyb + i * 4, o);
        }
    }
}
extern "C" int tri_bn_bwd_pair_reduce(const void* ya, const void* yb, const void* g, const void* relu_out, long M, int C, float* partial_a,
                                      float* partial_b, int act_fmt, void* stream) {
    if (C % 4 || !relu_out) { tri_set_error("tri_bn_bwd_pair_reduce: C % 4 == 0 and the saved output are required"); return TRI_ERR_ARG; }
    const int rows = bnb_rows(M), nblk = (int)((M + rows - 1) / rows);
    const int C4 = C / 4, tpr = C4 < 256 ? C4 : 256, rpp = 256 / tpr;
    const size_t smem = (size_t)rpp * tpr * 12 * sizeof(float);
    TRI_ACT_DISPATCH(act_fmt, bn_bwd_reduce_pair_kernel<T><<<nblk, 256, smem, (hipStream_t)stream>>>(
        (const T*)ya, (const T*)yb, (const T*)g, M, C, partial_a, partial_b, rows, (const T*)relu_out));
    return tri_check_launch("tri_bn_bwd_pair_reduce");
}
extern "C" int tri_bn_bwd_pair_finalize(const float* partial_a, const float* partial_b, int nblk, int C, int count_host, const float* gamma_a,
                                        const float* mean_a, const float* invstd_a, float* buf_a, const float* gamma_b, const float* mean_b,
                                        const float* invstd_b, float* buf_b, float out_scale, void* stream) {
    bn_bwd_finalize_pair_kernel<<<2 * C, 256, 0, (hipStream_t)stream>>>(partial_a, partial_b, nblk, C, count_host, gamma_a, mean_a, invstd_a, buf_a,
                                                                       gamma_b, mean_b, invstd_b, buf_b, out_scale);
    return tri_check_launch("tri_bn_bwd_pair_finalize");
}
extern "C" int tri_bn_bwd_pair_apply(const void* ya, const void* yb, const void* g, const void* relu_out, const float* buf_a, const float* buf_b,
                                     void* dya, void* dyb, void* g_masked, long M, int C, int act_fmt, void* stream) {
    if (C % 4) { tri_set_error("tri_bn_bwd_pair_apply: C must be a multiple of 4"); return TRI_ERR_ARG; }
    const long total4 = M * (C / 4);
    TRI_ACT_DISPATCH(act_fmt, bn_bwd_apply_pair_kernel<T><<<ew_grid(total4), 256, 0, (hipStream_t)stream>>>(
        (const T*)ya, (const T*)yb, (const T*)g, (const float4*)(buf_a + 2 * C), (const float4*)(buf_b + 2 * C), (T*)dya, (T*)dyb, total4, C / 4,
        (const T*)relu_out, (T*)g_masked));
    return tri_check_launch("tri_bn_bwd_pair_apply");
}

// ---- BatchNorm backward of a TINY tensor in one launch (tri_bn_bwd_small)
// The deepest voxel level holds a few hundred rows: its three passes (reduce on 4 workgroups: 22 us, finalize, apply, two launch
// boundaries) are pure latency.  Here one workgroup owns CH channels for ALL M positions: pass 1 sums g and g * y over the positions (every load of the pass in flight at once), the coefficients are formed in
// double by the workgroup itself, pass 2 re-reads y and g - L2 / L1 hits, the tensor was just read - and writes dy.  No records, no
// finalize launch, no second kernel.  A position row is C * 2 bytes, of which a workgroup reads CH * 2: the 128-byte lines are shared
// with the workgroups of the neighbouring channel groups, so groups are dealt to the XCDs in contiguous runs (block b: XCD b % 8 takes
// groups [b % 8 * nblk / 8 ...)) - each XCD's L2 then fetches only its own columns of the tensor.
// MEASURED (round 4, HIP events, eager): 256 x 512: 3.4 us against 22.4 + 1.4 + 1.7; but every wave load touches 64 cache lines for 16
// bytes each, so it loses from ~1 k rows on (2,048 x 256: 13.1 against 9.8 + 1.4 + 1.6; 3,072 x 512: 32 against 11; 12,288 x 256: 89
// against 11): ops.bn_bwd takes it for M <= 512 only.
template <typename T, int MASK, int CH>
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(const T* __restrict__ y, const T* g, long M, int C, const int* __restrict__ count_dev,
                                                           int count_host, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ rs,
                                                           const float* __restrict__ rb, const T* __restrict__ ro, T* gm,
                                                           const uint8_t* __restrict__ row_mask, int keep_inactive, T* dy,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float out_scale) {
    constexpr int Q = CH / 4;                                        // float4 quads per position
    __shared__ float red[4][2 * CH];
    __shared__ float coef[3 * CH];
    const int nblk = gridDim.x;
    const int grp = (nblk % 8 == 0) ? (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int c0 = grp * CH;
    const int t = threadIdx.x;
    float4 s4[Q], b4[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        s4[q] = make_float4(0.f, 0.f, 0.f, 0.f); b4[q] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (MASK == 1) { s4[q] = *(const float4*)(rs + c0 + 4 * q); b4[q] = *(const float4*)(rb + c0 + 4 * q); }
    }
    auto masked = [&](float4 gv, const float4& yv, const float4& ov, int q) -> float4 {
        if (MASK == 1) {
            gv.x = __fmaf_rn(yv.x, s4[q].x, b4[q].x) > 0.f ? gv.x : 0.f; gv.y = __fmaf_rn(yv.y, s4[q].y, b4[q].y) > 0.f ? gv.y : 0.f;
            gv.z = __fmaf_rn(yv.z, s4[q].z, b4[q].z) > 0.f ? gv.z : 0.f; gv.w = __fmaf_rn(yv.w, s4[q].w, b4[q].w) > 0.f ? gv.w : 0.f;
        }
        if (MASK == 2) { gv.x = ov.x > 0.f ? gv.x : 0.f; gv.y = ov.y > 0.f ? gv.y : 0.f; gv.z = ov.z > 0.f ? gv.z : 0.f; gv.w = ov.w > 0.f ? gv.w : 0.f; }
        return gv;
    };
    float4 sg[Q], sgy[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) sg[q] = sgy[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (long r = t; r < M; r += 256) {
        if (row_mask && !row_mask[r]) continue;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const long o = r * C + c0 + 4 * q;
            const float4 yv = Act<T>::ld4(y + o);
            float4 ov = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MASK == 2) ov = Act<T>::ld4(ro + o);
            const float4 gv = masked(Act<T>::ld4(g + o), yv, ov, q);
            sg[q].x += gv.x; sg[q].y += gv.y; sg[q].z += gv.z; sg[q].w += gv.w;
            sgy[q].x += gv.x * yv.x; sgy[q].y += gv.y * yv.y; sgy[q].z += gv.z * yv.z; sgy[q].w += gv.w * yv.w;
        }
    }
    {   // wave sums, then the four waves through LDS
        float v[2 * CH];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            v[4 * q] = sg[q].x; v[4 * q + 1] = sg[q].y; v[4 * q + 2] = sg[q].z; v[4 * q + 3] = sg[q].w;
            v[CH + 4 * q] = sgy[q].x; v[CH + 4 * q + 1] = sgy[q].y; v[CH + 4 * q + 2] = sgy[q].z; v[CH + 4 * q + 3] = sgy[q].w;
        }
#pragma unroll
        for (int k = 0; k < 2 * CH; ++k) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
        }
        if ((t & 63) == 0)
#pragma unroll
            for (int k = 0; k < 2 * CH; ++k) red[t >> 6][k] = v[k];
    }
    __syncthreads();
    if (t < CH) {
        const double s = ((double)red[0][t] + (double)red[1][t]) + ((double)red[2][t] + (double)red[3][t]);
        const double q = ((double)red[0][CH + t] + (double)red[1][CH + t]) + ((double)red[2][CH + t] + (double)red[3][CH + t]);
        const double n = (double)(count_dev ? *count_dev : count_host);
        const int c = c0 + t;
        if (n < 1.0) { dgamma[c] = 0.f; dbeta[c] = 0.f; coef[t] = 0.f; coef[CH + t] = 0.f; coef[2 * CH + t] = 0.f; }
        else {                                                        // as bn_bwd_finalize_kernel
            const double mu = mean[c], is = invstd[c], ga = gamma[c];
            const double dbe = s, dga = is * (q - mu * s);
            dgamma[c] = (float)(dga * (double)out_scale);
            dbeta[c] = (float)(dbe * (double)out_scale);
            const double k1 = ga * is, k3 = -ga * is * is * dga / n, k2 = -k1 * dbe / n - k3 * mu;
            coef[t] = (float)k1; coef[CH + t] = (float)k2; coef[2 * CH + t] = (float)k3;
        }
    }
    __syncthreads();
    float4 k1[Q], k2[Q], k3[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        k1[q] = *(const float4*)(coef + 4 * q); k2[q] = *(const float4*)(coef + CH + 4 * q); k3[q] = *(const float4*)(coef + 2 * CH + 4 * q);
    }
#pragma unroll 4
    for (long r = t; r < M; r += 256) {
        const bool live = !row_mask || row_mask[r];
        if (!live && keep_inactive) continue;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const long o = r * C + c0 + 4 * q;
            float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                const float4 yv = Act<T>::ld4(y + o);
                float4 ov = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MASK == 2) ov = Act<T>::ld4(ro + o);
                const float4 gv = masked(Act<T>::ld4(g + o), yv, ov, q);
                if (MASK == 2 && gm) Act<T>::st4(gm + o, gv);
                out.x = k1[q].x * gv.x + k2[q].x + k3[q].x * yv.x; out.y = k1[q].y * gv.y + k2[q].y + k3[q].y * yv.y;
                out.z = k1[q].z * gv.z + k2[q].z + k3[q].z * yv.z; out.w = k1[q].w * gv.w + k2[q].w + k3[q].w * yv.w;
            }
            Act<T>::st4(dy + o, out);
        }
    }
}
// The whole BatchNorm backward of a small layer: (dy, dgamma, dbeta) from y, g and the forward's mean / invstd; relu_scale / relu_shift,
// relu_out, g_masked, row_mask, keep_inactive as in tri_bn_bwd_reduce / tri_bn_bwd_apply (dy may alias g; g_masked may alias g).
// TRI_ERR_UNSUPPORTED for tensors it is not meant for (M > 16,384 rows or C % 8 != 0): the caller then runs the three passes.
extern "C" int tri_bn_bwd_small(const void* y, const void* g, long M, int C, const int* count_dev, int count_host, const float* gamma,
                                const float* mean, const float* invstd, const float* relu_scale, const float* relu_shift,
                                const void* relu_out, void* g_masked, const uint8_t* row_mask, int keep_inactive, void* dy, float* dgamma,
                                float* dbeta, float out_scale, int act_fmt, void* stream) {
    if (M > 16384 || M < 1 || C % 8 || C < 64) { tri_set_error("tri_bn_bwd_small: M <= 16384 rows, C % 8 == 0, C >= 64"); return TRI_ERR_UNSUPPORTED; }
    if (relu_scale && relu_out) { tri_set_error("tri_bn_bwd_small: give either relu_scale/shift or relu_out"); return TRI_ERR_ARG; }
    const int ch = C / 8 >= 64 ? 8 : 4;                             // at least 64 workgroups where the channel count allows
    hipStream_t st = (hipStream_t)stream;
#define TRI_BNS1(MASK_, CH_)                                                                                                             \
    TRI_ACT_DISPATCH(act_fmt, bn_bwd_small_kernel<T, MASK_, CH_><<<C / CH_, 256, 0, st>>>(                                               \
        (const T*)y, (const T*)g, M, C, count_dev, count_host, gamma, mean, invstd, relu_scale, relu_shift, (const T*)relu_out,         \
        (T*)g_masked, row_mask, keep_inactive, (T*)dy, dgamma, dbeta, out_scale))
#define TRI_BNS(MASK_) do { if (ch == 8) TRI_BNS1(MASK_, 8); else TRI_BNS1(MASK_, 4); } while (0)
    if (relu_out) TRI_BNS(2);
    else if (relu_scale) TRI_BNS(1);
    else TRI_BNS(0);
#undef TRI_BNS
#undef TRI_BNS1
    return tri_check_launch("tri_bn_bwd_small");
}

// --------------------------------------------------------------------- voxel: BN + ReLU + mask + 2^3 max-pool
// y [B,D,D,D,C] raw conv output, mask [B,D,D,D]; pooled [B,D/2,..,C], mask_out = OR of children
// The 2x2x2 window of pooled site `pos`: offset of its first child in the level's site order, and one bit per child (d,h,w scan order,
// bit k = child k active) from FOUR 2-byte mask loads.  All window kernels below read the mask this way and then issue the loads of the
// active children TOGETHER (inactive children read row 0 of the tensor - one cached line - and are discarded by a select): the first
// version tested mask[ip] child by child, which the compiler turned into sixteen dependent memory round trips per pooled site
// (byte, branch, row, branch, ...; ~11 us per grid-stride iteration; profiles/r5/NOTES_voxel.md).
struct PoolWin { unsigned base, bits; };
static __device__ __forceinline__ PoolWin pool_window(const uint8_t* __restrict__ mask, unsigned pos, unsigned D) {
    const unsigned Do = D >> 1;
    const unsigned ox = pos % Do; unsigned r = pos / Do;
    const unsigned oy = r % Do; r /= Do;
    const unsigned oz = r % Do, b = r / Do;
    PoolWin w;
    w.base = ((b * D + oz * 2) * D + oy * 2) * D + ox * 2;
    const unsigned m0 = *(const unsigned short*)(mask + w.base), m1 = *(const unsigned short*)(mask + w.base + D);
    const unsigned m2 = *(const unsigned short*)(mask + w.base + D * D), m3 = *(const unsigned short*)(mask + w.base + D * D + D);
    w.bits = ((m0 & 0xffu) ? 1u : 0u) | ((m0 >> 8) ? 2u : 0u) | ((m1 & 0xffu) ? 4u : 0u) | ((m1 >> 8) ? 8u : 0u) |
             ((m2 & 0xffu) ? 16u : 0u) | ((m2 >> 8) ? 32u : 0u) | ((m3 & 0xffu) ? 64u : 0u) | ((m3 >> 8) ? 128u : 0u);
    return w;
}
static __device__ __forceinline__ unsigned pool_child(unsigned base, unsigned D, int k) {
    return base + (unsigned)(k >> 2) * D * D + (unsigned)((k >> 1) & 1) * D + (unsigned)(k & 1);
}
template <typename T>
__global__ void bn_relu_pool3d_fwd_kernel(const T* __restrict__ y, const float4* __restrict__ scale, const float4* __restrict__ shift,
                                          const uint8_t* __restrict__ mask, int B, int D, int C4, T* __restrict__ pooled,
                                          uint8_t* __restrict__ mask_out) {
    const unsigned Do = D >> 1, npos = (unsigned)B * Do * Do * Do, total = npos * (unsigned)C4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned c = i % (unsigned)C4, pos = i / (unsigned)C4;
        const PoolWin w = pool_window(mask, pos, (unsigned)D);
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w.bits) {                                  // (84 % of the finest level's windows are empty: their lanes issue no row loads)
            const float4 s = scale[c], t = shift[c];
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const size_t row = (w.bits >> k) & 1u ? pool_child(w.base, (unsigned)D, k) : 0;
                v[k] = Act<T>::ld4(y + (row * C4 + c) * 4);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool on = (w.bits >> k) & 1u;
                const float zx = fmaxf(best.x, __fmaf_rn(v[k].x, s.x, t.x)), zy = fmaxf(best.y, __fmaf_rn(v[k].y, s.y, t.y));
                const float zz = fmaxf(best.z, __fmaf_rn(v[k].z, s.z, t.z)), zw = fmaxf(best.w, __fmaf_rn(v[k].w, s.w, t.w));
                best.x = on ? zx : best.x; best.y = on ? zy : best.y; best.z = on ? zz : best.z; best.w = on ? zw : best.w;
            }
        }
        Act<T>::st4(pooled + (size_t)i * 4, best);     // best >= 0: ReLU folded into the max with the zero init
        if (c == 0 && mask_out) mask_out[pos] = (uint8_t)(w.bits != 0);
    }
    // every site's byte is written above; the padding of mask_out (to 32 bytes) is zeroed here, so callers pre-fill nothing
    const unsigned pad = (npos + 31) / 32 * 32;
    if (mask_out && blockIdx.x == 0 && npos + threadIdx.x < pad) mask_out[npos + threadIdx.x] = 0;
}
// the window kernels index sites and (site, channel quad) pairs with 32-bit integers and read the mask two bytes at a time
static bool pool_args_ok(const char* who, const uint8_t* mask, int B, int D, int C) {
    const long sites = (long)B * D * D * D;
    if (C % 4 || D % 2 || sites >= (1L << 31) || sites / 8 * (C / 4) >= (1L << 31) || ((uintptr_t)mask & 1)) {
        char msg[256];
        snprintf(msg, sizeof msg, "%s: needs C %% 4 == 0, D %% 2 == 0, fewer than 2^31 sites and (pooled site, channel quad) pairs, and a 2-byte aligned mask", who);
        tri_set_error(msg);
        return false;
    }
    return true;
}
extern "C" int tri_bn_relu_pool3d_fwd(const void* y, const float* scale, const float* shift, const uint8_t* mask, int B, int D, int C,
                                      void* pooled, uint8_t* mask_out, int act_fmt, void* stream) {
    if (!pool_args_ok("tri_bn_relu_pool3d_fwd", mask, B, D, C)) return TRI_ERR_ARG;
    long total = (long)B * (D / 2) * (D / 2) * (D / 2) * (C / 4);
    TRI_ACT_DISPATCH(act_fmt, bn_relu_pool3d_fwd_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(
        (const T*)y, (const float4*)scale, (const float4*)shift, mask, B, D, C / 4, (T*)pooled, mask_out));
    return tri_check_launch("tri_bn_relu_pool3d_fwd");
}

// g[B,D,D,D,C] = gradient w.r.t. the BN output: dpooled routed to the FIRST child (d,h,w scan order, as
// torch.max_pool3d) whose post-ReLU value equals the pooled maximum and is > 0; zero elsewhere; rows of INACTIVE sites are
// left unwritten (tri_bn_bwd_reduce / tri_bn_bwd_apply skip them by the same mask, the latter writes their zeros).
// (with 16-bit storage the recomputed value is rounded like the stored maximum before the comparison)
// Routing of one pooled (site, channel quad) to its window: the rows of the active children are loaded together (see pool_window), the
// first-maximum rule is then evaluated in registers in scan order, and only active children are stored.  Returns through sg / sgy the
// sums of the routed gradient and of gradient * y over the window (the BatchNorm-backward sums; ignored by callers that do not need them).
template <typename T>
static __device__ __forceinline__ void route_window(const T* __restrict__ y, const float4& s, const float4& t, const PoolWin& w, unsigned D,
                                                    unsigned C4, unsigned c, const float4& pm, const float4& dp, T* __restrict__ g,
                                                    float4& sg, float4& sgy) {
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t row = (w.bits >> k) & 1u ? pool_child(w.base, D, k) : 0;
        v[k] = Act<T>::ld4(y + (row * C4 + c) * 4);
    }
    bool dx = false, dy = false, dz = false, dw = false;          // already routed
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool on = (w.bits >> k) & 1u;
        float zx = fmaxf(__fmaf_rn(v[k].x, s.x, t.x), 0.f), zy = fmaxf(__fmaf_rn(v[k].y, s.y, t.y), 0.f);
        float zz = fmaxf(__fmaf_rn(v[k].z, s.z, t.z), 0.f), zw = fmaxf(__fmaf_rn(v[k].w, s.w, t.w), 0.f);
        zx = Act<T>::rnd(zx); zy = Act<T>::rnd(zy); zz = Act<T>::rnd(zz); zw = Act<T>::rnd(zw);
        const bool hx = on && !dx && zx == pm.x && zx > 0.f, hy = on && !dy && zy == pm.y && zy > 0.f;
        const bool hz = on && !dz && zz == pm.z && zz > 0.f, hw = on && !dw && zw == pm.w && zw > 0.f;
        dx |= hx; dy |= hy; dz |= hz; dw |= hw;
        const float4 o = make_float4(hx ? dp.x : 0.f, hy ? dp.y : 0.f, hz ? dp.z : 0.f, hw ? dp.w : 0.f);
        if (on) Act<T>::st4(g + ((size_t)pool_child(w.base, D, k) * C4 + c) * 4, o);
        sg.x += o.x; sg.y += o.y; sg.z += o.z; sg.w += o.w;
        sgy.x += hx ? o.x * v[k].x : 0.f; sgy.y += hy ? o.y * v[k].y : 0.f; sgy.z += hz ? o.z * v[k].z : 0.f; sgy.w += hw ? o.w * v[k].w : 0.f;
    }
}
template <typename T>
__global__ void pool3d_bwd_route_kernel(const T* __restrict__ y, const float4* __restrict__ scale, const float4* __restrict__ shift,
                                        const uint8_t* __restrict__ mask, const T* __restrict__ pooled,
                                        const T* __restrict__ dpooled, int B, int D, int C4, T* __restrict__ g) {
    const unsigned Do = D >> 1, total = (unsigned)B * Do * Do * Do * (unsigned)C4;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned c = i % (unsigned)C4, pos = i / (unsigned)C4;
        const PoolWin w = pool_window(mask, pos, (unsigned)D);
        if (!w.bits) continue;                         // rows of inactive sites are never read (bn_bwd reduce / apply skip them)
        const float4 pm = Act<T>::ld4(pooled + (size_t)i * 4), dp = Act<T>::ld4(dpooled + (size_t)i * 4);
        float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgy = sg;
        route_window<T>(y, scale[c], shift[c], w, (unsigned)D, (unsigned)C4, c, pm, dp, g, sg, sgy);
    }
}
extern "C" int tri_pool3d_bwd_route(const void* y, const float* scale, const float* shift, const uint8_t* mask, const void* pooled,
                                    const void* dpooled, int B, int D, int C, void* g, int act_fmt, void* stream) {
    if (!pool_args_ok("tri_pool3d_bwd_route", mask, B, D, C)) return TRI_ERR_ARG;
    long total = (long)B * (D / 2) * (D / 2) * (D / 2) * (C / 4);
    TRI_ACT_DISPATCH(act_fmt, pool3d_bwd_route_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(
        (const T*)y, (const float4*)scale, (const float4*)shift, mask, (const T*)pooled, (const T*)dpooled, B, D, C / 4, (T*)g));
    return tri_check_launch("tri_pool3d_bwd_route");
}

// Row-list forms (round 4).  At 13-16 % occupancy the dense passes of the two finest voxel levels spend most of their threads on
// sites whose mask byte says "skip": these walk the compact lists tri_mask_compact already produced for the conv kernels instead.
// tri_pool3d_bwd_route_rows: the routing pass over the ACTIVE pooled sites (the next level's row list) - same values, same rows written.
// RED: the BatchNorm-backward sums of the level (sum g, sum g * y over the active sites = over the windows of the active pooled sites) leave
// as one [2][C] record per workgroup, as tri_pool3d_bwd_route_reduce does for the dense walk: no separate reduce pass over y and g.
// Needs 256 % C4 == 0 (a thread keeps its channel quad across the grid-stride loop).
template <typename T, bool RED>
__global__ __launch_bounds__(256) void pool3d_bwd_route_rows_kernel(const T* __restrict__ y, const float4* __restrict__ scale,
                                                                    const float4* __restrict__ shift, const uint8_t* __restrict__ mask,
                                                                    const T* __restrict__ pooled, const T* __restrict__ dpooled, int D, int C4,
                                                                    T* __restrict__ g, const int* __restrict__ out_pos,
                                                                    const int* __restrict__ out_count, float* __restrict__ partial) {
    __shared__ float sh[RED ? 256 : 1][8];
    const unsigned total = (unsigned)(*out_count) * (unsigned)C4;
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgy = sg;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned c = i % (unsigned)C4, pos = (unsigned)out_pos[i / (unsigned)C4];
        const PoolWin w = pool_window(mask, pos, (unsigned)D);
        const float4 pm = Act<T>::ld4(pooled + ((size_t)pos * C4 + c) * 4), dp = Act<T>::ld4(dpooled + ((size_t)pos * C4 + c) * 4);
        route_window<T>(y, scale[c], shift[c], w, (unsigned)D, (unsigned)C4, c, pm, dp, g, sg, sgy);
    }
    if (RED) {
        float* p = sh[threadIdx.x];
        p[0] = sg.x; p[1] = sg.y; p[2] = sg.z; p[3] = sg.w; p[4] = sgy.x; p[5] = sgy.y; p[6] = sgy.z; p[7] = sgy.w;
        __syncthreads();
        if ((int)threadIdx.x < C4) {
            const int c = threadIdx.x;
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = threadIdx.x; r < 256; r += C4)
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += sh[r][k];
            float* o = partial + (size_t)blockIdx.x * 2 * C4 * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[c * 4 + k] = a[k]; o[C4 * 4 + c * 4 + k] = a[4 + k]; }
        }
    }
}
static int route_rows_grid(int B, int D, int C) {
    long worst = (long)B * (D / 2) * (D / 2) * (D / 2) * (C / 4);
    return ew_grid(worst / 4 + 1);                                  // (sized for a quarter-full list; the loop is grid-strided)
}
extern "C" int tri_pool3d_bwd_route_rows(const void* y, const float* scale, const float* shift, const uint8_t* mask, const void* pooled,
                                         const void* dpooled, int B, int D, int C, void* g, const int* out_pos, const int* out_count,
                                         int act_fmt, void* stream) {
    if (!out_pos || !out_count) { tri_set_error("tri_pool3d_bwd_route_rows: row list required"); return TRI_ERR_ARG; }
    if (!pool_args_ok("tri_pool3d_bwd_route_rows", mask, B, D, C)) return TRI_ERR_ARG;
    TRI_ACT_DISPATCH(act_fmt, (pool3d_bwd_route_rows_kernel<T, false><<<route_rows_grid(B, D, C), 256, 0, (hipStream_t)stream>>>(
        (const T*)y, (const float4*)scale, (const float4*)shift, mask, (const T*)pooled, (const T*)dpooled, D, C / 4, (T*)g, out_pos, out_count,
        nullptr)));
    return tri_check_launch("tri_pool3d_bwd_route_rows");
}
// The same walk with the BatchNorm-backward sums folded in: partial [tri_pool3d_bwd_route_rows_num_blocks][2][C] is the input of
// tri_bn_bwd_finalize (then tri_bn_bwd_apply with the site mask, or tri_bn_bwd_apply_rows with this level's own list).  C / 4 must divide 256.
extern "C" int tri_pool3d_bwd_route_rows_num_blocks(int B, int D, int C) { return route_rows_grid(B, D, C); }
extern "C" int tri_pool3d_bwd_route_rows_reduce(const void* y, const float* scale, const float* shift, const uint8_t* mask, const void* pooled,
                                                const void* dpooled, int B, int D, int C, void* g, const int* out_pos, const int* out_count,
                                                float* partial, int act_fmt, void* stream) {
    if (!out_pos || !out_count || !partial) { tri_set_error("tri_pool3d_bwd_route_rows_reduce: row list and partial required"); return TRI_ERR_ARG; }
    if (C % 4 || C / 4 > 256 || 256 % (C / 4)) { tri_set_error("tri_pool3d_bwd_route_rows_reduce: C / 4 must divide 256"); return TRI_ERR_ARG; }
    if (!pool_args_ok("tri_pool3d_bwd_route_rows_reduce", mask, B, D, C)) return TRI_ERR_ARG;
    TRI_ACT_DISPATCH(act_fmt, (pool3d_bwd_route_rows_kernel<T, true><<<route_rows_grid(B, D, C), 256, 0, (hipStream_t)stream>>>(
        (const T*)y, (const float4*)scale, (const float4*)shift, mask, (const T*)pooled, (const T*)dpooled, D, C / 4, (T*)g, out_pos, out_count,
        partial)));
    return tri_check_launch("tri_pool3d_bwd_route_rows_reduce");
}
// BatchNorm backward over a compact row list (g is final: no ReLU mask; rows outside the list are neither read nor written - the
// keep_inactive contract of tri_bn_bwd_apply).  partial: [BNR_BLOCKS][2][C] scratch.
#define BNR_BLOCKS 512
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_rows_kernel(const T* __restrict__ y, const T* __restrict__ g, int C, const int* __restrict__ row_pos,
                                                                 const int* __restrict__ row_count, float* __restrict__ partial) {
    extern __shared__ float sh[];                                    // [rpp][tpr][8]
    const int C4 = C >> 2;
    const int tpr = C4 < 256 ? C4 : 256, rpp = 256 / tpr, cpt = (C4 + tpr - 1) / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const int n = *row_count;
    for (int cc = 0; cc < cpt; ++cc) {
        const int c4 = tc + cc * tpr;
        float4 sg = make_float4(0, 0, 0, 0), sgy = make_float4(0, 0, 0, 0);
        if (c4 < C4 && tr < rpp)
#pragma unroll 4
            for (int i = blockIdx.x * rpp + tr; i < n; i += gridDim.x * rpp) {
                const long r = row_pos[i];
                const float4 gv = Act<T>::ld4(g + r * C + c4 * 4), yv = Act<T>::ld4(y + r * C + c4 * 4);
                sg.x += gv.x; sg.y += gv.y; sg.z += gv.z; sg.w += gv.w;
                sgy.x += gv.x * yv.x; sgy.y += gv.y * yv.y; sgy.z += gv.z * yv.z; sgy.w += gv.w * yv.w;
            }
        __syncthreads();
        if (c4 < C4 && tr < rpp) {
            float* p = sh + ((size_t)tr * tpr + tc) * 8;
            p[0] = sg.x; p[1] = sg.y; p[2] = sg.z; p[3] = sg.w; p[4] = sgy.x; p[5] = sgy.y; p[6] = sgy.z; p[7] = sgy.w;
        }
        __syncthreads();
        if (tr == 0 && c4 < C4) {
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int rr = 0; rr < rpp; ++rr) {
                const float* p = sh + ((size_t)rr * tpr + tc) * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += p[k];
            }
            float* o = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[c4 * 4 + k] = a[k]; o[C + c4 * 4 + k] = a[4 + k]; }
        }
    }
}
template <typename T>
__global__ void bn_bwd_apply_rows_kernel(const T* __restrict__ y, const T* g, const float4* __restrict__ c1, const float4* __restrict__ c2,
                                         const float4* __restrict__ c3, T* dy, int C4, const int* __restrict__ row_pos,
                                         const int* __restrict__ row_count) {
    const long total = (long)(*row_count) * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const long o = ((long)row_pos[i / C4] * C4 + c) * 4;
        const float4 yv = Act<T>::ld4(y + o), gv = Act<T>::ld4(g + o), a = c1[c], b = c2[c], d = c3[c];
        Act<T>::st4(dy + o, make_float4(a.x * gv.x + b.x + d.x * yv.x, a.y * gv.y + b.y + d.y * yv.y, a.z * gv.z + b.z + d.z * yv.z,
                                        a.w * gv.w + b.w + d.w * yv.w));
    }
}
extern "C" size_t tri_bn_bwd_rows_scratch(int C) { return (size_t)BNR_BLOCKS * 2 * C * sizeof(float) + 3 * (size_t)C * sizeof(float); }
// (dy, dgamma, dbeta) of a BatchNorm whose g, y and dy are only defined on the rows of the list; max_rows = capacity of the list
extern "C" int tri_bn_bwd_rows(const void* y, const void* g, int C, const int* row_pos, const int* row_count, long max_rows, const float* gamma,
                               const float* mean, const float* invstd, void* dy, float* dgamma, float* dbeta, float out_scale, void* scratch,
                               int act_fmt, void* stream) {
    if (C % 4 || !row_pos || !row_count || !scratch) { tri_set_error("tri_bn_bwd_rows: C % 4, row list and scratch required"); return TRI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)scratch;
    float* co = partial + (size_t)BNR_BLOCKS * 2 * C;
    const int C4 = C / 4, tpr = C4 < 256 ? C4 : 256, rpp = 256 / tpr;
    const size_t smem = (size_t)rpp * tpr * 8 * sizeof(float);
    TRI_ACT_DISPATCH(act_fmt, bn_bwd_reduce_rows_kernel<T><<<BNR_BLOCKS, 256, smem, st>>>((const T*)y, (const T*)g, C, row_pos, row_count, partial));
    bn_bwd_finalize_kernel<<<C, 256, 0, st>>>(partial, BNR_BLOCKS, C, row_count, 0, gamma, mean, invstd, dgamma, dbeta, co, co + C, co + 2 * C, out_scale);
    const int grid = ew_grid(max_rows * C4 / 4 + 1);
    TRI_ACT_DISPATCH(act_fmt, bn_bwd_apply_rows_kernel<T><<<grid, 256, 0, st>>>((const T*)y, (const T*)g, (const float4*)co, (const float4*)(co + C),
                                                                                 (const float4*)(co + 2 * C), (T*)dy, C4, row_pos, row_count));
    return tri_check_launch("tri_bn_bwd_rows");
}

// dy = c1 * g + c2 + c3 * y on the rows of the list only (the third pass of tri_bn_bwd_rows on its own, for callers whose sums came
// from tri_pool3d_bwd_route_rows_reduce); dy may alias g.
extern "C" int tri_bn_bwd_apply_rows(const void* y, const void* g, const float* c1, const float* c2, const float* c3, void* dy, int C,
                                     const int* row_pos, const int* row_count, long max_rows, int act_fmt, void* stream) {
    if (C % 4 || !row_pos || !row_count) { tri_set_error("tri_bn_bwd_apply_rows: C % 4 and a row list required"); return TRI_ERR_ARG; }
    const int C4 = C / 4, grid = ew_grid(max_rows * C4 / 4 + 1);
    TRI_ACT_DISPATCH(act_fmt, bn_bwd_apply_rows_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)y, (const T*)g, (const float4*)c1,
        (const float4*)c2, (const float4*)c3, (T*)dy, C4, row_pos, row_count));
    return tri_check_launch("tri_bn_bwd_apply_rows");
}

// The same routing with the BatchNorm-backward sums of the level folded in: every routed value and the y it belongs to are in
// registers here, so the per-channel sums of g and g * y over the active sites (what tri_bn_bwd_reduce would re-read both tensors
// for) leave as one [2][C] record per workgroup for tri_bn_bwd_finalize - one launch less per level on the voxel tower's backward.
// Needs 256 % (C / 4) == 0 (a thread keeps its channel quad across the grid-stride loop).
template <typename T>
__global__ __launch_bounds__(256) void pool3d_bwd_route_reduce_kernel(const T* __restrict__ y, const float4* __restrict__ scale,
                                                                      const float4* __restrict__ shift, const uint8_t* __restrict__ mask,
                                                                      const T* __restrict__ pooled, const T* __restrict__ dpooled, int B, int D,
                                                                      int C4, T* __restrict__ g, float* __restrict__ partial) {
    __shared__ float sh[256][8];
    const int Do = D >> 1;
    const long total = (long)B * Do * Do * Do * C4;
    const int c = threadIdx.x % C4;                                  // (256 % C4 == 0: the quad is the same on every grid-stride pass)
    const float4 s = scale[c], t = shift[c];
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgy = sg;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        const PoolWin w = pool_window(mask, i / (unsigned)C4, (unsigned)D);
        if (!w.bits) continue;
        const float4 pm = Act<T>::ld4(pooled + (size_t)i * 4), dp = Act<T>::ld4(dpooled + (size_t)i * 4);
        route_window<T>(y, s, t, w, (unsigned)D, (unsigned)C4, (unsigned)c, pm, dp, g, sg, sgy);
    }
    float* p = sh[threadIdx.x];
    p[0] = sg.x; p[1] = sg.y; p[2] = sg.z; p[3] = sg.w; p[4] = sgy.x; p[5] = sgy.y; p[6] = sgy.z; p[7] = sgy.w;
    __syncthreads();
    if ((int)threadIdx.x < C4) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = threadIdx.x; r < 256; r += C4)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += sh[r][k];
        float* o = partial + (size_t)blockIdx.x * 2 * C4 * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[c * 4 + k] = a[k]; o[C4 * 4 + c * 4 + k] = a[4 + k]; }
    }
}
extern "C" int tri_pool3d_bwd_route_reduce_num_blocks(int B, int D, int C) {
    return ew_grid((long)B * (D / 2) * (D / 2) * (D / 2) * (C / 4));
}
// g as tri_pool3d_bwd_route; partial [tri_pool3d_bwd_route_reduce_num_blocks][2][C] = per-workgroup sums of g and g * y over the active
// sites, the input of tri_bn_bwd_finalize (then tri_bn_bwd_apply with the same site mask).  C / 4 must divide 256.
extern "C" int tri_pool3d_bwd_route_reduce(const void* y, const float* scale, const float* shift, const uint8_t* mask, const void* pooled,
                                           const void* dpooled, int B, int D, int C, void* g, float* partial, int act_fmt, void* stream) {
    if (C % 4 || C / 4 > 256 || 256 % (C / 4)) { tri_set_error("tri_pool3d_bwd_route_reduce: C / 4 must divide 256"); return TRI_ERR_ARG; }
    if (!pool_args_ok("tri_pool3d_bwd_route_reduce", mask, B, D, C)) return TRI_ERR_ARG;
    const int nblk = tri_pool3d_bwd_route_reduce_num_blocks(B, D, C);
    TRI_ACT_DISPATCH(act_fmt, pool3d_bwd_route_reduce_kernel<T><<<nblk, 256, 0, (hipStream_t)stream>>>(
        (const T*)y, (const float4*)scale, (const float4*)shift, mask, (const T*)pooled, (const T*)dpooled, B, D, C / 4, (T*)g, partial));
    return tri_check_launch("tri_pool3d_bwd_route_reduce");
}

// ----------------------------------------------------------------------------- ResNet stem: 3x3 / stride 2 / pad 1
// Forward also records, per output element, WHICH of the 9 window taps won (first maximum in (kh,kw) scan order, the
// torch.max_pool2d tie rule) as one byte; backward is then a gather over the <= 4 windows covering an input pixel:
// 4 byte reads + 4 float4 reads instead of re-scanning 36 inputs.  Deterministic, no atomics.
template <typename T>
__global__ void maxpool2d_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C4, T* __restrict__ out,
                                     uchar4* __restrict__ arg, const float4* __restrict__ scale, const float4* __restrict__ shift) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;          // floor((H + 2 - 3)/2) + 1
    const long total = (long)N * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4);
        long pos = i / C4;
        int ow = (int)(pos % Wo); long r = pos / Wo;
        int oh = (int)(r % Ho); int n = (int)(r / Ho);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 bi = make_uchar4(0, 0, 0, 0);
        float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) { s4 = scale[c]; b4 = shift[c]; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                float4 v = Act<T>::ld4(x + ((((long)n * H + ih) * W + iw) * C4 + c) * 4);
                if (scale) {                                       // pooling relu(bn(x)) without materialising it
                    v.x = fmaxf(__fmaf_rn(v.x, s4.x, b4.x), 0.f); v.y = fmaxf(__fmaf_rn(v.y, s4.y, b4.y), 0.f);
                    v.z = fmaxf(__fmaf_rn(v.z, s4.z, b4.z), 0.f); v.w = fmaxf(__fmaf_rn(v.w, s4.w, b4.w), 0.f);
                }
                unsigned char k = (unsigned char)(kh * 3 + kw);
                if (v.x > best.x) { best.x = v.x; bi.x = k; }
                if (v.y > best.y) { best.y = v.y; bi.y = k; }
                if (v.z > best.z) { best.z = v.z; bi.z = k; }
                if (v.w > best.w) { best.w = v.w; bi.w = k; }
            }
        }
        Act<T>::st4(out + i * 4, best);
        if (arg) arg[i] = bi;
    }
}
// 16-bit storage, C % 8 == 0: one thread = one output position x EIGHT channels (16-byte loads of the nine taps, one 16-byte store of
// the maxima, 8 bytes of the tap map); the 4-channel form above moves 8 bytes per load and ran at 3.3 TB/s on the stem's 100 MB tensor.
// Same scan order and tie rule per channel.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2d_fwd8_kernel(const T* __restrict__ x, int N, int H, int W, int C8, T* __restrict__ out,
                                                             uint2* __restrict__ arg, const float* __restrict__ scale, const float* __restrict__ shift) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    // One thread = POOL8_ROWS consecutive output rows of one (image, column, channel octet): output rows oh and oh + 1 share input row
    // 2 oh + 1, which stays in registers (6 loads per output instead of 9, and no second fetch of the shared row by another workgroup /
    // XCD: 152 MB of fabric reads per launch for the 100 MB tensor before).  (XCD-contiguous runs of workgroups - xcd_remap - with one
    // output per thread: 41 -> 54 us, not kept.)
    constexpr int G = 4;
    const int Hg = (Ho + G - 1) / G;
    const long total = (long)N * Hg * Wo * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        long pos = i / C8;
        const int ow = (int)(pos % Wo); long r = pos / Wo;
        const int og = (int)(r % Hg); const int n = (int)(r / Hg);
        float s8[8], b8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s8[k] = 1.f; b8[k] = 0.f; }
        if (scale) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { s8[k] = scale[c * 8 + k]; b8[k] = shift[c * 8 + k]; }
        }
        bool okw[3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) okw[kw] = (unsigned)(ow * 2 - 1 + kw) < (unsigned)W;
        auto load_row = [&](int ih, uint4 (&d)[3]) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                d[kw] = make_uint4(0u, 0u, 0u, 0u);
                if ((unsigned)ih < (unsigned)H && okw[kw]) d[kw] = *(const uint4*)(x + ((((long)n * H + ih) * W + ow * 2 - 1 + kw) * C8 + c) * 8);
            }
        };
        uint4 row0[3], row1[3], row2[3];
        const int oh0 = og * G;
        load_row(oh0 * 2 - 1, row0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int oh = oh0 + g;
            if (oh >= Ho) break;
            load_row(oh * 2, row1);
            load_row(oh * 2 + 1, row2);
            float best[8];
            unsigned bi[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0u; }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if ((unsigned)(oh * 2 - 1 + kh) >= (unsigned)H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    if (!okw[kw]) continue;
                    const uint4 q = kh == 0 ? row0[kw] : (kh == 1 ? row1[kw] : row2[kw]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float e = (float)((const T*)&q)[k];
                        if (scale) e = fmaxf(__fmaf_rn(e, s8[k], b8[k]), 0.f);
                        if (e > best[k]) { best[k] = e; bi[k] = (unsigned)(kh * 3 + kw); }
                    }
                }
            }
            uint4 o;
            T* op = (T*)&o;
#pragma unroll
            for (int k = 0; k < 8; ++k) op[k] = (T)best[k];
            const long oi = (((long)n * Ho + oh) * Wo + ow) * C8 + c;
            *(uint4*)(out + oi * 8) = o;
            if (arg) arg[oi] = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) row0[kw] = row2[kw];
        }
    }
}
// bn_scale / bn_shift (optional, [C]): pool relu(x * scale + shift) - the ResNet stem's BN + ReLU + max-pool in one pass
extern "C" int tri_maxpool2d_fwd(const void* x, int N, int H, int W, int C, void* out, uint8_t* arg, const float* bn_scale,
                                 const float* bn_shift, int act_fmt, void* stream) {
    static int pool8 = -1;                                         // A/B switch: TRICOLO_POOL8=0 keeps the 4-channel form
    if (pool8 < 0) pool8 = 1;
    if (pool8 && act_fmt != TRI_FMT_F32 && C % 8 == 0) {
        const long total8 = (long)N * (((H + 1) / 2 + 3) / 4) * ((W + 1) / 2) * (C / 8);   // (four output rows per thread)
        if (act_fmt == TRI_FMT_F16)
            maxpool2d_fwd8_kernel<f16_t><<<ew_grid(total8), 256, 0, (hipStream_t)stream>>>((const f16_t*)x, N, H, W, C / 8, (f16_t*)out, (uint2*)arg, bn_scale, bn_shift);
        else
            maxpool2d_fwd8_kernel<bf16_t><<<ew_grid(total8), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, N, H, W, C / 8, (bf16_t*)out, (uint2*)arg, bn_scale, bn_shift);
        return tri_check_launch("tri_maxpool2d_fwd");
    }
    long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    TRI_ACT_DISPATCH(act_fmt, maxpool2d_fwd_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const T*)x, N, H, W, C / 4, (T*)out, (uchar4*)arg,
                                                                                                  (const float4*)bn_scale, (const float4*)bn_shift));
    return tri_check_launch("tri_maxpool2d_fwd");
}

template <typename T>
__global__ void maxpool2d_bwd_kernel(const uchar4* __restrict__ arg, const T* __restrict__ dout, int N, int H, int W, int C4,
                                     T* __restrict__ dx) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4);
        long pos = i / C4;
        int w = (int)(pos % W); long r = pos / W;
        int h = (int)(r % H); int n = (int)(r / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oh = h / 2; oh <= (h + 1) / 2; ++oh) {
            if (oh >= Ho) continue;
            for (int ow = w / 2; ow <= (w + 1) / 2; ++ow) {
                if (ow >= Wo) continue;
                unsigned char me = (unsigned char)((h - (oh * 2 - 1)) * 3 + (w - (ow * 2 - 1)));   // my tap index in that window
                long o = (((long)n * Ho + oh) * Wo + ow) * C4 + c;
                uchar4 a = arg[o];
                float4 d = Act<T>::ld4(dout + o * 4);
                if (a.x == me) acc.x += d.x;
                if (a.y == me) acc.y += d.y;
                if (a.z == me) acc.z += d.z;
                if (a.w == me) acc.w += d.w;
            }
        }
        Act<T>::st4(dx + i * 4, acc);
    }
}
// Even H and W: one thread per 2x2 input block and channel quad.  The block's four pixels can only have won in the four
// windows (bh..bh+1, bw..bw+1), so 4 (arg, dout) pairs are loaded once instead of 9 for the four pixels separately.
template <typename T>
__global__ void maxpool2d_bwd2x2_kernel(const uchar4* __restrict__ arg, const T* __restrict__ dout, int N, int H, int W, int C4,
                                        T* __restrict__ dx) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long pos = i / C4;
        const int bw = (int)(pos % Wo); long r = pos / Wo;
        const int bh = (int)(r % Ho); const int n = (int)(r / Ho);
        uchar4 a[2][2];
        float4 d[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                a[u][v] = make_uchar4(255, 255, 255, 255);
                d[u][v] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bh + u < Ho && bw + v < Wo) {
                    const long o = (((long)n * Ho + bh + u) * Wo + bw + v) * C4 + c;
                    a[u][v] = arg[o];
                    d[u][v] = Act<T>::ld4(dout + o * 4);
                }
            }
#define TRI_PICK(A, D, TAP) make_float4(A.x == TAP ? D.x : 0.f, A.y == TAP ? D.y : 0.f, A.z == TAP ? D.z : 0.f, A.w == TAP ? D.w : 0.f)
#define TRI_ADD4(P, Q) make_float4(P.x + Q.x, P.y + Q.y, P.z + Q.z, P.w + Q.w)
        // tap index of pixel (h, w) in window (oh, ow) = (h - 2 oh + 1) * 3 + (w - 2 ow + 1)
        const float4 p00 = TRI_PICK(a[0][0], d[0][0], 4);                                                  // (2bh,   2bw)
        const float4 p01 = TRI_ADD4(TRI_PICK(a[0][0], d[0][0], 5), TRI_PICK(a[0][1], d[0][1], 3));         // (2bh,   2bw+1)
        const float4 p10 = TRI_ADD4(TRI_PICK(a[0][0], d[0][0], 7), TRI_PICK(a[1][0], d[1][0], 1));         // (2bh+1, 2bw)
        float4 p11 = TRI_ADD4(TRI_PICK(a[0][0], d[0][0], 8), TRI_PICK(a[0][1], d[0][1], 6));               // (2bh+1, 2bw+1), same window
        const float4 p11b = TRI_ADD4(TRI_PICK(a[1][0], d[1][0], 2), TRI_PICK(a[1][1], d[1][1], 0));        // order as the per-pixel kernel
        p11 = TRI_ADD4(p11, p11b);
#undef TRI_PICK
#undef TRI_ADD4
        T* o = dx + ((((long)n * H + 2 * bh) * W + 2 * bw) * C4 + c) * 4;
        Act<T>::st4(o, p00);
        Act<T>::st4(o + (long)C4 * 4, p01);
        Act<T>::st4(o + (long)W * C4 * 4, p10);
        Act<T>::st4(o + ((long)W + 1) * C4 * 4, p11);
    }
}
extern "C" int tri_maxpool2d_bwd(const uint8_t* arg, const void* dout, int N, int H, int W, int C, void* dx, int act_fmt, void* stream) {
    if (H % 2 == 0 && W % 2 == 0) {
        long total = (long)N * (H / 2) * (W / 2) * (C / 4);
        TRI_ACT_DISPATCH(act_fmt, maxpool2d_bwd2x2_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const uchar4*)arg, (const T*)dout, N, H, W, C / 4, (T*)dx));
        return tri_check_launch("tri_maxpool2d_bwd");
    }
    long total = (long)N * H * W * (C / 4);
    TRI_ACT_DISPATCH(act_fmt, maxpool2d_bwd_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const uchar4*)arg, (const T*)dout, N, H, W, C / 4, (T*)dx));
    return tri_check_launch("tri_maxpool2d_bwd");
}

// ---- stem backward without the max-pool backward pass ------------------------------------------------------------------
// conv 7x7/2 -> BN -> ReLU -> MaxPool2d(3, 2, 1) (mv_cnn.py:44, torchvision stem).  The gradient of relu(bn(y)) is the pooled
// gradient routed to each window's winning tap - recomputable from the byte map and dpool (1/4 the positions) on the fly.  So
// the BatchNorm backward reads (y, arg, dpool) instead of (y, dz): maxpool2d_bwd's pass (25 + 12 MB in, 100 MB out at the bench
// shape) and the two re-reads of its 100 MB output disappear.  One thread = one 2x2 block of y positions x 4 channels, as
// maxpool2d_bwd2x2_kernel: the block's pixels can only have won in the four windows (bh..bh+1, bw..bw+1).
template <typename T>
__device__ __forceinline__ void stem_route_2x2(const uchar4* __restrict__ arg, const T* __restrict__ dpool, int n, int bh, int bw, int Ho,
                                               int Wo, int C4, int c, float4 g[4]) {
    uchar4 a[2][2];
    float4 d[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            a[u][v] = make_uchar4(255, 255, 255, 255);
            d[u][v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bh + u < Ho && bw + v < Wo) {
                const long o = (((long)n * Ho + bh + u) * Wo + bw + v) * C4 + c;
                a[u][v] = arg[o];
                d[u][v] = Act<T>::ld4(dpool + o * 4);
            }
        }
#define TRI_PICK(A, D, TAP) make_float4(A.x == TAP ? D.x : 0.f, A.y == TAP ? D.y : 0.f, A.z == TAP ? D.z : 0.f, A.w == TAP ? D.w : 0.f)
#define TRI_ADD4(P, Q) make_float4(P.x + Q.x, P.y + Q.y, P.z + Q.z, P.w + Q.w)
    g[0] = TRI_PICK(a[0][0], d[0][0], 4);                                                  // (2bh,   2bw)
    g[1] = TRI_ADD4(TRI_PICK(a[0][0], d[0][0], 5), TRI_PICK(a[0][1], d[0][1], 3));         // (2bh,   2bw+1)
    g[2] = TRI_ADD4(TRI_PICK(a[0][0], d[0][0], 7), TRI_PICK(a[1][0], d[1][0], 1));         // (2bh+1, 2bw)
    const float4 p = TRI_ADD4(TRI_PICK(a[0][0], d[0][0], 8), TRI_PICK(a[0][1], d[0][1], 6));
    const float4 q = TRI_ADD4(TRI_PICK(a[1][0], d[1][0], 2), TRI_PICK(a[1][1], d[1][1], 0));
    g[3] = TRI_ADD4(p, q);                                                                 // (2bh+1, 2bw+1): summation order of maxpool2d_bwd
#undef TRI_PICK
#undef TRI_ADD4
}
// values of the routed gradient are rounded to the storage type exactly where maxpool2d_bwd would have stored them
template <typename T>
__device__ __forceinline__ float4 rnd4(float4 v) { return make_float4(Act<T>::rnd(v.x), Act<T>::rnd(v.y), Act<T>::rnd(v.z), Act<T>::rnd(v.w)); }

// 2x2 blocks (4 positions each) per workgroup: 64 as bnb_rows of large tensors; tuning aid TRICOLO_STEM_BLOCKS
static int stem_blocks_per_wg() {
    static int v = -1;
    if (v < 0) v = 64;
    return v;
}
// 16-bit storage, C % 8 == 0: one thread = one 2x2 block of positions x EIGHT channels - 16-byte loads of y and the pooled gradient,
// 8-byte loads of the tap map (the 4-channel form below moves 8 / 4 bytes per load and ran at 2.1 TB/s on the stem's 100 MB tensor).
// Same routing, rounding, mask and summation order per channel as the 4-channel form.
template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_reduce8_kernel(const T* __restrict__ y, const uint8_t* __restrict__ arg, const T* __restrict__ dpool,
                                                               int N, int H, int W, int C, float* __restrict__ partial,
                                                               const float* __restrict__ rs, const float* __restrict__ rb, int STEM_BLOCKS_PER_WG) {
    extern __shared__ float sh[];                                  // [items_per_pass][C8][16]
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const int tpr = C8 < 256 ? C8 : 256, rpp = 256 / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const long nb = (long)N * Ho * Wo;
    const long r0 = (long)blockIdx.x * STEM_BLOCKS_PER_WG, r1 = r0 + STEM_BLOCKS_PER_WG < nb ? r0 + STEM_BLOCKS_PER_WG : nb;
    float sg[8], sgy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sg[k] = 0.f; sgy[k] = 0.f; }
    if (tc < C8 && tr < rpp) {
        float s8[8], b8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s8[k] = rs[tc * 8 + k]; b8[k] = rb[tc * 8 + k]; }
#pragma unroll 2
        for (long r = r0 + tr; r < r1; r += rpp) {
            const int bw = (int)(r % Wo); const long q = r / Wo;
            const int bh = (int)(q % Ho), n = (int)(q / Ho);
            uint4 yr[4], dr[2][2];
            uint2 ar[2][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) yr[k] = *(const uint4*)(y + ((((long)n * H + 2 * bh + (k >> 1)) * W + 2 * bw + (k & 1)) * C + tc * 8));
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    ar[u][v] = make_uint2(0xffffffffu, 0xffffffffu);
                    dr[u][v] = make_uint4(0u, 0u, 0u, 0u);
                    if (bh + u < Ho && bw + v < Wo) {
                        const long o = (((long)n * Ho + bh + u) * Wo + bw + v) * C + tc * 8;
                        ar[u][v] = *(const uint2*)(arg + o);
                        dr[u][v] = *(const uint4*)(dpool + o);
                    }
                }
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                float d[2][2];
                unsigned a[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        d[u][v] = (float)((const T*)&dr[u][v])[ch];
                        a[u][v] = ((ch < 4 ? ar[u][v].x : ar[u][v].y) >> (8 * (ch & 3))) & 255u;
                    }
#define TRI_PK(U, V, TAP) (a[U][V] == (TAP) ? d[U][V] : 0.f)
                float g[4];
                g[0] = TRI_PK(0, 0, 4u);
                g[1] = TRI_PK(0, 0, 5u) + TRI_PK(0, 1, 3u);
                g[2] = TRI_PK(0, 0, 7u) + TRI_PK(1, 0, 1u);
                g[3] = (TRI_PK(0, 0, 8u) + TRI_PK(0, 1, 6u)) + (TRI_PK(1, 0, 2u) + TRI_PK(1, 1, 0u));
#undef TRI_PK
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float yv = (float)((const T*)&yr[k])[ch];
                    float gv = Act<T>::rnd(g[k]);
                    gv = __fmaf_rn(yv, s8[ch], b8[ch]) > 0.f ? gv : 0.f;
                    sg[ch] += gv;
                    sgy[ch] += gv * yv;
                }
            }
        }
        float* p = sh + ((size_t)tr * tpr + tc) * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) { p[k] = sg[k]; p[8 + k] = sgy[k]; }
    }
    __syncthreads();
    if (tr == 0 && tc < C8) {
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = 0.f;
        for (int rr = 0; rr < rpp; ++rr) {
            const float* p = sh + ((size_t)rr * tpr + tc) * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] += p[k];
        }
        float* o = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[tc * 8 + k] = a[k]; o[C + tc * 8 + k] = a[8 + k]; }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_reduce_kernel(const T* __restrict__ y, const uchar4* __restrict__ arg, const T* __restrict__ dpool,
                                                              int N, int H, int W, int C, float* __restrict__ partial,
                                                              const float4* __restrict__ rs, const float4* __restrict__ rb, int STEM_BLOCKS_PER_WG) {
    extern __shared__ float sh[];                                  // [items_per_pass][C4][8]
    const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const int tpr = C4 < 256 ? C4 : 256, rpp = 256 / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const long nb = (long)N * Ho * Wo;
    const long r0 = (long)blockIdx.x * STEM_BLOCKS_PER_WG, r1 = r0 + STEM_BLOCKS_PER_WG < nb ? r0 + STEM_BLOCKS_PER_WG : nb;
    float4 sg = make_float4(0, 0, 0, 0), sgy = make_float4(0, 0, 0, 0);
    if (tc < C4 && tr < rpp) {
        const float4 s4 = rs[tc], b4 = rb[tc];
#pragma unroll 4
        for (long r = r0 + tr; r < r1; r += rpp) {                          // (4 iterations at 64 channels: all 48 loads in flight at once)
            const int bw = (int)(r % Wo); const long q = r / Wo;
            const int bh = (int)(q % Ho), n = (int)(q / Ho);
            float4 g[4];
            stem_route_2x2<T>(arg, dpool, n, bh, bw, Ho, Wo, C4, tc, g);
            const T* yp = y + ((((long)n * H + 2 * bh) * W + 2 * bw) * C4 + tc) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 yv = Act<T>::ld4(yp + ((long)(k >> 1) * W + (k & 1)) * C4 * 4);
                float4 gv = rnd4<T>(g[k]);
                gv.x = __fmaf_rn(yv.x, s4.x, b4.x) > 0.f ? gv.x : 0.f; gv.y = __fmaf_rn(yv.y, s4.y, b4.y) > 0.f ? gv.y : 0.f;
                gv.z = __fmaf_rn(yv.z, s4.z, b4.z) > 0.f ? gv.z : 0.f; gv.w = __fmaf_rn(yv.w, s4.w, b4.w) > 0.f ? gv.w : 0.f;
                sg.x += gv.x; sg.y += gv.y; sg.z += gv.z; sg.w += gv.w;
                sgy.x += gv.x * yv.x; sgy.y += gv.y * yv.y; sgy.z += gv.z * yv.z; sgy.w += gv.w * yv.w;
            }
        }
        float* p = sh + ((size_t)tr * tpr + tc) * 8;
        p[0] = sg.x; p[1] = sg.y; p[2] = sg.z; p[3] = sg.w; p[4] = sgy.x; p[5] = sgy.y; p[6] = sgy.z; p[7] = sgy.w;
    }
    __syncthreads();
    if (tr == 0 && tc < C4) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int rr = 0; rr < rpp; ++rr) {
            const float* p = sh + ((size_t)rr * tpr + tc) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += p[k];
        }
        float* o = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[tc * 4 + k] = a[k]; o[C + tc * 4 + k] = a[4 + k]; }
    }
}
template <typename T>
__global__ void stem_bwd_apply_kernel(const T* __restrict__ y, const uchar4* __restrict__ arg, const T* __restrict__ dpool, int N, int H, int W,
                                      int C4, const float4* __restrict__ c1, const float4* __restrict__ c2, const float4* __restrict__ c3,
                                      const float4* __restrict__ rs, const float4* __restrict__ rb, T* __restrict__ dy) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)N * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long pos = i / C4;
        const int bw = (int)(pos % Wo); const long q = pos / Wo;
        const int bh = (int)(q % Ho), n = (int)(q / Ho);
        float4 g[4];
        stem_route_2x2<T>(arg, dpool, n, bh, bw, Ho, Wo, C4, c, g);
        const float4 a = c1[c], b = c2[c], d = c3[c], s4 = rs[c], b4 = rb[c];
        const long base = ((((long)n * H + 2 * bh) * W + 2 * bw) * C4 + c) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long o = base + ((long)(k >> 1) * W + (k & 1)) * C4 * 4;
            const float4 yv = Act<T>::ld4(y + o);
            float4 gv = rnd4<T>(g[k]);
            gv.x = __fmaf_rn(yv.x, s4.x, b4.x) > 0.f ? gv.x : 0.f; gv.y = __fmaf_rn(yv.y, s4.y, b4.y) > 0.f ? gv.y : 0.f;
            gv.z = __fmaf_rn(yv.z, s4.z, b4.z) > 0.f ? gv.z : 0.f; gv.w = __fmaf_rn(yv.w, s4.w, b4.w) > 0.f ? gv.w : 0.f;
            // (explicit FMA chain: conv_stem_wgrad_kernel's BNF staging forms the same values, bit for bit - conv_wgrad.hip stem_dy8)
            Act<T>::st4(dy + o, make_float4(fp32_rounded(__fmaf_rn(d.x, yv.x, __fmaf_rn(a.x, gv.x, b.x))),
                                            fp32_rounded(__fmaf_rn(d.y, yv.y, __fmaf_rn(a.y, gv.y, b.y))),
                                            fp32_rounded(__fmaf_rn(d.z, yv.z, __fmaf_rn(a.z, gv.z, b.z))),
                                            fp32_rounded(__fmaf_rn(d.w, yv.w, __fmaf_rn(a.w, gv.w, b.w)))));
        }
    }
}
extern "C" int tri_maxpool_bn_bwd_num_blocks(int N, int H, int W) {
    const int per = stem_blocks_per_wg();
    return (int)(((long)N * (H / 2) * (W / 2) + per - 1) / per);
}
static int stem_args_ok(int H, int W, int C, const void* rs, const void* rb) {
    if (H % 2 || W % 2 || C % 4 || C / 4 > 256) { tri_set_error("tri_maxpool_bn_bwd: needs even H, W and C % 4 == 0, C <= 1024"); return 0; }
    if (!rs || !rb) { tri_set_error("tri_maxpool_bn_bwd: the forward's BN scale / shift are required (ReLU mask)"); return 0; }
    return 1;
}
// partial [tri_maxpool_bn_bwd_num_blocks][2][C]: per-workgroup sums of g and g*y, g = gradient of relu(bn(y)) before the max-pool;
// finish with tri_bn_bwd_finalize (count = N*H*W) and tri_maxpool_bn_bwd_apply
extern "C" int tri_maxpool_bn_bwd_reduce(const void* y, const uint8_t* arg, const void* dpool, int N, int H, int W, int C, float* partial,
                                         const float* relu_scale, const float* relu_shift, int act_fmt, void* stream) {
    if (!stem_args_ok(H, W, C, relu_scale, relu_shift)) return TRI_ERR_ARG;
    const int nblk = tri_maxpool_bn_bwd_num_blocks(N, H, W);
    if (act_fmt != TRI_FMT_F32 && C % 8 == 0 && (C / 8 >= 256 || 256 % (C / 8) == 0)) {
        const int C8 = C / 8, tpr8 = C8 < 256 ? C8 : 256, rpp8 = 256 / tpr8;
        const size_t smem8 = (size_t)rpp8 * tpr8 * 16 * sizeof(float);
        if (act_fmt == TRI_FMT_F16)
            stem_bwd_reduce8_kernel<f16_t><<<nblk, 256, smem8, (hipStream_t)stream>>>((const f16_t*)y, arg, (const f16_t*)dpool, N, H, W, C, partial,
                                                                                     relu_scale, relu_shift, stem_blocks_per_wg());
        else
            stem_bwd_reduce8_kernel<bf16_t><<<nblk, 256, smem8, (hipStream_t)stream>>>((const bf16_t*)y, arg, (const bf16_t*)dpool, N, H, W, C, partial,
                                                                                      relu_scale, relu_shift, stem_blocks_per_wg());
        return tri_check_launch("tri_maxpool_bn_bwd_reduce");
    }
    const int C4 = C / 4, tpr = C4 < 256 ? C4 : 256, rpp = 256 / tpr;
    const size_t smem = (size_t)rpp * tpr * 8 * sizeof(float);
    TRI_ACT_DISPATCH(act_fmt, stem_bwd_reduce_kernel<T><<<nblk, 256, smem, (hipStream_t)stream>>>(
        (const T*)y, (const uchar4*)arg, (const T*)dpool, N, H, W, C, partial, (const float4*)relu_scale, (const float4*)relu_shift,
        stem_blocks_per_wg()));
    return tri_check_launch("tri_maxpool_bn_bwd_reduce");
}
// ---- the same two sums from the POOLED tensors (round 4) -----------------------------------------------------------------------
// g - the gradient of relu(bn(y)) - is the pooled gradient routed to each window's winning tap, so sum(g) and sum(g * y) are sums over
// WINDOWS: sum_w dpool[w] [p[w] > 0] and sum_w dpool[w] [p[w] > 0] y[winner(w)], where p = maxpool(relu(bn(y))) is the tensor the next
// layer kept anyway and, for an active window, y[winner] = (p - shift) / scale.  The pass then reads 2 x 25 MB of pooled-resolution
// tensors instead of y (100 MB) + tap map + dpool: 45 -> ~12 us at the bench shape.  A window whose recovered y would be ill-conditioned
// (|p| > 64 |gamma| for f16 storage, 8 |gamma| for bf16: a channel with gamma ~ 0 and a large shift; never at initialisation) reads the winner's stored y through the tap
// map instead, so every channel keeps the accuracy of the stored activations.  The routed gradient is no longer rounded to the
// storage type per position (the reference's max-pool backward has no such rounding either).
#define STEM_POOLED_PER_WG 256                                      // pooled positions per workgroup
template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_reduce_pooled_kernel(const T* __restrict__ pooled, const T* __restrict__ dpool, const T* __restrict__ y,
                                                                     const uint8_t* __restrict__ arg, int N, int H, int W, int C,
                                                                     float* __restrict__ partial, const float* __restrict__ rs,
                                                                     const float* __restrict__ rb, const float* __restrict__ gamma) {
    extern __shared__ float sh[];                                  // [rows per pass][C8][16]
    const int C8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const int tpr = C8 < 256 ? C8 : 256, rpp = 256 / tpr;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const long nb = (long)N * Ho * Wo;
    const long r0 = (long)blockIdx.x * STEM_POOLED_PER_WG, r1 = r0 + STEM_POOLED_PER_WG < nb ? r0 + STEM_POOLED_PER_WG : nb;
    float sg[8], sgy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sg[k] = 0.f; sgy[k] = 0.f; }
    if (tc < C8 && tr < rpp) {
        float s8[8], inv8[8], b8[8], lim8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float s_ = rs[tc * 8 + k];
            b8[k] = rb[tc * 8 + k];
            s8[k] = s_ != 0.f ? s_ : 1.f;
            inv8[k] = 1.0f / s8[k];
            // |p| above this: the stored y.  p is rounded to the storage type (relative error 2^-9 for bf16, 2^-11 for f16), so the recovered
            // xhat = (p - shift) / scale is off by up to |p| / |gamma| rounding units: the threshold keeps that under 2^-6 of a standard
            // deviation for both types (f16: 64 |gamma|, bf16: 8 |gamma| - ADVICE r4: 64 let trained-like bf16 channels with
            // |beta / gamma| ~ 10 drift by 0.02 in xhat)
            lim8[k] = s_ != 0.f ? (Act<T>::SIG_BITS <= 8 ? 8.f : 64.f) * fabsf(gamma[tc * 8 + k]) : -1.f;
        }
        // four rows per pass, all eight loads issued before the first is used; branch-free per channel (the quotient by one
        // reciprocal + one residual correction: exact whenever (p - shift) / scale is representable), ONE rare branch per row for
        // the windows that must read the stored y
        constexpr int U = 4;
        for (long rb0 = r0 + tr; rb0 < r1; rb0 += (long)U * rpp) {
            uint4 prs[U], drs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long r = rb0 + (long)u * rpp;
                prs[u] = make_uint4(0u, 0u, 0u, 0u); drs[u] = prs[u];         // (p = 0: an inactive window)
                if (r < r1) { prs[u] = *(const uint4*)(pooled + r * C + tc * 8); drs[u] = *(const uint4*)(dpool + r * C + tc * 8); }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint4 pr = prs[u], dr = drs[u];
                float yv8[8], d8[8];
                bool far = false;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    const float pv = (float)((const T*)&pr)[ch];
                    const float num = pv - b8[ch];
                    float qv = num * inv8[ch];
                    qv = __fmaf_rn(__fmaf_rn(-qv, s8[ch], num), inv8[ch], qv);
                    yv8[ch] = qv;
                    d8[ch] = pv > 0.f ? (float)((const T*)&dr)[ch] : 0.f;
                    far = far || (pv > 0.f && pv > lim8[ch]);
                }
                if (far) {                                         // (rare) the winners' stored activations
                    const long r = rb0 + (long)u * rpp;
                    const int ow = (int)(r % Wo); const long q = r / Wo;
                    const int oh = (int)(q % Ho), n = (int)(q / Ho);
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
                        const float pv = (float)((const T*)&pr)[ch];
                        if (pv > 0.f && pv > lim8[ch]) {
                            const int tap = arg[r * C + tc * 8 + ch];
                            yv8[ch] = (float)y[(((long)n * H + 2 * oh - 1 + tap / 3) * W + 2 * ow - 1 + tap % 3) * C + tc * 8 + ch];
                        }
                    }
                }
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) { sg[ch] += d8[ch]; sgy[ch] += d8[ch] * yv8[ch]; }
            }
        }
        float* q = sh + ((size_t)tr * tpr + tc) * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) { q[k] = sg[k]; q[8 + k] = sgy[k]; }
    }
    __syncthreads();
    if (tr == 0 && tc < C8) {
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = 0.f;
        for (int rr = 0; rr < rpp; ++rr) {
            const float* q = sh + ((size_t)rr * tpr + tc) * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] += q[k];
        }
        float* o = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[tc * 8 + k] = a[k]; o[C + tc * 8 + k] = a[8 + k]; }
    }
}
extern "C" int tri_maxpool_bn_bwd_pooled_num_blocks(int N, int H, int W) {
    return (int)(((long)N * (H / 2) * (W / 2) + STEM_POOLED_PER_WG - 1) / STEM_POOLED_PER_WG);
}
// partial [tri_maxpool_bn_bwd_pooled_num_blocks][2][C]; 16-bit storage, C % 8 == 0, C / 8 a divisor of 256 (else TRI_ERR_UNSUPPORTED:
// use tri_maxpool_bn_bwd_reduce).  pooled = the forward's max-pool output, gamma = the BatchNorm weight.
extern "C" int tri_maxpool_bn_bwd_reduce_pooled(const void* pooled, const void* dpool, const void* y, const uint8_t* arg, int N, int H, int W, int C,
                                                float* partial, const float* relu_scale, const float* relu_shift, const float* gamma,
                                                int act_fmt, void* stream) {
    if (!stem_args_ok(H, W, C, relu_scale, relu_shift)) return TRI_ERR_ARG;
    if (act_fmt == TRI_FMT_F32 || C % 8 || 256 % (C / 8) || !gamma || !arg || !y) {
        tri_set_error("tri_maxpool_bn_bwd_reduce_pooled: 16-bit storage with C / 8 a divisor of 256 only");
        return TRI_ERR_UNSUPPORTED;
    }
    const int nblk = tri_maxpool_bn_bwd_pooled_num_blocks(N, H, W);
    const size_t smem = (size_t)256 * 16 * sizeof(float);
    if (act_fmt == TRI_FMT_F16)
        stem_bwd_reduce_pooled_kernel<f16_t><<<nblk, 256, smem, (hipStream_t)stream>>>((const f16_t*)pooled, (const f16_t*)dpool, (const f16_t*)y, arg, N, H, W,
                                                                                      C, partial, relu_scale, relu_shift, gamma);
    else
        stem_bwd_reduce_pooled_kernel<bf16_t><<<nblk, 256, smem, (hipStream_t)stream>>>((const bf16_t*)pooled, (const bf16_t*)dpool, (const bf16_t*)y, arg, N, H,
                                                                                       W, C, partial, relu_scale, relu_shift, gamma);
    return tri_check_launch("tri_maxpool_bn_bwd_reduce_pooled");
}
extern "C" int tri_maxpool_bn_bwd_apply(const void* y, const uint8_t* arg, const void* dpool, int N, int H, int W, int C, const float* c1,
                                        const float* c2, const float* c3, const float* relu_scale, const float* relu_shift, void* dy,
                                        int act_fmt, void* stream) {
    if (!stem_args_ok(H, W, C, relu_scale, relu_shift)) return TRI_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    TRI_ACT_DISPATCH(act_fmt, stem_bwd_apply_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(
        (const T*)y, (const uchar4*)arg, (const T*)dpool, N, H, W, C / 4, (const float4*)c1, (const float4*)c2, (const float4*)c3,
        (const float4*)relu_scale, (const float4*)relu_shift, (T*)dy));
    return tri_check_launch("tri_maxpool_bn_bwd_apply");
}

// ----------------------------------------------------- global average pool + max over the views of one shape
// x [B*V, HW, C] -> out [B, C], argmax view index [B, C] (first maximum, as torch.max(dim=1))
// block = (shape b, 128 channels): 32 channel quads x 8 view slots; a thread sums its views over HW in the k order of a plain loop
// (same fp32 result as one thread per (b, c) - which ran 96 dependent-latency loads per thread on 64 workgroups, 25 us for 3 MB),
// then the 8 slots are merged through LDS with torch.max's tie rule (first maximum).
template <typename T>
__global__ __launch_bounds__(256) void avgpool_viewmax_fwd_kernel(const T* __restrict__ x, int B, int V, int HW, int C, float* __restrict__ out,
                                                                  int* __restrict__ arg) {
    __shared__ float4 sb[8][32];
    __shared__ int4 si[8][32];
    const int q = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const int b = blockIdx.x, c = blockIdx.y * 128 + q * 4;
    const float inv = 1.0f / (float)HW;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int4 bi = make_int4(0, 0, 0, 0);
    if (c < C) {
        for (int v = slot; v < V; v += 8) {
            const T* p = x + ((long)(b * V + v) * HW) * C + c;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8                                                   // (rolled: one load round trip per pixel, 16 in a row for a 4x4 map)
            for (int k = 0; k < HW; ++k) {
                const float4 e = Act<T>::ld4(p + (long)k * C);
                s.x += e.x; s.y += e.y; s.z += e.z; s.w += e.w;
            }
            s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
            if (s.x > best.x) { best.x = s.x; bi.x = v; }
            if (s.y > best.y) { best.y = s.y; bi.y = v; }
            if (s.z > best.z) { best.z = s.z; bi.z = v; }
            if (s.w > best.w) { best.w = s.w; bi.w = v; }
        }
    }
    sb[slot][q] = best; si[slot][q] = bi;
    __syncthreads();
    if (slot == 0 && c < C) {
#pragma unroll
        for (int s2 = 1; s2 < 8; ++s2) {
            const float4 o = sb[s2][q];
            const int4 oi = si[s2][q];
            // a slot's candidate is its FIRST maximum; between slots the smaller view index wins a tie
            if (o.x > best.x || (o.x == best.x && o.x > -INFINITY && oi.x < bi.x)) { best.x = o.x; bi.x = oi.x; }
            if (o.y > best.y || (o.y == best.y && o.y > -INFINITY && oi.y < bi.y)) { best.y = o.y; bi.y = oi.y; }
            if (o.z > best.z || (o.z == best.z && o.z > -INFINITY && oi.z < bi.z)) { best.z = o.z; bi.z = oi.z; }
            if (o.w > best.w || (o.w == best.w && o.w > -INFINITY && oi.w < bi.w)) { best.w = o.w; bi.w = oi.w; }
        }
        *(float4*)(out + (long)b * C + c) = best;
        *(int4*)(arg + (long)b * C + c) = bi;
    }
}
extern "C" int tri_avgpool_viewmax_fwd(const void* x, int B, int V, int HW, int C, float* out, int* arg, int act_fmt, void* stream) {
    if (C % 4) { tri_set_error("tri_avgpool_viewmax_fwd: C must be a multiple of 4"); return TRI_ERR_ARG; }
    TRI_ACT_DISPATCH(act_fmt, avgpool_viewmax_fwd_kernel<T><<<dim3(B, (C + 127) / 128), 256, 0, (hipStream_t)stream>>>((const T*)x, B, V, HW, C, out, arg));
    return tri_check_launch("tri_avgpool_viewmax_fwd");
}
template <typename T>
__global__ void avgpool_viewmax_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ arg, int B, int V, int HW, int C,
                                           T* __restrict__ dx, float scale) {
    const long total = (long)B * V * HW * C;
    float inv = scale / (float)HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long r = i / C / HW;
        int v = (int)(r % V), b = (int)(r / V);
        dx[i] = (T)((arg[(long)b * C + c] == v) ? dout[(long)b * C + c] * inv : 0.f);
    }
}
extern "C" int tri_avgpool_viewmax_bwd(const float* dout, const int* arg, int B, int V, int HW, int C, void* dx, int act_fmt, float scale,
                                       void* stream) {
    long total = (long)B * V * HW * C;
    TRI_ACT_DISPATCH(act_fmt, avgpool_viewmax_bwd_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(dout, arg, B, V, HW, C, (T*)dx, scale));
    return tri_check_launch("tri_avgpool_viewmax_bwd");
}
