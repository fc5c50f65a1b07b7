// Fused NT-Xent / InfoNCE loss, forward + backward, for gfx950.
//
// Replaces the ~15 torch launches per modality pair of /root/reference/tricolo/loss/nt_xent.py:55-74
// (2x normalize, eye, 2x matmul, 2x div, 2x log_softmax, 2x mul/sum, axpy) and their autograd graph with five
// small launches: row-normalise (x2), S = a_hat b_hat^T / T in exact fp32 (LDS-tiled FMA: the all-pairs matrix is
// tiny, parity matters more than MFMA here), row/column log-sum-exp with wave shuffles, scalar loss, and the
// gradient  dS = [alpha (softmax_row - I) + (1 - alpha)(softmax_col - I)] / B  contracted on the fly with the
// other modality and pushed through the normalise Jacobian - S^T is never materialised or recomputed.
#include "common.h"
#include "../../include/tricolo_hip.h"

// S[i][j] = <a_i, b_j> * inv_T ; 64x64 tile per block, 4x4 outputs per thread, K chunks of 16
__global__ __launch_bounds__(256) void ntx_sim_kernel(const float* __restrict__ a, const float* __restrict__ b, int B, int D, float inv_T,
                                                      float* __restrict__ S) {
    __shared__ float sa[16][65], sb[16][65];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < D; k0 += 16) {
        for (int e = t; e < 64 * 16; e += 256) {
            int r = e >> 4, k = e & 15;
            sa[k][r] = (i0 + r < B && k0 + k < D) ? a[(long)(i0 + r) * D + k0 + k] : 0.f;
            sb[k][r] = (j0 + r < B && k0 + k < D) ? b[(long)(j0 + r) * D + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = sa[k][ty * 4 + u]; bv[u] = sb[k][tx * 4 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) acc[u][w] = fmaf(av[u], bv[w], acc[u][w]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            int i = i0 + ty * 4 + u, j = j0 + tx * 4 + w;
            if (i < B && j < B) S[(long)i * B + j] = acc[u][w] * inv_T;
        }
}

// Small batches (the per-GPU batch of the training step): the 64x64 tiling above would run on one or a handful of CUs.
// One block per row i keeps a_i in LDS; each wave takes columns j = wave, wave + 4, ... four at a time (16-byte loads of
// b_j along D, 8 independent loads in flight), reduces with wave shuffles.  D % 4 == 0.
__global__ __launch_bounds__(256) void ntx_sim_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, int B, int D,
                                                           float inv_T, float* __restrict__ S) {
    extern __shared__ float sa[];
    const int i = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int k = t * 4; k < D; k += 1024) *(float4*)(sa + k) = *(const float4*)(a + (long)i * D + k);
    __syncthreads();
    for (int jb = wave * 4; jb < B; jb += 16) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = lane * 4; k < D; k += 256) {
            const float4 av = *(const float4*)(sa + k);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = jb + u < B ? jb + u : B - 1;
                const float4 bv = *(const float4*)(b + (long)j * D + k);
                acc[u] = fmaf(av.x, bv.x, fmaf(av.y, bv.y, fmaf(av.z, bv.z, fmaf(av.w, bv.w, acc[u]))));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = wave_sum(acc[u]);
            if (lane == 0 && jb + u < B) S[(long)i * B + jb + u] = v * inv_T;
        }
    }
}

__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}

// one block per index i: lse over row i and over column i, and that row's loss contribution
__global__ __launch_bounds__(256) void ntx_lse_kernel(const float* __restrict__ S, int B, float alpha, float* __restrict__ lse_row,
                                                      float* __restrict__ lse_col, float* __restrict__ rowloss) {
    __shared__ float sh[4];
    const int i = blockIdx.x, t = threadIdx.x;
    float mr = -INFINITY, mc = -INFINITY;
    for (int j = t; j < B; j += 256) { mr = fmaxf(mr, S[(long)i * B + j]); mc = fmaxf(mc, S[(long)j * B + i]); }
    mr = block_max(mr, sh);
    mc = block_max(mc, sh);
    float sr = 0.f, sc = 0.f;
    for (int j = t; j < B; j += 256) { sr += expf(S[(long)i * B + j] - mr); sc += expf(S[(long)j * B + i] - mc); }
    sr = block_sum(sr, sh);
    sc = block_sum(sc, sh);
    if (t == 0) {
        float lr = mr + logf(sr), lc = mc + logf(sc), d = S[(long)i * B + i];
        lse_row[i] = lr;
        lse_col[i] = lc;
        rowloss[i] = -(alpha * (d - lr) + (1.f - alpha) * (d - lc)) / (float)B;
    }
}

__global__ void ntx_loss_sum_kernel(const float* __restrict__ rowloss, int B, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < B; ++i) s += (double)rowloss[i];      // fixed order: reproducible
        *loss = (float)s;
    }
}

// block (i, side): side 0 -> d a_i, side 1 -> d b_i
__global__ __launch_bounds__(256) void ntx_grad_kernel(const float* __restrict__ S, const float* __restrict__ lse_row,
                                                       const float* __restrict__ lse_col, const float* __restrict__ ahat,
                                                       const float* __restrict__ bhat, const float* __restrict__ na,
                                                       const float* __restrict__ nb, int B, int D, float inv_T, float alpha, int norm,
                                                       float eps, float* __restrict__ dza, float* __restrict__ dzb,
                                                       const float* __restrict__ dloss) {
    extern __shared__ float w[];                 // [B] coefficients dS_ij (side 0) or dS_ji (side 1), times 1/T
    __shared__ float sh[4];
    const int i = blockIdx.x, side = blockIdx.y, t = threadIdx.x;
    const float invB = (dloss ? *dloss : 1.0f) / (float)B;      // upstream d(total)/d(loss) folded into the coefficients
    for (int j = t; j < B; j += 256) {
        float s, pr, pc;
        if (side == 0) { s = S[(long)i * B + j]; pr = expf(s - lse_row[i]); pc = expf(s - lse_col[j]); }
        else           { s = S[(long)j * B + i]; pr = expf(s - lse_row[j]); pc = expf(s - lse_col[i]); }
        float delta = (i == j) ? 1.f : 0.f;
        w[j] = (alpha * (pr - delta) + (1.f - alpha) * (pc - delta)) * invB * inv_T;
    }
    __syncthreads();
    const float* other = side == 0 ? bhat : ahat;
    const float* self = side == 0 ? ahat : bhat;
    const float* nrm = side == 0 ? na : nb;
    float* out = side == 0 ? dza : dzb;
    float dot = 0.f;
    float g[8];                                  // D <= 2048; fixed unroll keeps g[] in registers
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int c = t + q * 256;
        g[q] = 0.f;
        if (c < D) {
            float acc = 0.f;
            for (int j = 0; j < B; ++j) acc = fmaf(w[j], other[(long)j * D + c], acc);
            g[q] = acc;
            dot += acc * self[(long)i * D + c];
        }
    }
    float inv = 1.0f;
    if (norm) {
        dot = block_sum(dot, sh);
        inv = 1.0f / fmaxf(nrm[i], eps);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int c = t + q * 256;
        if (c < D) out[(long)i * D + c] = norm ? (g[q] - self[(long)i * D + c] * dot) * inv : g[q];
    }
}

__global__ void l2norm_rows_kernel(const float* __restrict__ x, int rows, int D, float eps, float* __restrict__ z, float* __restrict__ norm) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (long)row * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += p[i] * p[i];
    s = wave_sum(s);
    float nrm = sqrtf(s);
    float inv = 1.0f / fmaxf(nrm, eps);
    for (int i = lane; i < D; i += 64) z[(long)row * D + i] = p[i] * inv;
    if (lane == 0) norm[row] = nrm;
}

static inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }

extern "C" size_t tri_ntxent_workspace(int B, int D) {
    return 2 * al256((size_t)B * D * 4) + 5 * al256((size_t)B * 4) + al256((size_t)B * B * 4);
}

// loss (device scalar) = alpha * CE_rows(S) + (1 - alpha) * CE_cols(S),  S = norm(za) norm(zb)^T / T; dza / dzb (optional)
// receive d loss / d za, d loss / d zb.  zis -> za, zjs -> zb in the reference's argument order (alpha is asymmetric).
extern "C" int tri_ntxent_fwd_bwd(const float* za, const float* zb, int B, int D, float temperature, float alpha, int norm,
                                  float* loss, float* dza, float* dzb, void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 1 || D < 1 || D > 2048 || B > 8192) { tri_set_error("tri_ntxent: need 1<=B<=8192, 1<=D<=2048"); return TRI_ERR_ARG; }
    if (workspace_bytes < tri_ntxent_workspace(B, D)) { tri_set_error("tri_ntxent: workspace too small"); return TRI_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)workspace;
    float* ahat = (float*)w; w += al256((size_t)B * D * 4);
    float* bhat = (float*)w; w += al256((size_t)B * D * 4);
    float* na = (float*)w; w += al256((size_t)B * 4);
    float* nb = (float*)w; w += al256((size_t)B * 4);
    float* lse_row = (float*)w; w += al256((size_t)B * 4);
    float* lse_col = (float*)w; w += al256((size_t)B * 4);
    float* rowloss = (float*)w; w += al256((size_t)B * 4);
    float* S = (float*)w;
    const float eps = 1e-12f;                       // F.normalize default (nt_xent.py:56-57)
    const float* a = za;
    const float* b = zb;
    if (norm) {
        l2norm_rows_kernel<<<(B + 3) / 4, 256, 0, s>>>(za, B, D, eps, ahat, na);
        l2norm_rows_kernel<<<(B + 3) / 4, 256, 0, s>>>(zb, B, D, eps, bhat, nb);
        a = ahat; b = bhat;
    }
    float inv_T = 1.0f / temperature;
    if (B <= 512 && D % 4 == 0 && D <= 8192) ntx_sim_rows_kernel<<<B, 256, (size_t)D * sizeof(float), s>>>(a, b, B, D, inv_T, S);
    else ntx_sim_kernel<<<dim3((B + 63) / 64, (B + 63) / 64), 256, 0, s>>>(a, b, B, D, inv_T, S);
    ntx_lse_kernel<<<B, 256, 0, s>>>(S, B, alpha, lse_row, lse_col, rowloss);
    ntx_loss_sum_kernel<<<1, 64, 0, s>>>(rowloss, B, loss);
    if (dza && dzb)
        ntx_grad_kernel<<<dim3(B, 2), 256, (size_t)B * sizeof(float), s>>>(S, lse_row, lse_col, a, b, na, nb, B, D, inv_T, alpha, norm,
                                                                            eps, dza, dzb, nullptr);
    return tri_check_launch("tri_ntxent_fwd_bwd");
}

// The gradient half on its own, for a `workspace` that a tri_ntxent_fwd_bwd(..., dza = dzb = NULL, ...) call filled: dza / dzb =
// (*dloss) * d loss / d za, zb.  Lets autograd's backward launch ONE kernel per pair with the upstream scalar folded in
// (no elementwise `grad * dloss` passes).
extern "C" int tri_ntxent_bwd(const float* za, const float* zb, int B, int D, float temperature, float alpha, int norm,
                              const float* dloss, float* dza, float* dzb, const void* workspace, size_t workspace_bytes, void* stream) {
    if (B < 1 || D < 1 || D > 2048 || B > 8192) { tri_set_error("tri_ntxent: need 1<=B<=8192, 1<=D<=2048"); return TRI_ERR_ARG; }
    if (workspace_bytes < tri_ntxent_workspace(B, D)) { tri_set_error("tri_ntxent: workspace too small"); return TRI_ERR_ARG; }
    const char* w = (const char*)workspace;
    const float* ahat = (const float*)w; w += al256((size_t)B * D * 4);
    const float* bhat = (const float*)w; w += al256((size_t)B * D * 4);
    const float* na = (const float*)w; w += al256((size_t)B * 4);
    const float* nb = (const float*)w; w += al256((size_t)B * 4);
    const float* lse_row = (const float*)w; w += al256((size_t)B * 4);
    const float* lse_col = (const float*)w; w += al256((size_t)B * 4);
    w += al256((size_t)B * 4);
    const float* S = (const float*)w;
    ntx_grad_kernel<<<dim3(B, 2), 256, (size_t)B * sizeof(float), (hipStream_t)stream>>>(
        S, lse_row, lse_col, norm ? ahat : za, norm ? bhat : zb, na, nb, B, D, 1.0f / temperature, alpha, norm, 1e-12f, dza, dzb, dloss);
    return tri_check_launch("tri_ntxent_bwd");
}
