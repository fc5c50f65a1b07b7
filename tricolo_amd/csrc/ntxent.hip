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

// ================================================================================================ all pairs of a step at once
// TriCoLoNet._calculate_losses (tricolo_net.py:56-63) evaluates the loss for every pair of modalities and sums them: per pair
// five launches here (fifteen in the reference), plus the additions, plus autograd's accumulation of the two gradients every
// embedding receives - ~25 launches of ~5 us each, in series, with all three towers waiting.  The multi form does the same
// arithmetic for M = 2 or 3 modalities in FOUR launches forward and ONE backward: every modality is normalised once, the
// pair index is a grid dimension, the total is formed by the loss-sum kernel in the reference's order ((l0 + l1) + l2 in
// fp32), and the gradient kernel sums an embedding's contributions from its pairs before the normalise Jacobian (linear).
#define NTX_MAX_MOD 3
struct NtxMulti {
    const float* z[NTX_MAX_MOD];
    float* zhat[NTX_MAX_MOD];
    float* nrm[NTX_MAX_MOD];
    float* S[NTX_MAX_MOD];               // per pair
    float* lse_row[NTX_MAX_MOD];
    float* lse_col[NTX_MAX_MOD];
    float* rowloss[NTX_MAX_MOD];
    const float* dpair[NTX_MAX_MOD];     // backward: upstream gradient of each pair's loss (device scalars, may be NULL)
    const float* dtotal;                 //           ... and of the total
    float* dz[NTX_MAX_MOD];
    int pa[NTX_MAX_MOD], pb[NTX_MAX_MOD];
    int M, P, B, D, norm;
    float inv_T, alpha, eps;
};

__global__ void ntxm_l2norm_kernel(const NtxMulti p) {
    const int m = blockIdx.y;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= p.B) return;
    const float* x = p.z[m] + (long)row * p.D;
    float s = 0.f;
#pragma unroll 8                                                   // (rolled loops wait out one load round trip per step: misc.hip l2norm kernels)
    for (int i = lane; i < p.D; i += 64) s += x[i] * x[i];
    s = wave_sum(s);
    float nrm = sqrtf(s);
    float inv = 1.0f / fmaxf(nrm, p.eps);
#pragma unroll 8
    for (int i = lane; i < p.D; i += 64) p.zhat[m][(long)row * p.D + i] = x[i] * inv;
    if (lane == 0) p.nrm[m][row] = nrm;
}

// block (row i, pair q): the arithmetic of ntx_sim_rows_kernel
__global__ __launch_bounds__(256) void ntxm_sim_rows_kernel(const NtxMulti p) {
    extern __shared__ float sa[];
    const int i = blockIdx.x, q = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int B = p.B, D = p.D;
    const float* a = p.norm ? p.zhat[p.pa[q]] : p.z[p.pa[q]];
    const float* b = p.norm ? p.zhat[p.pb[q]] : p.z[p.pb[q]];
    float* S = p.S[q];
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
            if (lane == 0 && jb + u < B) S[(long)i * B + jb + u] = v * p.inv_T;
        }
    }
}

__global__ __launch_bounds__(256) void ntxm_lse_kernel(const NtxMulti p) {
    __shared__ float sh[4];
    const int i = blockIdx.x, q = blockIdx.y, t = threadIdx.x, B = p.B;
    const float* S = p.S[q];
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
        p.lse_row[q][i] = lr;
        p.lse_col[q][i] = lc;
        p.rowloss[q][i] = -(p.alpha * (d - lr) + (1.f - p.alpha) * (d - lc)) / (float)B;
    }
}

// one wave per pair sums its row losses (double, fixed order per lane + shuffle tree: reproducible); thread 0 forms the total
__global__ void ntxm_loss_sum_kernel(const NtxMulti p, float* __restrict__ losses) {
    __shared__ float lp[NTX_MAX_MOD];
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (q < p.P) {
        double s = 0.0;
        for (int i = lane; i < p.B; i += 64) s += (double)p.rowloss[q][i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) { lp[q] = (float)s; losses[q] = (float)s; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;                                           // Python's sum(): ((0 + l0) + l1) + l2 in fp32
        for (int q2 = 0; q2 < p.P; ++q2) tot += lp[q2];
        losses[p.P] = tot;
    }
}

// block (row i, modality m): d z_m[i] summed over the pairs m takes part in
__global__ __launch_bounds__(256) void ntxm_grad_kernel(const NtxMulti p) {
    extern __shared__ float w[];                 // [2][B] coefficients of the (at most two) pairs of this modality
    __shared__ float sh[4];
    const int i = blockIdx.x, m = blockIdx.y, t = threadIdx.x, B = p.B, D = p.D;
    const float* other[2] = {nullptr, nullptr};
    int np = 0;
    const float gt = p.dtotal ? *p.dtotal : 0.f;
    for (int q = 0; q < p.P; ++q) {
        const int side = p.pa[q] == m ? 0 : (p.pb[q] == m ? 1 : -1);
        if (side < 0) continue;
        const float up = gt + (p.dpair[q] ? *p.dpair[q] : 0.f);
        const float invB = up / (float)B;
        const float* S = p.S[q];
        const float* lr = p.lse_row[q];
        const float* lc = p.lse_col[q];
        for (int j = t; j < B; j += 256) {
            float s, pr, pc;
            if (side == 0) { s = S[(long)i * B + j]; pr = expf(s - lr[i]); pc = expf(s - lc[j]); }
            else           { s = S[(long)j * B + i]; pr = expf(s - lr[j]); pc = expf(s - lc[i]); }
            float delta = (i == j) ? 1.f : 0.f;
            w[np * B + j] = (p.alpha * (pr - delta) + (1.f - p.alpha) * (pc - delta)) * invB * p.inv_T;
        }
        const int o = side == 0 ? p.pb[q] : p.pa[q];
        other[np] = p.norm ? p.zhat[o] : p.z[o];
        ++np;
    }
    __syncthreads();
    const float* self = p.norm ? p.zhat[m] : p.z[m];
    float dot = 0.f;
    float g[8];                                  // D <= 2048
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int c = t + q * 256;
        g[q] = 0.f;
        if (c < D) {
            float tot = 0.f;
            for (int k = 0; k < np; ++k) {       // pair by pair, each in the single-pair kernel's order, then summed
                float acc = 0.f;
                const float* ot = other[k];
                const float* wk = w + k * B;
#pragma unroll 8                                                   // (same summation order; eight loads in flight instead of one)
                for (int j = 0; j < B; ++j) acc = fmaf(wk[j], ot[(long)j * D + c], acc);
                tot += acc;
            }
            g[q] = tot;
            dot += tot * self[(long)i * D + c];
        }
    }
    float inv = 1.0f;
    if (p.norm) {
        dot = block_sum(dot, sh);
        inv = 1.0f / fmaxf(p.nrm[m][i], p.eps);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int c = t + q * 256;
        if (c < D) p.dz[m][(long)i * D + c] = p.norm ? (g[q] - self[(long)i * D + c] * dot) * inv : g[q];
    }
}

extern "C" size_t tri_ntxent_multi_workspace(int M, int B, int D) {
    const int P = M * (M - 1) / 2;
    return (size_t)M * (al256((size_t)B * D * 4) + al256((size_t)B * 4)) + (size_t)P * (3 * al256((size_t)B * 4) + al256((size_t)B * B * 4));
}

static int ntxm_setup(NtxMulti& p, const float* const* z, int M, int B, int D, float temperature, float alpha, int norm, void* workspace,
                      size_t workspace_bytes) {
    if (M < 2 || M > NTX_MAX_MOD) { tri_set_error("tri_ntxent_multi: 2 or 3 modalities"); return TRI_ERR_ARG; }
    if (B < 1 || B > 512 || D < 4 || D > 2048 || D % 4) { tri_set_error("tri_ntxent_multi: need 1<=B<=512, 4<=D<=2048, D%4==0 (else use tri_ntxent_fwd_bwd per pair)"); return TRI_ERR_UNSUPPORTED; }
    if (!z || !workspace || workspace_bytes < tri_ntxent_multi_workspace(M, B, D)) { tri_set_error("tri_ntxent_multi: workspace too small"); return TRI_ERR_ARG; }
    p.M = M; p.P = M * (M - 1) / 2; p.B = B; p.D = D; p.norm = norm; p.inv_T = 1.0f / temperature; p.alpha = alpha; p.eps = 1e-12f;
    char* w = (char*)workspace;
    for (int m = 0; m < M; ++m) {
        if (!z[m]) { tri_set_error("tri_ntxent_multi: NULL embedding"); return TRI_ERR_ARG; }
        p.z[m] = z[m];
        p.zhat[m] = (float*)w; w += al256((size_t)B * D * 4);
        p.nrm[m] = (float*)w; w += al256((size_t)B * 4);
    }
    int q = 0;
    for (int a = 0; a < M; ++a)                    // itertools.combinations order: the earlier modality is the alpha side
        for (int b = a + 1; b < M; ++b, ++q) {
            p.pa[q] = a; p.pb[q] = b;
            p.lse_row[q] = (float*)w; w += al256((size_t)B * 4);
            p.lse_col[q] = (float*)w; w += al256((size_t)B * 4);
            p.rowloss[q] = (float*)w; w += al256((size_t)B * 4);
            p.S[q] = (float*)w; w += al256((size_t)B * B * 4);
        }
    return 0;
}

// losses [P + 1] (device): the pair losses in combination order ((0,1), (0,2), (1,2)), then their sum
extern "C" int tri_ntxent_multi_fwd(const float* const* z, int M, int B, int D, float temperature, float alpha, int norm, float* losses,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    NtxMulti p{};
    int rc = ntxm_setup(p, z, M, B, D, temperature, alpha, norm, workspace, workspace_bytes);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (norm) ntxm_l2norm_kernel<<<dim3((B + 3) / 4, M), 256, 0, s>>>(p);
    ntxm_sim_rows_kernel<<<dim3(B, p.P), 256, (size_t)D * sizeof(float), s>>>(p);
    ntxm_lse_kernel<<<dim3(B, p.P), 256, 0, s>>>(p);
    ntxm_loss_sum_kernel<<<1, 64 * NTX_MAX_MOD, 0, s>>>(p, losses);
    return tri_check_launch("tri_ntxent_multi_fwd");
}

// dz[m] = sum over the pairs of m of (dtotal + dpair[pair]) * d loss_pair / d z_m, from the workspace the forward call filled
extern "C" int tri_ntxent_multi_bwd(const float* const* z, int M, int B, int D, float temperature, float alpha, int norm,
                                    const float* const* dpair, const float* dtotal, float* const* dz, const void* workspace,
                                    size_t workspace_bytes, void* stream) {
    NtxMulti p{};
    int rc = ntxm_setup(p, z, M, B, D, temperature, alpha, norm, (void*)workspace, workspace_bytes);
    if (rc) return rc;
    if (!dz) { tri_set_error("tri_ntxent_multi_bwd: dz is NULL"); return TRI_ERR_ARG; }
    for (int m = 0; m < M; ++m) {
        if (!dz[m]) { tri_set_error("tri_ntxent_multi_bwd: NULL gradient buffer"); return TRI_ERR_ARG; }
        p.dz[m] = dz[m];
    }
    for (int q = 0; q < p.P; ++q) p.dpair[q] = dpair ? dpair[q] : nullptr;
    p.dtotal = dtotal;
    ntxm_grad_kernel<<<dim3(B, M), 256, (size_t)2 * B * sizeof(float), (hipStream_t)stream>>>(p);
    return tri_check_launch("tri_ntxent_multi_bwd");
}
