// Dense layers with at most 64 rows (the towers' MLP heads at the per-GPU batch of the training step), gfx950.
//
// Replaces nn.Linear forward / backward at sparse_cnn.py:39-44, mv_cnn.py:21-26 (net_2, mlp), bigru.py:12 and
// clip_text.py:9-14 when rows <= 64.  Through the general conv path such a layer costs a weight-packing launch, a
// split-K GEMM, its finish kernel (forward and again for the data gradient), an activation-gradient pass, a wgrad
// launch, its slab reduce and a column sum: ~10 launches of 5-11 us each for a few MFLOP.  Here it is three:
//   tri_linear_small_fwd    y  = act(x W^T + b)                      one wave per 16 output columns x K / 4
//   tri_linear_small_dgrad  dx = (dout * act'(y)) W                  one wave per 16 input columns x N / 4
//   tri_linear_small_wgrad  dW = (dout * act'(y))^T x,  db = colsum  one wave per 16 x 16 tile of dW
// Operands are read as fp32 straight from the parameter / activation tensors and split to bf16 in registers (hi, or
// hi + lo with the three-product scheme in bf16x3 mode), so there is no packed copy of W to keep in sync.
// MFMA 16x16x32 bf16: lane l supplies row (l & 15), k = 8 (l >> 4) .. + 7 of both operands and receives
// D[4 (l >> 4) + r][l & 15].
#include "common.h"
#include "../../include/tricolo_hip.h"

__device__ __forceinline__ void split8(const float4& a, const float4& b, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = (bf16_t)v[i];
        lo[i] = (bf16_t)(v[i] - (float)hi[i]);
    }
}
template <int NSPLIT>
__device__ __forceinline__ f32x4 mma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 c) {
    if (NSPLIT == 2) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
__device__ __forceinline__ float act_grad(float d, float o, int act) {
    return act == 1 ? (o > 0.f ? d : 0.f) : (act == 2 ? d * (1.f - o * o) : d);
}

// y[m][n] = act(sum_k x[m][k] W[n][k] + b[n]);  grid = N / 16, 4 waves split K, M <= 64
template <int NSPLIT>
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int M, int K, int N,
                                                               int act) {
    __shared__ f32x4 red[4][4][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int MT = (M + 15) >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int kq = K >> 2;                                           // this wave's share of the contraction
    const float* wrow = w + (size_t)(n0 + fr) * K + wave * kq + fq * 8;
    if (MT <= 2 && kq == 128) {
        // the heads' shape (<= 32 rows, K = 512): all 24 loads of the wave first, then the arithmetic - the rolled loop below waits out
        // one L2 round trip per 32-wide step and per row tile (most of the kernel's 8.5 us)
        float4 wv[4][2], xv[4][2][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wv[q][0] = *(const float4*)(wrow + q * 32);
            wv[q][1] = *(const float4*)(wrow + q * 32 + 4);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = j * 16 + fr;
                const float* xr = x + (size_t)(m < M ? m : 0) * K + wave * kq + fq * 8 + q * 32;
                xv[q][j][0] = *(const float4*)xr;
                xv[q][j][1] = *(const float4*)(xr + 4);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x8 ah, al;
            split8(wv[q][0], wv[q][1], ah, al);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j < MT) {
                    const bool ok = j * 16 + fr < M;
                    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    bf16x8 bh, bl;
                    split8(ok ? xv[q][j][0] : z4, ok ? xv[q][j][1] : z4, bh, bl);
                    acc[j] = mma3<NSPLIT>(ah, al, bh, bl, acc[j]);
                }
            }
        }
    } else
#pragma unroll 4
    for (int k0 = 0; k0 < kq; k0 += 32) {
        bf16x8 ah, al;
        split8(*(const float4*)(wrow + k0), *(const float4*)(wrow + k0 + 4), ah, al);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < MT) {
                const int m = j * 16 + fr;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (m < M) {
                    const float* xr = x + (size_t)m * K + wave * kq + fq * 8 + k0;
                    v0 = *(const float4*)xr;
                    v1 = *(const float4*)(xr + 4);
                }
                bf16x8 bh, bl;
                split8(v0, v1, bh, bl);
                acc[j] = mma3<NSPLIT>(ah, al, bh, bl, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    const int j = wave;                                              // wave j finishes row tile j
    if (j < MT) {
        f32x4 s = red[0][j][lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) s += red[ww][j][lane];
        const int m = j * 16 + fr, n = n0 + fq * 4;
        if (m < M) {
            if (bias) s += *(const f32x4*)(bias + n);
            if (act == 1) { s[0] = fmaxf(s[0], 0.f); s[1] = fmaxf(s[1], 0.f); s[2] = fmaxf(s[2], 0.f); s[3] = fmaxf(s[3], 0.f); }
            else if (act == 2) { s[0] = tanhf(s[0]); s[1] = tanhf(s[1]); s[2] = tanhf(s[2]); s[3] = tanhf(s[3]); }
            *(f32x4*)(y + (size_t)m * N + n) = s;
        }
    }
}

// dx[m][k] = sum_n g[m][n] W[n][k],  g = dout * act'(y);  grid = K / 16, 4 waves split N
template <int NSPLIT>
__device__ __forceinline__ void linear_small_dgrad_body(const float* __restrict__ dout, const float* __restrict__ yout,
                                                        const float* __restrict__ w, float* __restrict__ dx, int M, int K, int N, int act,
                                                        int bx, f32x4 (*red)[4][64]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
    const int k0 = bx * 16;
    const int MT = (M + 15) >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nq = N >> 2;
#pragma unroll 4
    for (int nb = wave * nq; nb < (wave + 1) * nq; nb += 32) {
        // A = W^T: row i = k0 + fr, contraction index n = nb + 8 fq + jj (16 lanes read 16 consecutive k of one W row)
        float av[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) av[jj] = w[(size_t)(nb + fq * 8 + jj) * K + k0 + fr];
        bf16x8 ah, al;
        split8(make_float4(av[0], av[1], av[2], av[3]), make_float4(av[4], av[5], av[6], av[7]), ah, al);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < MT) {
                const int m = j * 16 + fr;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (m < M) {
                    const size_t o = (size_t)m * N + nb + fq * 8;
                    v0 = *(const float4*)(dout + o);
                    v1 = *(const float4*)(dout + o + 4);
                    if (act) {
                        const float4 o0 = *(const float4*)(yout + o), o1 = *(const float4*)(yout + o + 4);
                        v0.x = act_grad(v0.x, o0.x, act); v0.y = act_grad(v0.y, o0.y, act); v0.z = act_grad(v0.z, o0.z, act); v0.w = act_grad(v0.w, o0.w, act);
                        v1.x = act_grad(v1.x, o1.x, act); v1.y = act_grad(v1.y, o1.y, act); v1.z = act_grad(v1.z, o1.z, act); v1.w = act_grad(v1.w, o1.w, act);
                    }
                }
                bf16x8 bh, bl;
                split8(v0, v1, bh, bl);
                acc[j] = mma3<NSPLIT>(ah, al, bh, bl, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    const int j = wave;
    if (j < MT) {
        f32x4 s = red[0][j][lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) s += red[ww][j][lane];
        const int m = j * 16 + fr;
        if (m < M) *(f32x4*)(dx + (size_t)m * K + k0 + fq * 4) = s;
    }
}
template <int NSPLIT>
__global__ __launch_bounds__(256) void linear_small_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ yout,
                                                                 const float* __restrict__ w, float* __restrict__ dx, int M, int K, int N,
                                                                 int act) {
    __shared__ f32x4 red[4][4][64];
    linear_small_dgrad_body<NSPLIT>(dout, yout, w, dx, M, K, N, act, blockIdx.x, red);
}

// dW[n][k] = sum_m g[m][n] x[m][k],  db[n] = sum_m g[m][n];  grid = (N / 16, K / 64): one 16 x 16 tile per wave, so that
// all of a wave's loads are independent and issued at once (four tiles per wave in sequence measured 21 us per layer)
template <int NSPLIT>
__device__ __forceinline__ void linear_small_wgrad_body(const float* __restrict__ x, const float* __restrict__ dout,
                                                        const float* __restrict__ yout, float* __restrict__ dw, float* __restrict__ db,
                                                        int M, int K, int N, int act, int bx, int by) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
    const int n0 = bx * 16, k0 = by * 64 + wave * 16;
    const int MS = (M + 31) >> 5;                                    // 32-row contraction steps (1 or 2)
    bf16x8 ah[2], al[2];
    float gsum = 0.f;                                                // this lane's share of column n0 + fr: rows 8 fq .. 8 fq + 7 of both steps
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float av[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int m = s * 32 + fq * 8 + jj;
            float g = 0.f;
            if (s < MS && m < M) {
                const size_t o = (size_t)m * N + n0 + fr;
                g = dout[o];
                if (act) g = act_grad(g, yout[o], act);
            }
            av[jj] = g;
            gsum += g;
        }
        split8(make_float4(av[0], av[1], av[2], av[3]), make_float4(av[4], av[5], av[6], av[7]), ah[s], al[s]);
    }
    if (db && by == 0 && wave == 0) {
        // bias gradient from the values the fragments were built from: the four k-groups of a column add up through two shuffles (a
        // serial loop over the rows - one dependent load per row on 16 lanes - was the longest thing in the launch: ~20 us at 32 rows)
        gsum += __shfl_xor(gsum, 16);
        gsum += __shfl_xor(gsum, 32);
        if (lane < 16) db[n0 + lane] = gsum;
    }
    if (k0 < K) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s < MS) {
                float bv[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int m = s * 32 + fq * 8 + jj;
                    bv[jj] = m < M ? x[(size_t)m * K + k0 + fr] : 0.f;
                }
                bf16x8 bh, bl;
                split8(make_float4(bv[0], bv[1], bv[2], bv[3]), make_float4(bv[4], bv[5], bv[6], bv[7]), bh, bl);
                acc = mma3<NSPLIT>(ah[s], al[s], bh, bl, acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dw[(size_t)(n0 + fq * 4 + r) * K + k0 + fr] = acc[r];
    }
}
template <int NSPLIT>
__global__ __launch_bounds__(256) void linear_small_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                                 const float* __restrict__ yout, float* __restrict__ dw,
                                                                 float* __restrict__ db, int M, int K, int N, int act) {
    linear_small_wgrad_body<NSPLIT>(x, dout, yout, dw, db, M, K, N, act, blockIdx.x, blockIdx.y);
}
// Both halves of a layer's backward in one launch (they share nothing but their inputs): the first N/16 * ceil(K/64) workgroups
// are the weight-gradient tiles, the remaining K/16 the data-gradient columns.  Same arithmetic as the two kernels above.
template <int NSPLIT>
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                               const float* __restrict__ yout, const float* __restrict__ w,
                                                               float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int M,
                                                               int K, int N, int act) {
    __shared__ f32x4 red[4][4][64];
    const int nbx = N / 16, nw = nbx * ((K + 63) / 64);
    const int b = blockIdx.x;
    if (b < nw) linear_small_wgrad_body<NSPLIT>(x, dout, yout, dw, db, M, K, N, act, b % nbx, b / nbx);
    else linear_small_dgrad_body<NSPLIT>(dout, yout, w, dx, M, K, N, act, b - nw, red);
}

static int linear_small_ok(int M, int K, int N) {
    return M >= 1 && M <= 64 && K % 128 == 0 && N % 128 == 0;
}
extern "C" int tri_linear_small_supported(int M, int K, int N) { return linear_small_ok(M, K, N); }

extern "C" int tri_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, int split3,
                                    void* stream) {
    if (!linear_small_ok(M, K, N)) { tri_set_error("linear_small: needs rows <= 64, K % 128 == 0, N % 128 == 0"); return TRI_ERR_UNSUPPORTED; }
    if (split3) linear_small_fwd_kernel<2><<<N / 16, 256, 0, (hipStream_t)stream>>>(x, w, bias, y, M, K, N, act);
    else linear_small_fwd_kernel<1><<<N / 16, 256, 0, (hipStream_t)stream>>>(x, w, bias, y, M, K, N, act);
    return tri_check_launch("tri_linear_small_fwd");
}

extern "C" int tri_linear_small_dgrad(const float* dout, const float* y, const float* w, float* dx, int M, int K, int N, int act,
                                      int split3, void* stream) {
    if (!linear_small_ok(M, K, N)) { tri_set_error("linear_small: needs rows <= 64, K % 128 == 0, N % 128 == 0"); return TRI_ERR_UNSUPPORTED; }
    if (split3) linear_small_dgrad_kernel<2><<<K / 16, 256, 0, (hipStream_t)stream>>>(dout, y, w, dx, M, K, N, act);
    else linear_small_dgrad_kernel<1><<<K / 16, 256, 0, (hipStream_t)stream>>>(dout, y, w, dx, M, K, N, act);
    return tri_check_launch("tri_linear_small_dgrad");
}

extern "C" int tri_linear_small_wgrad(const float* x, const float* dout, const float* y, float* dw, float* db, int M, int K, int N,
                                      int act, int split3, void* stream) {
    if (!linear_small_ok(M, K, N)) { tri_set_error("linear_small: needs rows <= 64, K % 128 == 0, N % 128 == 0"); return TRI_ERR_UNSUPPORTED; }
    dim3 grid(N / 16, (K + 63) / 64);
    if (split3) linear_small_wgrad_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(x, dout, y, dw, db, M, K, N, act);
    else linear_small_wgrad_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(x, dout, y, dw, db, M, K, N, act);
    return tri_check_launch("tri_linear_small_wgrad");
}

// dW, db and dx of one layer in ONE launch (tri_linear_small_wgrad + tri_linear_small_dgrad: same results)
extern "C" int tri_linear_small_bwd(const float* x, const float* dout, const float* y, const float* w, float* dx, float* dw, float* db, int M,
                                    int K, int N, int act, int split3, void* stream) {
    if (!linear_small_ok(M, K, N)) { tri_set_error("linear_small: needs rows <= 64, K % 128 == 0, N % 128 == 0"); return TRI_ERR_UNSUPPORTED; }
    if (!dx || !dw) { tri_set_error("linear_small_bwd: dx and dw are required (use the single entry points otherwise)"); return TRI_ERR_ARG; }
    const int blocks = (N / 16) * ((K + 63) / 64) + K / 16;
    if (split3) linear_small_bwd_kernel<2><<<blocks, 256, 0, (hipStream_t)stream>>>(x, dout, y, w, dx, dw, db, M, K, N, act);
    else linear_small_bwd_kernel<1><<<blocks, 256, 0, (hipStream_t)stream>>>(x, dout, y, w, dx, dw, db, M, K, N, act);
    return tri_check_launch("tri_linear_small_bwd");
}
