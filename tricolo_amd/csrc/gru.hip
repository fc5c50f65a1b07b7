// Persistent bidirectional GRU recurrence for gfx950 (hidden 128, any batch), forward and backward in time.
//
// Replaces the cuDNN/MIOpen RNN call behind nn.GRU in /root/reference/tricolo/model/module/text_encoder/bigru.py:11,17
// (96 sequential steps x 2 directions; ~3,900 tiny launches per training step through MIOpen on ROCm).
// The input projection x_t W_ih^T + b_ih for all steps is one MFMA GEMM outside (conv_igemm as a 1x1 layer); this
// file is the sequential part:   r = s(xr + hr), z = s(xz + hz), n = tanh(xn + r * hn), h' = (1 - z) n + z h
// with (hr, hz, hn) = h W_hh^T + b_hh, gate order (r, z, n) as torch.nn.GRU.  No packing: pads are stepped through.
//
// One workgroup = 16 batch rows of one direction, 8 waves; wave w owns hidden units [16w, 16w+16) of all three gates,
// so the MFMA C layout (row = 4*(lane>>4)+reg, col = lane&15) gives every lane the r, z, n pre-activations of the SAME
// (row, unit) and the cell update is lane-local.  W_hh never leaves registers (12 B-fragments per wave, +12 for the
// lo half in split mode); h_{t-1} is re-published through LDS as bf16 (hi, lo) with an XOR-swizzled 256-B row image
// (conflict-free ds_read_b128); the fp32 state stays in registers.  One barrier per time step.
#include "common.h"
#include "../../include/tricolo_hip.h"

#define GRU_H 128

// Gate non-linearities on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).  The cell update is the longest part
// of a time step once the operands are prefetched - 8 sigmoids and 4 tanh per lane, two waves per SIMD - and libm's tanhf plus two
// IEEE divisions cost ~3x the 36 MFMAs of the step.  Absolute error ~1e-7 (tanh through 1 - 2 / (1 + e^2x) cancels near 0, where
// its absolute error stays at one ulp of 1): two orders below the parity bounds this path is tested to.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// LDS image of a [16][ncols] bf16 matrix: 16-byte chunk c of row r stored at chunk (c ^ (r & 15))
__device__ __forceinline__ int himg_off(int row, int chunk, int row_bytes) { return row * row_bytes + ((chunk ^ (row & 15)) << 4); }

// E = MFMA operand type: bf16_t (NSPLIT 1: single products, 2: the 3-product hi/lo split) or f16_t (NSPLIT 1, the f16 mode: |h| < 1 and
// |W_hh| < 0.1 sit comfortably in f16's range, 11 significand bits against bf16's 8 keep the final embedding within 4e-5 of float64 over
// the 96 steps - bf16 single products: 2.5e-4 - at a third of the split mode's MFMAs, which are a step's longest serial part)
template <int NSPLIT, typename E>
__global__ __launch_bounds__(512) void gru_fwd_kernel(const float* __restrict__ xproj,   // [L][B][768]
                                                      const float* __restrict__ w_hh,    // [2][384][128]
                                                      const float* __restrict__ b_hh,    // [2][384]
                                                      int B, int L,
                                                      float* __restrict__ hs,            // [2][L][B][128]  h after step t
                                                      float* __restrict__ gates,         // [2][L][B][128][4] r, z, n, hn(+b) per unit
                                                      float* __restrict__ hfinal) {      // [B][256]
    __shared__ __attribute__((aligned(16))) char lds[2 * NSPLIT * 16 * 256];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int dir = blockIdx.y, b0 = blockIdx.x * 16;
    const int fr = lane & 15, fq = lane >> 4;
    const int unit = 16 * w + fr;
    const float* W = w_hh + (size_t)dir * 384 * GRU_H;

    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    // W_hh fragments: B operand of D[row][unit] += h[row][k] * W[g*128 + unit][k]
    v8 wh[3][4], wl[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* src = W + (size_t)(g * GRU_H + unit) * GRU_H + ks * 32 + fq * 8;
            float4 a = *(const float4*)src, c = *(const float4*)(src + 4);
            float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                E h = (E)v[j];
                wh[g][ks][j] = h;
                if (NSPLIT == 2) wl[g][ks][j] = (E)(v[j] - (float)h);
            }
        }
    const float bhr = b_hh[dir * 384 + unit], bhz = b_hh[dir * 384 + 128 + unit], bhn = b_hh[dir * 384 + 256 + unit];

    float h[4] = {0.f, 0.f, 0.f, 0.f};
    // publish h_0 = 0
    for (int i = t; i < 2 * NSPLIT * 16 * 256 / 4; i += 512) ((int*)lds)[i] = 0;
    __syncthreads();

    // Input projections are prefetched THREE steps ahead: a step is ~600 cycles of MFMAs and gate math, a global load ~1-2 us.
    // (Loaded at the top of the step that uses them, every one of the 96 steps waited a full load latency: 2.2 us per step,
    // 214 us for the kernel - on the text tower's forward, which the loss of the whole step waits for.)
    //
    // Round 6: the step is VALU-ISSUE bound, not a latency chain - four CUs run the whole recurrence, two waves per SIMD, and the ISA of the
    // rolled loop was 289 instructions per step of which ~50 were register moves (rotating the prefetch buffers) and ~45 64-bit address
    // arithmetic for 12 loads and 8 stores; at 4 cycles per wave64 VALU instruction (16 for the 24 exp / rcp) that is the 1.1 us a step
    // took.  Here six steps are unrolled: the prefetch buffer of a step and the LDS image it reads / writes are compile-time choices (no
    // moves, no toggling), and every global access is (a scalar base that moves with the step) + (a per-lane 32-bit offset computed once).
    // Same arithmetic in the same order: the results are bit-identical.
    constexpr int PF = 3;
    float xq[PF][12];
    unsigned xoff[4], hoff[4], loff[4];
    bool rowok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = fq * 4 + r, b = b0 + row;
        rowok[r] = b < B;
        xoff[r] = (unsigned)((b < B ? b : 0) * 768 + dir * 384 + unit);
        hoff[r] = (unsigned)((b < B ? b : 0) * GRU_H + unit);
        loff[r] = (unsigned)(himg_off(row, unit >> 3, 256) + (unit & 7) * 2);
    }
    const size_t xstride = (size_t)B * 768, hstride = (size_t)B * GRU_H;
    auto load_x = [&](int s_, float* dst) {
        const int t_ = dir == 0 ? s_ : L - 1 - s_;
        const float* xs = xproj + (size_t)(s_ < L ? t_ : 0) * xstride;      // (uniform; past the end: a harmless re-read of step 0's row)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* xp = xs + xoff[r];
            dst[r] = xp[0]; dst[4 + r] = xp[128]; dst[8 + r] = xp[256];
        }
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) load_x(d, xq[d]);
    float* const hs_d = hs + (size_t)dir * L * hstride;
    float* const gates_d = gates + (size_t)dir * L * hstride * 4;
#pragma unroll 1
    for (int s0 = 0; s0 < L; s0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int s = s0 + u;
            if (s < L) {                                                // (uniform)
                constexpr int NS = NSPLIT * 16 * 256;
                const int tt = dir == 0 ? s : L - 1 - s;
                float* const xb = xq[u % PF];
                float xr[4], xz[4], xn[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { xr[r] = xb[r]; xz[r] = xb[4 + r]; xn[r] = xb[8 + r]; }
                load_x(s + PF, xb);
                const char* hb = lds + (u & 1) * NS;
                f32x4 acc[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    int off = himg_off(fr, ks * 4 + fq, 256);
                    v8 ah = *(const v8*)(hb + off);
                    v8 al;
                    if (NSPLIT == 2) al = *(const v8*)(hb + 16 * 256 + off);
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        if (NSPLIT == 2) {
                            acc[g] = MM::mma(al, wh[g][ks], acc[g]);
                            acc[g] = MM::mma(ah, wl[g][ks], acc[g]);
                        }
                        acc[g] = MM::mma(ah, wh[g][ks], acc[g]);
                    }
                }
                char* hn_buf = lds + ((u & 1) ^ 1) * NS;
                float* const hs_s = hs_d + (size_t)tt * hstride;        // (uniform bases of this step's rows)
                float* const gates_s = gates_d + (size_t)tt * hstride * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float rg = sigmoidf_(xr[r] + acc[0][r] + bhr);
                    float zg = sigmoidf_(xz[r] + acc[1][r] + bhz);
                    float ghn = acc[2][r] + bhn;
                    float ng = tanhf_(xn[r] + rg * ghn);
                    float hnew = (1.f - zg) * ng + zg * h[r];
                    h[r] = hnew;
                    if (rowok[r]) {
                        hs_s[hoff[r]] = hnew;
                        *(float4*)(gates_s + (size_t)hoff[r] * 4) = make_float4(rg, zg, ng, ghn);   // one 16-byte store per (row, unit)
                    }
                    // re-publish as 16-bit (hi, lo): element (row, unit) -> chunk unit/8, byte (unit%8)*2
                    E hh = (E)hnew;
                    *(E*)(hn_buf + loff[r]) = hh;
                    if (NSPLIT == 2) *(E*)(hn_buf + 16 * 256 + loff[r]) = (E)(hnew - (float)hh);
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int b = b0 + fq * 4 + r;
        if (b < B) hfinal[(size_t)b * 256 + dir * GRU_H + unit] = h[r];
    }
}

// Backward through time.  dhfinal [B][256] seeds dh; per step the lane-local cell backward produces the gate
// pre-activation gradients dgi (w.r.t. x W_ih^T + b_ih) and dgh (w.r.t. h W_hh^T + b_hh), stores both for the batched
// weight-gradient GEMMs, and dh_{t-1} = dh * z + dgh @ W_hh through MFMA (W_hh column fragments resident in registers).
// (E = f16_t: the gate gradients go through LDS times 2^12 - they are 1e-2 .. 1e-7, f16's normal range ends at 6e-5 - and the MFMA
//  result is multiplied by 2^-12: both exact)
template <int NSPLIT, typename E>
__global__ __launch_bounds__(512) void gru_bwd_kernel(const float* __restrict__ dhfinal,  // [B][256]
                                                      const float* __restrict__ w_hh,     // [2][384][128]
                                                      const float* __restrict__ hs,       // [2][L][B][128]
                                                      const float* __restrict__ gates,    // [2][L][B][128][4]
                                                      int B, int L,
                                                      float* __restrict__ dgi,            // [L][B][768]
                                                      float* __restrict__ dgh,            // [2][L][B][384]
                                                      float* __restrict__ hprev,          // [2][L][B][128]
                                                      float* __restrict__ dbias) {        // [chunks][2][4][128]: sum_t,rows of (dr, dz, dn_i, dn_h)
    __shared__ __attribute__((aligned(16))) char lds[2 * NSPLIT * 16 * 768];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int dir = blockIdx.y, b0 = blockIdx.x * 16;
    const int fr = lane & 15, fq = lane >> 4;
    const int unit = 16 * w + fr;
    const float* W = w_hh + (size_t)dir * 384 * GRU_H;

    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    constexpr float GS = sizeof(E) == 2 && NSPLIT == 1 && __is_same(E, f16_t) ? 4096.f : 1.f;
    // B operand of D[row][unit] += dgh[row][k] * W[k][unit],  k over the 384 gate units
    v8 wh[12], wl[12];
#pragma unroll
    for (int ks = 0; ks < 12; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = W[(size_t)(ks * 32 + fq * 8 + j) * GRU_H + unit];
            E hgh = (E)v;
            wh[ks][j] = hgh;
            if (NSPLIT == 2) wl[ks][j] = (E)(v - (float)hgh);
        }

    float dh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int b = b0 + fq * 4 + r;
        dh[r] = b < B ? dhfinal[(size_t)b * 256 + dir * GRU_H + unit] : 0.f;
    }
    float sb[4] = {0.f, 0.f, 0.f, 0.f};                      // bias-gradient partial sums of this lane's (rows, unit)
    // saved gates and h_{t-1} of a step are prefetched two steps ahead (see gru_fwd_kernel: every step used to wait for them)
    constexpr int PF = 2;
    float gq[PF][20];                                         // [row][r, z, n, hn] + h_prev[row]
    auto load_g = [&](int s_, float* dst) {
        const int sc = s_ >= 0 ? s_ : 0;
        const int t_ = dir == 0 ? sc : L - 1 - sc;
        const int tp_ = dir == 0 ? t_ - 1 : t_ + 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + fq * 4 + r, bc = b < B ? b : 0;
            const size_t o = ((size_t)dir * L + t_) * B + bc;
            const float4 gv = *(const float4*)(gates + (o * GRU_H + unit) * 4);
            dst[r * 4 + 0] = gv.x; dst[r * 4 + 1] = gv.y; dst[r * 4 + 2] = gv.z; dst[r * 4 + 3] = gv.w;
            dst[16 + r] = sc > 0 ? hs[(((size_t)dir * L + tp_) * B + bc) * GRU_H + unit] : 0.f;
        }
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) load_g(L - 1 - d, gq[d]);
    int cur = 0;
    for (int s = L - 1; s >= 0; --s) {
        const int tt = dir == 0 ? s : L - 1 - s;              // time index processed at forward step s
        char* gb = lds + cur * (NSPLIT * 16 * 768);
        float dhz[4];
        float gcur[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) gcur[k] = gq[0][k];
#pragma unroll
        for (int d = 0; d + 1 < PF; ++d)
#pragma unroll
            for (int k = 0; k < 20; ++k) gq[d][k] = gq[d + 1][k];
        load_g(s - PF, gq[PF - 1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = fq * 4 + r, b = b0 + row;
            float dr_pre = 0.f, dz_pre = 0.f, dn_pre = 0.f, dgn = 0.f, hp = 0.f;
            dhz[r] = 0.f;
            if (b < B) {
                size_t o = ((size_t)dir * L + tt) * B + b;
                float rg = gcur[r * 4 + 0], zg = gcur[r * 4 + 1], ng = gcur[r * 4 + 2], ghn = gcur[r * 4 + 3];
                hp = gcur[16 + r];
                float d = dh[r];
                float dn = d * (1.f - zg);
                float dz = d * (hp - ng);
                dhz[r] = d * zg;
                dn_pre = dn * (1.f - ng * ng);
                dz_pre = dz * zg * (1.f - zg);
                dr_pre = dn_pre * ghn * rg * (1.f - rg);
                dgn = dn_pre * rg;
                float* gi = dgi + ((size_t)tt * B + b) * 768 + dir * 384 + unit;
                gi[0] = dr_pre; gi[128] = dz_pre; gi[256] = dn_pre;
                float* gh = dgh + o * 384 + unit;
                gh[0] = dr_pre; gh[128] = dz_pre; gh[256] = dgn;
                hprev[o * GRU_H + unit] = hp;
            }
            float vals[3] = {dr_pre * GS, dz_pre * GS, dgn * GS};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                int col = g * GRU_H + unit;
                E hh = (E)vals[g];
                int off = himg_off(row, col >> 3, 768) + (col & 7) * 2;
                *(E*)(gb + off) = hh;
                if (NSPLIT == 2) *(E*)(gb + 16 * 768 + off) = (E)(vals[g] - (float)hh);
            }
            // (round 6) the bias sums are taken AFTER the scaled copies: the stored gradients stay live past `vals`, so the compiler cannot
            // form dgn * GS etc. in place - it had put `v_pk_mul_f32 v[74:75], v[74:75], GS` two instructions behind
            // `global_store_dword ..., v74` / `v75`, and about one replay of the training step in 300 stored the scaled value in the lanes
            // 48..63 of one wave (tools/r6/determinism.py: one row, 16 units, |diff| = 4095 x the value)
            __builtin_amdgcn_sched_barrier(0);
            if (b < B) { sb[0] += dr_pre; sb[1] += dz_pre; sb[2] += dn_pre; sb[3] += dgn; }
        }
        __syncthreads();
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            int off = himg_off(fr, ks * 4 + fq, 768);
            v8 ah = *(const v8*)(gb + off);
            if (NSPLIT == 2) {
                v8 al = *(const v8*)(gb + 16 * 768 + off);
                acc = MM::mma(al, wh[ks], acc);
                acc = MM::mma(ah, wl[ks], acc);
            }
            acc = MM::mma(ah, wh[ks], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[r] = dhz[r] + acc[r] * (1.f / GS);
        cur ^= 1;                                             // next step writes the other image: one barrier per step
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = sb[k];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (fq == 0) dbias[(((size_t)blockIdx.x * 2 + dir) * 4 + k) * GRU_H + unit] = v;
    }
}

extern "C" int tri_gru_fwd(const float* xproj, const float* w_hh, const float* b_hh, int B, int L, float* hs, float* gates,
                           float* hfinal, int split3, void* stream) {
    if (B < 1 || L < 1) { tri_set_error("tri_gru_fwd: B, L must be positive"); return TRI_ERR_ARG; }
    dim3 grid((B + 15) / 16, 2);
    // split3: 0 single bf16 products, 1 the 3-product bf16 split, 2 single f16 products (the f16 mode)
    if (split3 == 2) gru_fwd_kernel<1, f16_t><<<grid, 512, 0, (hipStream_t)stream>>>(xproj, w_hh, b_hh, B, L, hs, gates, hfinal);
    else if (split3) gru_fwd_kernel<2, bf16_t><<<grid, 512, 0, (hipStream_t)stream>>>(xproj, w_hh, b_hh, B, L, hs, gates, hfinal);
    else gru_fwd_kernel<1, bf16_t><<<grid, 512, 0, (hipStream_t)stream>>>(xproj, w_hh, b_hh, B, L, hs, gates, hfinal);
    return tri_check_launch("tri_gru_fwd");
}

extern "C" int tri_gru_bwd(const float* dhfinal, const float* w_hh, const float* hs, const float* gates, int B, int L, float* dgi,
                           float* dgh, float* hprev, float* dbias, int split3, void* stream) {
    if (B < 1 || L < 1) { tri_set_error("tri_gru_bwd: B, L must be positive"); return TRI_ERR_ARG; }
    dim3 grid((B + 15) / 16, 2);
    if (split3 == 2) gru_bwd_kernel<1, f16_t><<<grid, 512, 0, (hipStream_t)stream>>>(dhfinal, w_hh, hs, gates, B, L, dgi, dgh, hprev, dbias);
    else if (split3) gru_bwd_kernel<2, bf16_t><<<grid, 512, 0, (hipStream_t)stream>>>(dhfinal, w_hh, hs, gates, B, L, dgi, dgh, hprev, dbias);
    else gru_bwd_kernel<1, bf16_t><<<grid, 512, 0, (hipStream_t)stream>>>(dhfinal, w_hh, hs, gates, B, L, dgi, dgh, hprev, dbias);
    return tri_check_launch("tri_gru_bwd");
}
