// Weight-gradient of the implicit-GEMM convolution on MFMA (gfx950).
//
//   dW[co][tap, ci] = sum_m dOut[m, co] * In[gather(m, tap), ci]          (m = output position)
//
// Replaces the wgrad kernels autograd reaches through spconv / cuDNN for the layers named in conv_igemm.hip.
// Both operands are contracted over POSITIONS, which are the slow axis of the channels-last tensors, so each
// staging thread loads a 4(position) x 4(channel) fp32 block, transposes it in registers and writes
// [channel][32 positions] bf16 rows: the same LDS operand image and ds_read_b128 fragments as the forward kernel.
// Split over positions (gridDim.y) into fp32 slabs that tri_wgrad_reduce sums in a fixed order (bitwise
// reproducible, no float atomics) straight into the reference's parameter layout.
// Submanifold layers pass the output-site mask: 32-position steps with no active site are skipped.
#include "common.h"
#include "../../include/tricolo_hip.h"

struct WgradArgs {
    const float* in;
    const float* dout;
    const uint8_t* row_mask;
    float* slab;                 // [splits][Cout][Kpad]
    int B, ID, IH, IW, Cin;
    int OD, OH, OW, Cout;
    int KD, KH, KW, stride, pd, ph, pw;
    int Kpad, M, ntaps, cin_shift, steps_per_split;
    FastDiv dOW, dOH, dOD, dCin;
};

template <int BI, int BJ, int NSPLIT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int WI = BI / 2, WJ = BJ / 2, TM = WI / 16, TN = WJ / 16;
    constexpr int X_BYTES = BI * 64, Y_BYTES = BJ * 64;
    constexpr int STAGE = NSPLIT * (X_BYTES + Y_BYTES);
    constexpr int XB = 8 * (BI / 4), YB = 8 * (BJ / 4);          // 4x4 register-transpose units per k-step
    constexpr int UNITS = (XB + YB + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lut = (int*)(smem + 2 * STAGE);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int JT = (p.Kpad + BJ - 1) / BJ;
    const int tile = blockIdx.x;
    const int it = tile / JT, jt = tile - it * JT;
    const int i0 = it * BI, j0 = jt * BJ;
    const int split = blockIdx.y;
    const int ks_begin = split * p.steps_per_split;
    const int nsteps_total = (p.M + 31) >> 5;
    const int ks_end = min(nsteps_total, ks_begin + p.steps_per_split);

    if (t < 64) {
        int kd = 0, kh = 0, kw = 0;
        if (t < p.ntaps) {
            kw = t % p.KW;
            int r = t / p.KW;
            kh = r % p.KH;
            kd = r / p.KH;
        }
        lut[t] = kd | (kh << 8) | (kw << 16);
    }
    __syncthreads();

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wi = wave >> 1, wj = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    float4 v[UNITS][4];

    // a 32-position step is live when any of its output sites is active (dense layers: always)
    auto step_live = [&](int ks) -> bool {
        if (!p.row_mask) return true;
        int m = ks * 32;
        const uint32_t* mp = (const uint32_t*)(p.row_mask + m);          // M is padded to 32 by the caller's mask buffer
        uint32_t any = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) any |= (m + q * 4 < p.M) ? mp[q] : 0u;
        return any != 0;
    };
    auto next_live = [&](int ks) -> int {
        while (ks < ks_end && !step_live(ks)) ++ks;
        return ks;
    };

    auto load_global = [&](int ks) {
        const int mbase = ks * 32;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            int unit = t + u * 256;
            if (unit < XB) {
                int cg = unit % (BI / 4), mg = unit / (BI / 4);
                int co = i0 + cg * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int m = mbase + mg * 4 + i;
                    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m < p.M && co < p.Cout) x = *(const float4*)(p.dout + (size_t)m * p.Cout + co);
                    v[u][i] = x;
                }
            } else if (unit < XB + YB) {
                int uy = unit - XB;
                int jg = uy % (BJ / 4), mg = uy / (BJ / 4);
                int j = j0 + jg * 4;
                int tap, c;
                if (p.cin_shift >= 0) { tap = j >> p.cin_shift; c = j & ((1 << p.cin_shift) - 1); }
                else { tap = (int)fdiv((uint32_t)j, p.dCin); c = j - tap * p.Cin; }
                bool tv = tap < p.ntaps;
                int code = lut[tv ? tap : 0];
                int kd = code & 255, kh = (code >> 8) & 255, kw = (code >> 16) & 255;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int m = mbase + mg * 4 + i;
                    bool ok = tv && m < p.M;
                    uint32_t mm = ok ? (uint32_t)m : 0u;
                    uint32_t q1 = fdiv(mm, p.dOW);
                    int ow = mm - q1 * p.OW;
                    uint32_t q2 = fdiv(q1, p.dOH);
                    int oh = q1 - q2 * p.OH;
                    uint32_t b = fdiv(q2, p.dOD);
                    int od = q2 - b * p.OD;
                    int iz = od * p.stride - p.pd + kd, iy = oh * p.stride - p.ph + kh, ix = ow * p.stride - p.pw + kw;
                    ok = ok && (unsigned)iz < (unsigned)p.ID && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
                    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) {
                        size_t off = ((size_t)(((int)b * p.ID + iz) * p.IH + iy) * p.IW + ix) * p.Cin + c;
                        x = *(const float4*)(p.in + off);
                    }
                    v[u][i] = x;
                }
            }
        }
    };
    auto store_lds = [&](int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            int unit = t + u * 256;
            if (unit >= XB + YB) continue;
            bool isx = unit < XB;
            int uu = isx ? unit : unit - XB;
            int per = isx ? (BI / 4) : (BJ / 4);
            int cg = uu % per, mg = uu / per;
            char* tb = isx ? base : base + NSPLIT * X_BYTES;
            int lo_off = isx ? X_BYTES : Y_BYTES;
            // register transpose: channel c of the 4 consecutive positions -> one 8-byte LDS write
            float4 tr[4];
            tr[0] = make_float4(v[u][0].x, v[u][1].x, v[u][2].x, v[u][3].x);
            tr[1] = make_float4(v[u][0].y, v[u][1].y, v[u][2].y, v[u][3].y);
            tr[2] = make_float4(v[u][0].z, v[u][1].z, v[u][2].z, v[u][3].z);
            tr[3] = make_float4(v[u][0].w, v[u][1].w, v[u][2].w, v[u][3].w);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int off = tile_off(cg * 4 + c, mg >> 1) + (mg & 1) * 8;
                if (NSPLIT == 2) {
                    bf16x4 h, l;
                    split_bf16(tr[c], h, l);
                    *(bf16x4*)(tb + off) = h;
                    *(bf16x4*)(tb + lo_off + off) = l;
                } else {
                    *(bf16x4*)(tb + off) = to_bf16x4(tr[c]);
                }
            }
        }
    };
    auto compute = [&](int buf) {
        const char* xb = smem + buf * STAGE;
        const char* yb = xb + NSPLIT * X_BYTES;
        bf16x8 ah[TM], al[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            int off = tile_off(wi * WI + a * 16 + fr, fq);
            ah[a] = *(const bf16x8*)(xb + off);
            if (NSPLIT == 2) al[a] = *(const bf16x8*)(xb + X_BYTES + off);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            int off = tile_off(wj * WJ + b * 16 + fr, fq);
            bf16x8 bhf = *(const bf16x8*)(yb + off);
            bf16x8 blf;
            if (NSPLIT == 2) blf = *(const bf16x8*)(yb + Y_BYTES + off);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                if (NSPLIT == 2) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[a], bhf, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], blf, acc[a][b], 0, 0, 0);
                }
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bhf, acc[a][b], 0, 0, 0);
            }
        }
    };

    int ks = next_live(ks_begin);
    int buf = 0;
    if (ks < ks_end) {
        load_global(ks);
        store_lds(0);
        __syncthreads();
        while (ks < ks_end) {
            int nxt = next_live(ks + 1);
            if (nxt < ks_end) load_global(nxt);
            compute(buf);
            if (nxt < ks_end) store_lds(buf ^ 1);
            __syncthreads();
            buf ^= 1;
            ks = nxt;
        }
    }

    float* slab = p.slab + (size_t)split * p.Cout * p.Kpad;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int co = i0 + wi * WI + a * 16 + fq * 4 + r;
            if (co >= p.Cout) continue;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                int j = j0 + wj * WJ + b * 16 + fr;
                if (j < p.Kpad) slab[(size_t)co * p.Kpad + j] = acc[a][b][r];
            }
        }
}

// dw[co*s_co + tap*s_tap + ci*s_ci] = sum_split slab[split][co][tap*cin_stored + ci]   (ci < cin_real)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, int Cout, int Kpad, int ntaps, int cin_stored,
                                    int cin_real, float* __restrict__ dw, long s_co, long s_tap, long s_ci) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)Cout * ntaps * cin_stored;
    if (idx >= total) return;
    int co = (int)(idx / (ntaps * cin_stored));
    int k = (int)(idx - (long)co * ntaps * cin_stored);
    int tap = k / cin_stored, ci = k - tap * cin_stored;
    if (ci >= cin_real) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slab[((size_t)z * Cout + co) * Kpad + k];
    dw[co * s_co + tap * s_tap + ci * s_ci] = s;
}


static int ilog2_exact(int v) {
    for (int s = 0; s < 31; ++s) if ((1 << s) == v) return s;
    return -1;
}

static void wgrad_plan(const TriConvDesc* d, int* BI, int* tiles, int* splits, int* steps_per_split, int* Kpad) {
    int ntaps = d->KD * d->KH * d->KW;
    *Kpad = (ntaps * d->Cin + 31) / 32 * 32;
    *BI = (d->Cout % 128 == 0 && *Kpad >= 128) ? 128 : 64;
    int BJ = *BI;
    int it = (d->Cout + *BI - 1) / *BI, jt = (*Kpad + BJ - 1) / BJ;
    *tiles = it * jt;
    long M = (long)d->B * d->OD * d->OH * d->OW;
    int steps = (int)((M + 31) / 32);
    int want = (768 + *tiles - 1) / *tiles;                     // aim at ~768 workgroups (3 per CU): slab traffic grows with splits
    int max_by_steps = steps / 4 > 0 ? steps / 4 : 1;           // at least 4 k-steps per split
    int s = want < max_by_steps ? want : max_by_steps;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    *steps_per_split = (steps + s - 1) / s;
    *splits = (steps + *steps_per_split - 1) / *steps_per_split;
}

extern "C" size_t tri_conv_wgrad_workspace(const TriConvDesc* d) {
    int BI, tiles, splits, sps, Kpad;
    wgrad_plan(d, &BI, &tiles, &splits, &sps, &Kpad);
    return (size_t)splits * d->Cout * Kpad * sizeof(float);
}

template <int BI, int NSPLIT>
static int launch_wgrad(const WgradArgs& a, int tiles, int splits, hipStream_t stream) {
    constexpr int STAGE = NSPLIT * (BI * 64 + BI * 64);
    size_t smem = 2 * STAGE + 256;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<BI, BI, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    conv_wgrad_kernel<BI, BI, NSPLIT><<<dim3(tiles, splits), 256, smem, stream>>>(a);
    return tri_check_launch("tri_conv_wgrad");
}

// dw (addressed by element strides s_co / s_tap / s_ci, i.e. directly in the reference's parameter layout)
//   = sum over positions of dout x im2col(in).  row_mask (optional, per output position, buffer padded to a
// multiple of 32 bytes) marks live positions; split3 != 0 selects the 3-product bf16 split mode.
extern "C" int tri_conv_wgrad(const TriConvDesc* d, const float* in, const float* dout, const uint8_t* row_mask, void* workspace,
                              size_t workspace_bytes, float* dw, long s_co, long s_tap, long s_ci, int cin_real, int split3,
                              void* stream) {
    if (d->Cin % 4 != 0 || d->Cout % 4 != 0) { tri_set_error("wgrad: channels must be multiples of 4"); return TRI_ERR_ARG; }
    int BI, tiles, splits, sps, Kpad;
    wgrad_plan(d, &BI, &tiles, &splits, &sps, &Kpad);
    if (workspace_bytes < (size_t)splits * d->Cout * Kpad * sizeof(float)) { tri_set_error("wgrad: workspace too small"); return TRI_ERR_ARG; }
    WgradArgs a{};
    a.in = in; a.dout = dout; a.row_mask = row_mask; a.slab = (float*)workspace;
    a.B = d->B; a.ID = d->ID; a.IH = d->IH; a.IW = d->IW; a.Cin = d->Cin;
    a.OD = d->OD; a.OH = d->OH; a.OW = d->OW; a.Cout = d->Cout;
    a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pd = d->pad_d; a.ph = d->pad_h; a.pw = d->pad_w;
    a.ntaps = d->KD * d->KH * d->KW;
    if (a.ntaps > 64) { tri_set_error("wgrad: more than 64 taps unsupported"); return TRI_ERR_UNSUPPORTED; }
    a.Kpad = Kpad;
    a.M = d->B * d->OD * d->OH * d->OW;
    a.cin_shift = ilog2_exact(a.Cin);
    a.steps_per_split = sps;
    a.dOW = make_fastdiv(a.OW); a.dOH = make_fastdiv(a.OH); a.dOD = make_fastdiv(a.OD); a.dCin = make_fastdiv(a.Cin);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (BI == 128) rc = split3 ? launch_wgrad<128, 2>(a, tiles, splits, s) : launch_wgrad<128, 1>(a, tiles, splits, s);
    else rc = split3 ? launch_wgrad<64, 2>(a, tiles, splits, s) : launch_wgrad<64, 1>(a, tiles, splits, s);
    if (rc) return rc;
    long total = (long)d->Cout * a.ntaps * d->Cin;
    wgrad_reduce_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>((const float*)workspace, splits, d->Cout, Kpad, a.ntaps, d->Cin,
                                                                    cin_real, dw, s_co, s_tap, s_ci);
    return tri_check_launch("tri_wgrad_reduce");
}
