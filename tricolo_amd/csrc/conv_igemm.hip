// Implicit-GEMM convolution on MFMA for gfx950: forward and data-gradient of every conv / linear layer of the
// TriCoLo towers (3D submanifold 3x3x3, 2D 7x7/2, 3x3/1, 3x3/2, 1x1/2, and dense layers as 1x1x1).
//
// Replaces the third-party kernels behind /root/reference/tricolo/model/module/voxel_encoder/sparse_cnn.py:12-32
// (spconv.SubMConv3d), img_encoder/mv_cnn.py:29 (torchvision ResNet-18 convs via cuDNN) and the nn.Linear calls
// (sparse_cnn.py:39-44, mv_cnn.py:21-26, bigru.py:12, clip_text.py:9-14).
//
// Layout: activations channels-last fp32 [B, D, H, W, C] in HBM, so one im2col row segment (one tap, 32
// channels) is 128 contiguous bytes.  GEMM view:  out[m, n] = sum_k A[m, k] * W[n, k],  m = output position,
// k = tap * Cin + ci, W pre-packed [Cout][Kpad] bf16 (hi and, for the 3-product split mode, lo).
// Tile 128(M) x BN x 32(K), 256 threads = 4 waves, v_mfma_f32_16x16x32_bf16, fp32 accumulate.
// Precision modes: NSPLIT=1  bf16 operands;  NSPLIT=2  x = hi + lo split, acc += a_lo*b_hi + a_hi*b_lo + a_hi*b_hi
// (three bf16 MFMAs, ~2^-17 relative operand error: fp32-grade parity at 3/16 of the f32-MFMA cost).
#include "common.h"
#include "../../include/tricolo_hip.h"

struct ConvArgs {
    const float* in;
    const bf16_t* w_hi;
    const bf16_t* w_lo;
    float* out;
    const uint8_t* row_mask;   // per output position; 0 -> row forced to zero, all-zero tiles are skipped
    const float* bias;
    float* stats;              // [num_mtiles][2][Cout] per-tile column sum / sum of squares (BatchNorm statistics)
    int B, ID, IH, IW, Cin;
    int OD, OH, OW, Cout;
    int KD, KH, KW, stride, pd, ph, pw;
    int transposed, act, accumulate;
    int Kpad, M, ntaps, cin_shift;
    FastDiv dOW, dOH, dOD, dCin;
};

template <int BN, int NSPLIT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int BM = 128;
    constexpr int WAVES_N = (BN >= 64) ? 2 : 1, WAVES_M = 4 / WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
    constexpr int STAGE = NSPLIT * (A_BYTES + B_BYTES);
    constexpr int BCH = (BN * 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lut = (int*)(smem + 2 * STAGE);                 // [64] packed (kd | kh<<8 | kw<<16)
    float* red = (float*)(smem + 2 * STAGE + 256);       // [WAVES_M][BN][2]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int NT = p.Cout / BN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = wg / NT, ntile = wg - mtile * NT;
    const int m0 = mtile * BM, n0 = ntile * BN;

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

    // ---- per-thread im2col rows: 4 rows (t>>3) + 32 i, one float4 (4 consecutive k) per row and k-step
    const int k4 = t & 7;
    int rb[4], rz[4], ry[4], rx[4];
    bool rv[4];
    int any_active = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + (t >> 3) + 32 * i;
        bool valid = m < p.M;
        uint32_t mm = valid ? (uint32_t)m : 0u;
        uint32_t q1 = fdiv(mm, p.dOW);
        int ow = mm - q1 * p.OW;
        uint32_t q2 = fdiv(q1, p.dOH);
        int oh = q1 - q2 * p.OH;
        uint32_t b = fdiv(q2, p.dOD);
        int od = q2 - b * p.OD;
        if (p.row_mask) valid = valid && (p.row_mask[mm] != 0);
        rv[i] = valid;
        any_active |= valid ? 1 : 0;
        rb[i] = (int)b * p.ID * p.IH * p.IW;
        if (p.transposed) {
            rz[i] = od + p.pd; ry[i] = oh + p.ph; rx[i] = ow + p.pw;
        } else {
            rz[i] = od * p.stride - p.pd; ry[i] = oh * p.stride - p.ph; rx[i] = ow * p.stride - p.pw;
        }
    }
    any_active = __syncthreads_or(any_active);       // also publishes lut[]

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;

    if (any_active) {
        const int nk = p.Kpad >> 5;
        const int sshift = (p.stride == 2) ? 1 : 0;
        float4 av[4];
        uint4 bh0, bh1, bl0, bl1;       // named, not an array: hipcc keeps conditionally-written arrays in scratch
        bh0 = bh1 = bl0 = bl1 = make_uint4(0, 0, 0, 0);

        auto load_global = [&](int ks) {
            int kb = ks * 32 + k4 * 4;
            int tap, c;
            if (p.cin_shift >= 0) { tap = kb >> p.cin_shift; c = kb & ((1 << p.cin_shift) - 1); }
            else { tap = (int)fdiv((uint32_t)kb, p.dCin); c = kb - tap * p.Cin; }
            bool tv = tap < p.ntaps;
            int code = lut[tv ? tap : 0];
            int kd = code & 255, kh = (code >> 8) & 255, kw = (code >> 16) & 255;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int iz, iy, ix;
                bool ok = rv[i] && tv;
                if (p.transposed) {
                    int tz = rz[i] - kd, ty = ry[i] - kh, tx = rx[i] - kw;
                    ok = ok && ((tz | ty | tx) >= 0) && (((tz | ty | tx) & (p.stride - 1)) == 0);
                    iz = tz >> sshift; iy = ty >> sshift; ix = tx >> sshift;
                    ok = ok && iz < p.ID && iy < p.IH && ix < p.IW;
                } else {
                    iz = rz[i] + kd; iy = ry[i] + kh; ix = rx[i] + kw;
                    ok = ok && (unsigned)iz < (unsigned)p.ID && (unsigned)iy < (unsigned)p.IH &&
                         (unsigned)ix < (unsigned)p.IW;
                }
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    size_t off = ((size_t)(rb[i] + (iz * p.IH + iy) * p.IW + ix)) * p.Cin + c;
                    v = *(const float4*)(p.in + off);
                }
                av[i] = v;
            }
            {
                int idx = t;
                if (BN * 4 >= 256 || idx < BN * 4) {
                    size_t off = (size_t)(n0 + (idx >> 2)) * p.Kpad + ks * 32 + (idx & 3) * 8;
                    bh0 = *(const uint4*)(p.w_hi + off);
                    if (NSPLIT == 2) bl0 = *(const uint4*)(p.w_lo + off);
                }
            }
            if (BCH == 2) {
                int idx = t + 256;
                size_t off = (size_t)(n0 + (idx >> 2)) * p.Kpad + ks * 32 + (idx & 3) * 8;
                bh1 = *(const uint4*)(p.w_hi + off);
                if (NSPLIT == 2) bl1 = *(const uint4*)(p.w_lo + off);
            }
        };
        auto store_lds = [&](int buf) {
            char* base = smem + buf * STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (t >> 3) + 32 * i;
                int off = tile_off(row, k4 >> 1) + (k4 & 1) * 8;
                if (NSPLIT == 2) {
                    bf16x4 h, l;
                    split_bf16(av[i], h, l);
                    *(bf16x4*)(base + off) = h;
                    *(bf16x4*)(base + A_BYTES + off) = l;
                } else {
                    *(bf16x4*)(base + off) = to_bf16x4(av[i]);
                }
            }
            char* bb = base + NSPLIT * A_BYTES;
            {
                int idx = t;
                if (BN * 4 >= 256 || idx < BN * 4) {
                    int off = tile_off(idx >> 2, idx & 3);
                    *(uint4*)(bb + off) = bh0;
                    if (NSPLIT == 2) *(uint4*)(bb + B_BYTES + off) = bl0;
                }
            }
            if (BCH == 2) {
                int idx = t + 256;
                int off = tile_off(idx >> 2, idx & 3);
                *(uint4*)(bb + off) = bh1;
                if (NSPLIT == 2) *(uint4*)(bb + B_BYTES + off) = bl1;
            }
        };
        auto compute = [&](int buf) {
            const char* base = smem + buf * STAGE;
            const char* bb = base + NSPLIT * A_BYTES;
            bf16x8 ah[TM], al[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                int off = tile_off(wm * WM + a * 16 + fr, fq);
                ah[a] = *(const bf16x8*)(base + off);
                if (NSPLIT == 2) al[a] = *(const bf16x8*)(base + A_BYTES + off);
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                int off = tile_off(wn * WN + b * 16 + fr, fq);
                bf16x8 bhf = *(const bf16x8*)(bb + off);
                bf16x8 blf;
                if (NSPLIT == 2) blf = *(const bf16x8*)(bb + B_BYTES + off);
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

        load_global(0);
        store_lds(0);
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk) load_global(ks + 1);      // issue early: latency hides under the MFMAs
            compute(ks & 1);
            if (ks + 1 < nk) store_lds((ks + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue: mask / bias / activation / accumulate / per-tile BatchNorm partial sums
    float cs[TN], cq[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) { cs[b] = 0.f; cq[b] = 0.f; }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = wm * WM + a * 16 + fq * 4 + r;
            int m = m0 + row;
            if (m < p.M) {
                bool live = p.row_mask ? (p.row_mask[m] != 0) : true;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    int n = n0 + wn * WN + b * 16 + fr;
                    float v = live ? acc[a][b][r] : 0.f;
                    if (p.bias) v += p.bias[n];
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    else if (p.act == 2) v = tanhf(v);
                    if (!live) v = 0.f;
                    size_t o = (size_t)m * p.Cout + n;
                    if (p.accumulate) v += p.out[o];
                    p.out[o] = v;
                    cs[b] += v;
                    cq[b] += v * v;
                }
            }
        }
    }
    if (p.stats) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            cs[b] += __shfl_xor(cs[b], 16); cs[b] += __shfl_xor(cs[b], 32);
            cq[b] += __shfl_xor(cq[b], 16); cq[b] += __shfl_xor(cq[b], 32);
            if (fq == 0) {
                int col = wn * WN + b * 16 + fr;
                red[(wm * BN + col) * 2 + 0] = cs[b];
                red[(wm * BN + col) * 2 + 1] = cq[b];
            }
        }
        __syncthreads();
        if (t < BN) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) { s += red[(w * BN + t) * 2]; q += red[(w * BN + t) * 2 + 1]; }
            p.stats[((size_t)mtile * 2 + 0) * p.Cout + n0 + t] = s;
            p.stats[((size_t)mtile * 2 + 1) * p.Cout + n0 + t] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight packing
// dst[row][tap * inner_pad + i] (bf16 hi / lo, zero padded to Kpad) from an fp32 tensor addressed by strides.
// forward:  row = co, inner = ci;   dgrad: row = ci, inner = co  (same tensor, swapped strides).
__global__ void weight_prep_kernel(const float* __restrict__ w, long s_row, long s_tap, long s_inner, int rows, int ntaps,
                                   int inner, int inner_pad, int Kpad, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)rows * Kpad;
    if (idx >= total) return;
    int row = (int)(idx / Kpad), k = (int)(idx - (long)row * Kpad);
    int tap = k / inner_pad, i = k - tap * inner_pad;
    float v = 0.f;
    if (tap < ntaps && i < inner) v = w[row * s_row + tap * s_tap + i * s_inner];
    bf16_t h = (bf16_t)v;
    hi[idx] = h;
    if (lo) lo[idx] = (bf16_t)(v - (float)h);
}

extern "C" int tri_weight_prep(const float* w, long s_row, long s_tap, long s_inner, int rows, int ntaps, int inner,
                               int inner_pad, void* w_hi, void* w_lo, void* stream) {
    if (inner_pad % 4 != 0 || inner > inner_pad) { tri_set_error("tri_weight_prep: inner_pad must be a multiple of 4 >= inner"); return TRI_ERR_ARG; }
    int Kpad = (ntaps * inner_pad + 31) / 32 * 32;
    long total = (long)rows * Kpad;
    int blocks = (int)((total + 255) / 256);
    weight_prep_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(w, s_row, s_tap, s_inner, rows, ntaps, inner, inner_pad, Kpad,
                                                                (bf16_t*)w_hi, (bf16_t*)w_lo);
    return tri_check_launch("tri_weight_prep");
}

// ------------------------------------------------------------------------------------------------------ launcher

static int ilog2_exact(int v) {
    for (int s = 0; s < 31; ++s) if ((1 << s) == v) return s;
    return -1;
}

template <int BN, int NSPLIT>
static int launch_conv(const ConvArgs& a, hipStream_t stream) {
    constexpr int STAGE = NSPLIT * (128 * 64 + BN * 64);
    constexpr int WAVES_M = (BN >= 64) ? 2 : 4;
    size_t smem = 2 * STAGE + 256 + WAVES_M * BN * 2 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_igemm_kernel<BN, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    int mt = (a.M + 127) / 128, nt = a.Cout / BN;
    conv_igemm_kernel<BN, NSPLIT><<<mt * nt, 256, smem, stream>>>(a);
    return tri_check_launch("tri_conv");
}

static int conv_dispatch(ConvArgs& a, hipStream_t stream) {
    if (a.Cin % 4 != 0) { tri_set_error("conv: stored input channels must be a multiple of 4"); return TRI_ERR_ARG; }
    if (a.Cout % 32 != 0) { tri_set_error("conv: output channels must be a multiple of 32"); return TRI_ERR_ARG; }
    if (a.ntaps > 64) { tri_set_error("conv: more than 64 taps unsupported"); return TRI_ERR_UNSUPPORTED; }
    if (a.stride != 1 && a.stride != 2) { tri_set_error("conv: stride must be 1 or 2"); return TRI_ERR_UNSUPPORTED; }
    a.Kpad = (a.ntaps * a.Cin + 31) / 32 * 32;
    a.cin_shift = ilog2_exact(a.Cin);
    a.dOW = make_fastdiv(a.OW); a.dOH = make_fastdiv(a.OH); a.dOD = make_fastdiv(a.OD); a.dCin = make_fastdiv(a.Cin);
    bool split = a.w_lo != nullptr;
    if (a.Cout % 128 == 0) return split ? launch_conv<128, 2>(a, stream) : launch_conv<128, 1>(a, stream);
    if (a.Cout % 64 == 0) return split ? launch_conv<64, 2>(a, stream) : launch_conv<64, 1>(a, stream);
    return split ? launch_conv<32, 2>(a, stream) : launch_conv<32, 1>(a, stream);
}

extern "C" int tri_conv_kpad(int ntaps, int cin_stored) { return (ntaps * cin_stored + 31) / 32 * 32; }

extern "C" int tri_conv_num_mtiles(const TriConvDesc* d) {
    long M = (long)d->B * d->OD * d->OH * d->OW;
    return (int)((M + 127) / 128);
}

// out[B,OD,OH,OW,Cout] = conv(in[B,ID,IH,IW,Cin], W) (+bias, act 0 none / 1 relu / 2 tanh); rows with row_mask==0 are
// written as zeros (submanifold rule); stats != NULL receives per-128-row-tile column sums and sums of squares.
extern "C" int tri_conv_fwd(const TriConvDesc* d, const float* in, const void* w_hi, const void* w_lo, float* out,
                            const uint8_t* row_mask, const float* bias, int act, int accumulate, float* stats, void* stream) {
    ConvArgs a{};
    a.in = in; a.w_hi = (const bf16_t*)w_hi; a.w_lo = (const bf16_t*)w_lo; a.out = out;
    a.row_mask = row_mask; a.bias = bias; a.stats = stats;
    a.B = d->B; a.ID = d->ID; a.IH = d->IH; a.IW = d->IW; a.Cin = d->Cin;
    a.OD = d->OD; a.OH = d->OH; a.OW = d->OW; a.Cout = d->Cout;
    a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pd = d->pad_d; a.ph = d->pad_h; a.pw = d->pad_w;
    a.transposed = 0; a.act = act; a.accumulate = accumulate;
    a.ntaps = d->KD * d->KH * d->KW;
    a.M = d->B * d->OD * d->OH * d->OW;
    return conv_dispatch(a, (hipStream_t)stream);
}

// din[B,ID,IH,IW,Cin] (+)= conv_transpose(dout[B,OD,OH,OW,Cout], Wt), Wt packed [Cin][taps*Cout] by tri_weight_prep
// with swapped strides.  `d` is the FORWARD descriptor of the layer.
extern "C" int tri_conv_dgrad(const TriConvDesc* d, const float* dout, const void* wt_hi, const void* wt_lo, float* din,
                              const uint8_t* row_mask, int accumulate, void* stream) {
    ConvArgs a{};
    a.in = dout; a.w_hi = (const bf16_t*)wt_hi; a.w_lo = (const bf16_t*)wt_lo; a.out = din;
    a.row_mask = row_mask; a.bias = nullptr; a.stats = nullptr;
    a.B = d->B; a.ID = d->OD; a.IH = d->OH; a.IW = d->OW; a.Cin = d->Cout;       // gather source = dout grid
    a.OD = d->ID; a.OH = d->IH; a.OW = d->IW; a.Cout = d->Cin;                   // rows = input positions
    a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pd = d->pad_d; a.ph = d->pad_h; a.pw = d->pad_w;
    a.transposed = 1; a.act = 0; a.accumulate = accumulate;
    a.ntaps = d->KD * d->KH * d->KW;
    a.M = d->B * d->ID * d->IH * d->IW;
    return conv_dispatch(a, (hipStream_t)stream);
}
