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
#include <stdlib.h>
#include <map>
#include <mutex>
#include "../../include/tricolo_hip.h"
#include "conv_vox.h"

struct ConvArgs {
    const void* in;            // activations: fp32 or bf16 (kernel template parameter AT)
    const void* w_hi;          // packed operand rows [Cout][Kpad]: bf16 (hi), or f16 with f16 activation storage
    const void* w_lo;          // bf16 lo part (3-product mode) or NULL
    void* out;
    const uint8_t* row_mask;   // per output position; 0 -> row forced to zero, all-zero tiles are skipped
    const int* row_pos;        // optional row -> output position table (no split-K): rows may be visited in any order
    const int* row_count;      // optional DEVICE int: only rows [0, *row_count) of row_pos exist (compact active-site list)
    const float* bias;
    float* stats;              // [num_mtiles][2][Cout] per-tile column sum / sum of squares (BatchNorm statistics)
    float* slab;               // split-K partial sums [ksplit][M][Cout] (ksplit > 1)
    int B, ID, IH, IW, Cin;
    int OD, OH, OW, Cout;
    int KD, KH, KW, stride, pd, ph, pw;
    int transposed, act, accumulate;
    int Kpad, M, ntaps, cin_shift, ksplit, steps_per_split;
    unsigned in_bytes;
    int nunits;
    int h_tr, h_rows, h_slab_bytes, h_nr, h_mtiles;    // halo kernel: image rows per tile, slab rows, slab bytes, ring slots, row tiles
    int h_dbuf;
    int h_abl;                                          // tuning aid (TRICOLO_HALO_ABL): ablation bits, 0 in production
    int h_xcg, h_touch;                                 // conv_halo_rows_kernel: channel tiles per XCD block (0: linear tile order); L2 warm-up of the weights
    int h_swz;                                          // halo kernels: slab chunk swizzle of pixel (row, col) = (col + h_swz * row) & 7 (halo_swizzle())
#ifdef HALO_STAMPS
    long long* h_dbg;                                   // tools/probes/halo_probe.hip: per-workgroup (id, cycle) stamps of wave 0
#endif
    FastDiv dOW, dOH, dOD, dCin, dP, dH2;
};

// v rotated right by N lanes inside its row of 16 lanes (DPP row_ror)
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}

// Shared epilogue: acc[a][b][r] = out[m = a-tile row (lane & 15)][n = b-tile col 4 * (lane >> 4) + r].
template <typename AT, int BN, int TM, int TN, int WAVES_M, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x4 (&acc)[TM][TN], int m0, int n0, int mtile, int wm, int wn, int fr,
                                              int fq, int t, int split, int any_active, float* red, int Meff) {
    if (p.ksplit > 1) {
        // split-K: raw partial sums only; conv_splitk_finish_kernel applies mask / bias / activation / statistics
        if (!any_active) return;
        float* slab = p.slab + (size_t)split * p.M * p.Cout;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            int m = m0 + wm * WM + a * 16 + fr;
            if (m < Meff)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    int n = n0 + wn * WN + b * 16 + fq * 4;
                    *(f32x4*)(slab + (size_t)m * p.Cout + n) = acc[a][b];
                }
        }
        return;
    }

    // ---- epilogue: mask / bias / activation / accumulate / per-tile BatchNorm partial sums; 16-byte stores.
    // The conv+BN case (no bias, no activation, no accumulate) takes a branch-free path.
    const bool plain = (p.bias == nullptr) && (p.act == 0) && (p.accumulate == 0);
    f32x4 cs[TN], cq[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) { cs[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; cq[b] = cs[b]; }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = m0 + wm * WM + a * 16 + fr;
        if (m < Meff) {
            const float live = (p.row_mask && p.row_mask[m] == 0) ? 0.f : 1.f;
            const size_t mo = p.row_pos ? (size_t)p.row_pos[m] : (size_t)m;        // where this row lives in the output tensor
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int n = n0 + wn * WN + b * 16 + fq * 4;
                f32x4 v = acc[a][b];
                AT* o = (AT*)p.out + mo * p.Cout + n;
                if (!plain) {
                    if (p.bias) v += *(const f32x4*)(p.bias + n);
                    if (p.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    else if (p.act == 2) { v[0] = tanhf(v[0]); v[1] = tanhf(v[1]); v[2] = tanhf(v[2]); v[3] = tanhf(v[3]); }
                    v *= live;
                    if (p.accumulate) { float4 e = Act<AT>::ld4(o); v[0] += e.x; v[1] += e.y; v[2] += e.z; v[3] += e.w; }
                } else {
                    v *= live;
                }
                Act<AT>::st4(o, make_float4(v[0], v[1], v[2], v[3]));
                if (sizeof(AT) == 2) {                            // statistics of what BatchNorm will actually read back
                    v[0] = Act<AT>::rnd(v[0]); v[1] = Act<AT>::rnd(v[1]); v[2] = Act<AT>::rnd(v[2]); v[3] = Act<AT>::rnd(v[3]);
                }
                cs[b] += v;
                cq[b] += v * v;
            }
        }
    }
    if (p.stats) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = cs[b][r], q = cq[b][r];
                // the 16 rows of a fragment are the 16 lanes of one DPP row: rotate-and-add on the VALU instead of 8
                // ds_bpermute round trips per column (measured: statistics were 3,300 of a tile's ~31,000 cycles)
                s += row_ror<8>(s); q += row_ror<8>(q);
                s += row_ror<4>(s); q += row_ror<4>(q);
                s += row_ror<2>(s); q += row_ror<2>(q);
                s += row_ror<1>(s); q += row_ror<1>(q);
                if (fr == 0) {
                    int col = wn * WN + b * 16 + fq * 4 + r;
                    red[(wm * BN + col) * 2 + 0] = s;
                    red[(wm * BN + col) * 2 + 1] = q;
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

// im2col gather, per thread: 4 rows x one float4 of k per k-step.  Everything that depends only on the ROW is
// computed once (element offset of the row's origin voxel + a packed per-axis validity mask, 8 bits per axis);
// everything that depends only on the TAP comes from a 64-entry LDS table.  A load is then
//   voffset = valid ? (rowoff + tapoff + c) * 4 : OUT_OF_RANGE   ->  raw buffer load (hardware returns 0 out of range)
// i.e. ~5 VALU instructions instead of a full coordinate / bounds / address recomputation per load.
// taps k in [0,K) with 0 <= x0 + k < I
__device__ __forceinline__ unsigned axis_mask(int x0, int K, int I) {
    int lo = max(0, -x0), hi = min(K, I - x0);
    return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
}
// taps k with t = r - k >= 0, t % stride == 0, t / stride < I
__device__ __forceinline__ unsigned axis_mask_t(int r, int K, int I, int stride) {
    int hi = min(K, r + 1);                                   // k <= r
    int lo = max(0, r - (I - 1) * stride);                    // (r - k) / stride <= I - 1
    unsigned m = hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
    if (stride == 2) m &= (r & 1) ? 0xAAu : 0x55u;            // k must have the parity of r
    return m;
}

template <int BN, int NSPLIT, typename AT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int BM = 128;
    constexpr int WAVES_N = (BN >= 64) ? 2 : 1, WAVES_M = 4 / WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
    constexpr int STAGE = NSPLIT * (A_BYTES + B_BYTES);
    constexpr int BCH = (BN * 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lut_off = (int*)(smem + 2 * STAGE);             // [64] element offset of the tap
    int* lut_sh = lut_off + 64;                          // [64] packed shifts: kw | (8+kh)<<8 | (16+kd)<<16
    float* red = (float*)(smem + 2 * STAGE + 512);       // [WAVES_M][BN][2]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int NT = p.Cout / BN;
    // compact row lists fill only the first ceil(count / 128) tiles: the XCD remap would hand that contiguous range to ONE XCD
    // (measured: level 0 at 64^3 x 64 took 0.88 ms on an eighth of the chip against 0.67 ms for the 2.6x more tiles of the
    // tile-skipping path), so those launches keep the dispatcher's round-robin order
    const int wg = p.row_count ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = wg / NT, ntile = wg - mtile * NT;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int sshift = (p.stride == 2) ? 1 : 0;
    const int Meff = p.row_count ? min(*p.row_count, p.M) : p.M;
    if (m0 >= Meff) {                                   // tile past the end of the compact row list: empty statistics record, nothing else
        if (p.stats && p.ksplit == 1 && t < BN) {
            p.stats[((size_t)mtile * 2 + 0) * p.Cout + n0 + t] = 0.f;
            p.stats[((size_t)mtile * 2 + 1) * p.Cout + n0 + t] = 0.f;
        }
        return;
    }

    if (t < 64) {
        int kd = 0, kh = 0, kw = 0;
        if (t < p.ntaps) {
            kw = t % p.KW;
            int r = t / p.KW;
            kh = r % p.KH;
            kd = r / p.KH;
        }
        lut_sh[t] = kw | ((8 + kh) << 8) | ((16 + kd) << 16);
        int off = p.transposed ? -((((kd >> sshift) * p.IH + (kh >> sshift)) * p.IW + (kw >> sshift)) * p.Cin)
                               : (((kd * p.IH + kh) * p.IW + kw) * p.Cin);
        lut_off[t] = off;
    }

    const int k4 = t & 7;
    int rowoff[4];
    unsigned rmask[4];
    int any_active = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + (t >> 3) + 32 * i;
        bool valid = m < Meff;
        uint32_t mm = valid ? (uint32_t)(p.row_pos ? p.row_pos[m] : m) : 0u;      // output position of this tile row
        uint32_t q1 = fdiv(mm, p.dOW);
        int ow = mm - q1 * p.OW;
        uint32_t q2 = fdiv(q1, p.dOH);
        int oh = q1 - q2 * p.OH;
        uint32_t b = fdiv(q2, p.dOD);
        int od = q2 - b * p.OD;
        if (p.row_mask) valid = valid && (p.row_mask[mm] != 0);
        any_active |= valid ? 1 : 0;
        // per-axis validity bits in closed form: the valid taps of an axis form an interval (AND a parity class when
        // the transposed gather has stride 2), so no per-tap loop is needed
        unsigned mk;
        int z0, y0, x0;
        if (p.transposed) {
            int rz = od + p.pd, ry = oh + p.ph, rx = ow + p.pw;
            z0 = rz >> sshift; y0 = ry >> sshift; x0 = rx >> sshift;
            mk = axis_mask_t(rx, p.KW, p.IW, p.stride) | (axis_mask_t(ry, p.KH, p.IH, p.stride) << 8) |
                 (axis_mask_t(rz, p.KD, p.ID, p.stride) << 16);
        } else {
            z0 = od * p.stride - p.pd; y0 = oh * p.stride - p.ph; x0 = ow * p.stride - p.pw;
            mk = axis_mask(x0, p.KW, p.IW) | (axis_mask(y0, p.KH, p.IH) << 8) | (axis_mask(z0, p.KD, p.ID) << 16);
        }
        rmask[i] = valid ? mk : 0u;
        rowoff[i] = ((((int)b * p.ID + z0) * p.IH + y0) * p.IW + x0) * p.Cin;
    }
    any_active = __syncthreads_or(any_active);       // also publishes the tap tables

    // accumulators hold D^T tiles: acc[a][b][r] = out[m = a-tile row (lane&15)][n = b-tile col 4*(lane>>4) + r]
    // (weights are the MFMA "A" operand, activations the "B" operand) so the epilogue stores 16 bytes per lane.
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;
    const int split = blockIdx.y;

    if (any_active) {
        const int nk_total = p.Kpad >> 5;
        const int ks0 = split * p.steps_per_split;
        const int ks1 = min(nk_total, ks0 + p.steps_per_split);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
        uint4 av[4];
        uint4 bh0, bh1, bl0, bl1;       // named, not an array: hipcc keeps conditionally-written arrays in scratch
        bh0 = bh1 = bl0 = bl1 = make_uint4(0, 0, 0, 0);

        auto load_global = [&](int ks) {
            int kb = ks * 32 + k4 * 4;
            int tap, c;
            if (p.cin_shift >= 0) { tap = kb >> p.cin_shift; c = kb & ((1 << p.cin_shift) - 1); }
            else { tap = (int)fdiv((uint32_t)kb, p.dCin); c = kb - tap * p.Cin; }
            bool tv = tap < p.ntaps;
            int tsel = tv ? tap : 0;
            int toff = lut_off[tsel] + c;
            int sh = lut_sh[tsel];
            int sx = sh & 255, sy = (sh >> 8) & 255, sz = (sh >> 16) & 255;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned mk = rmask[i];
                bool ok = tv && (((mk >> sx) & (mk >> sy) & (mk >> sz)) & 1u);
                if (sizeof(AT) == 4) {
                    unsigned voff = ok ? (unsigned)((rowoff[i] + toff) << 2) : 0x80000000u;
                    av[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
                } else {                                          // bf16 activations: 4 channels = 8 bytes, already MFMA operands
                    unsigned voff = ok ? (unsigned)((rowoff[i] + toff) << 1) : 0x80000000u;
                    uint2 h = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0));
                    av[i] = make_uint4(h.x, h.y, 0, 0);
                }
            }
            {
                int idx = t;
                if (BN * 4 >= 256 || idx < BN * 4) {
                    size_t off = (size_t)(n0 + (idx >> 2)) * p.Kpad + ks * 32 + (idx & 3) * 8;
                    bh0 = *(const uint4*)((const uint16_t*)p.w_hi + off);
                    if (NSPLIT == 2) bl0 = *(const uint4*)((const uint16_t*)p.w_lo + off);
                }
            }
            if (BCH == 2) {
                int idx = t + 256;
                size_t off = (size_t)(n0 + (idx >> 2)) * p.Kpad + ks * 32 + (idx & 3) * 8;
                bh1 = *(const uint4*)((const uint16_t*)p.w_hi + off);
                if (NSPLIT == 2) bl1 = *(const uint4*)((const uint16_t*)p.w_lo + off);
            }
        };
        auto store_lds = [&](int buf) {
            char* base = smem + buf * STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = (t >> 3) + 32 * i;
                int off = tile_off(row, k4 >> 1) + (k4 & 1) * 8;
                if (sizeof(AT) == 2) {
                    *(uint2*)(base + off) = make_uint2(av[i].x, av[i].y);
                    if (NSPLIT == 2) *(uint2*)(base + A_BYTES + off) = make_uint2(0, 0);
                } else {
                    float4 v = __builtin_bit_cast(float4, av[i]);
                    if (NSPLIT == 2) {
                        bf16x4 h, l;
                        split_bf16(v, h, l);
                        *(bf16x4*)(base + off) = h;
                        *(bf16x4*)(base + A_BYTES + off) = l;
                    } else {
                        *(bf16x4*)(base + off) = to_bf16x4(v);
                    }
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
            typedef Mma<typename OpOf<AT>::E> MM;
            typedef typename MM::v8 v8;
            v8 ah[TM], al[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                int off = tile_off(wm * WM + a * 16 + fr, fq);
                ah[a] = *(const v8*)(base + off);
                if (NSPLIT == 2) al[a] = *(const v8*)(base + A_BYTES + off);
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                int off = tile_off(wn * WN + b * 16 + fr, fq);
                v8 bhf = *(const v8*)(bb + off);
                v8 blf;
                if (NSPLIT == 2) blf = *(const v8*)(bb + B_BYTES + off);
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    if (NSPLIT == 2) {
                        acc[a][b] = MM::mma(bhf, al[a], acc[a][b]);
                        acc[a][b] = MM::mma(blf, ah[a], acc[a][b]);
                    }
                    acc[a][b] = MM::mma(bhf, ah[a], acc[a][b]);
                }
            }
        };

        if (ks0 < ks1) {
            load_global(ks0);
            store_lds(0);
            __syncthreads();
            for (int ks = ks0; ks < ks1; ++ks) {
                const int buf = (ks - ks0) & 1;
                if (ks + 1 < ks1) load_global(ks + 1);      // issue early: latency hides under the MFMAs
                compute(buf);
                if (ks + 1 < ks1) store_lds(buf ^ 1);
                __syncthreads();
            }
        }
    }

    conv_epilogue<AT, BN, TM, TN, WAVES_M, WM, WN>(p, acc, m0, n0, mtile, wm, wn, fr, fq, t, split, any_active, red, Meff);
}

// byte offset of 16-byte chunk `chunk` of row `row` in a [rows][64 k] bf16 / f16 tile (128-B rows: 8 chunks, XOR with (row / 2) % 8)
__device__ __forceinline__ int dma_off(int row, int chunk) { return row * 128 + (((chunk ^ (row >> 1)) & 7) << 4); }

// ================================================================================================ LDS-DMA kernel
// bf16 activation storage + Cin % 64 == 0 (every 3x3 / 1x1 layer of ResNet-18 past the stem, voxel levels 2-4 and all
// their data gradients): the operands in HBM already ARE the MFMA operand bytes, so both tiles go global -> LDS with
// buffer_load_dwordx4 ... lds (no VGPR round trip, no cvt, no ds_write), BK = 64 (half the barriers / address work per
// MFMA, 16-32 MFMAs per wave-iteration) and the DMA of step k+1 in flight under the MFMAs of step k.
// One wave-instruction writes 1 KiB = 8 tile rows x 128 B contiguously (LDS address = wave base + lane * 16), so the
// bank swizzle is applied on the SOURCE side: lane (row, slot) fetches chunk slot ^ ((row >> 1) & 7) and the
// fragment reads apply the same XOR (cdna_hip_programming.md rule 21).  Invalid (row, tap) pairs use an out-of-range
// buffer offset, for which the buffer unit returns - and writes to LDS - zeros.
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define TRI_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        TRI_W(0) TRI_W(1) TRI_W(2) TRI_W(3) TRI_W(4) TRI_W(5) TRI_W(6) TRI_W(7) TRI_W(8) TRI_W(9) TRI_W(10) TRI_W(11) TRI_W(12)
        TRI_W(13) TRI_W(14) TRI_W(15) TRI_W(16) TRI_W(17) TRI_W(18) TRI_W(19) TRI_W(20) TRI_W(21) TRI_W(22) TRI_W(23) TRI_W(24)
        TRI_W(25) TRI_W(26) TRI_W(27) TRI_W(28) TRI_W(29) TRI_W(30) TRI_W(31) TRI_W(32)
#undef TRI_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int BN, int NST, typename AT>
__global__ __launch_bounds__(256) void conv_dma_kernel(const ConvArgs p) {
    typedef Mma<typename OpOf<AT>::E> MM;
    typedef typename MM::v8 v8;
    constexpr int BM = 128, BK = 64;
    constexpr int WAVES_N = 2, WAVES_M = 2;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lut_off = (int*)(smem + NST * STAGE);
    int* lut_sh = lut_off + 64;
    float* red = (float*)(smem + NST * STAGE + 512);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int NT = p.Cout / BN;
    // compact row lists fill only the first ceil(count / 128) tiles: the XCD remap would hand that contiguous range to ONE XCD
    // (measured: level 0 at 64^3 x 64 took 0.88 ms on an eighth of the chip against 0.67 ms for the 2.6x more tiles of the
    // tile-skipping path), so those launches keep the dispatcher's round-robin order
    const int wg = p.row_count ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = wg / NT, ntile = wg - mtile * NT;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int sshift = (p.stride == 2) ? 1 : 0;
    const int Meff = p.row_count ? min(*p.row_count, p.M) : p.M;
    if (m0 >= Meff) {                                   // tile past the end of the compact row list (see conv_igemm_kernel)
        if (p.stats && p.ksplit == 1 && t < BN) {
            p.stats[((size_t)mtile * 2 + 0) * p.Cout + n0 + t] = 0.f;
            p.stats[((size_t)mtile * 2 + 1) * p.Cout + n0 + t] = 0.f;
        }
        return;
    }

    if (t < 64) {
        int kd = 0, kh = 0, kw = 0;
        if (t < p.ntaps) {
            kw = t % p.KW;
            int r = t / p.KW;
            kh = r % p.KH;
            kd = r / p.KH;
        }
        lut_sh[t] = kw | ((8 + kh) << 8) | ((16 + kd) << 16);
        lut_off[t] = p.transposed ? -((((kd >> sshift) * p.IH + (kh >> sshift)) * p.IW + (kw >> sshift)) * p.Cin)
                                  : (((kd * p.IH + kh) * p.IW + kw) * p.Cin);
    }

    // DMA rows of this lane: row = (t >> 3) + 32 i, slot = t & 7 (16-byte chunk position inside the 128-B LDS row)
    const int slot = t & 7;
    int rowoff[4];
    unsigned rmask[4];
    int any_active = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = (t >> 3) + 32 * i;
        int m = m0 + row;
        bool valid = m < Meff;
        uint32_t mm = valid ? (uint32_t)(p.row_pos ? p.row_pos[m] : m) : 0u;       // output position of this tile row
        uint32_t q1 = fdiv(mm, p.dOW);
        int ow = mm - q1 * p.OW;
        uint32_t q2 = fdiv(q1, p.dOH);
        int oh = q1 - q2 * p.OH;
        uint32_t b = fdiv(q2, p.dOD);
        int od = q2 - b * p.OD;
        if (p.row_mask) valid = valid && (p.row_mask[mm] != 0);
        any_active |= valid ? 1 : 0;
        unsigned mk;
        int z0, y0, x0;
        if (p.transposed) {
            int rz = od + p.pd, ry = oh + p.ph, rx = ow + p.pw;
            z0 = rz >> sshift; y0 = ry >> sshift; x0 = rx >> sshift;
            mk = axis_mask_t(rx, p.KW, p.IW, p.stride) | (axis_mask_t(ry, p.KH, p.IH, p.stride) << 8) |
                 (axis_mask_t(rz, p.KD, p.ID, p.stride) << 16);
        } else {
            z0 = od * p.stride - p.pd; y0 = oh * p.stride - p.ph; x0 = ow * p.stride - p.pw;
            mk = axis_mask(x0, p.KW, p.IW) | (axis_mask(y0, p.KH, p.IH) << 8) | (axis_mask(z0, p.KD, p.ID) << 16);
        }
        rmask[i] = valid ? mk : 0u;
        // byte offset of the chunk this lane fetches for that row (tap / channel-step offsets are added per k-step)
        rowoff[i] = (((((int)b * p.ID + z0) * p.IH + y0) * p.IW + x0) * p.Cin + (slot ^ ((row >> 1) & 7)) * 8) * 2;
    }
    // Rows visited out of order (row_pos: the stride-2 data gradient sorts them by tap-parity class) share few taps per tile.
    // The per-axis OR of the rows' validity bits bounds the taps any row of the tile can use; k-steps of other taps are
    // dropped from this workgroup's step list (exactly the 5-8 of 9 taps whose parity cannot match for a one-class tile).
    __shared__ unsigned wave_axes[4];
    __shared__ int live_steps[256], nlive_s;
    if (p.row_pos) {
        unsigned ax = rmask[0] | rmask[1] | rmask[2] | rmask[3];
        ax |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)ax, 0x128, 0xf, 0xf, false);
        ax |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)ax, 0x124, 0xf, 0xf, false);
        ax |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)ax, 0x122, 0xf, 0xf, false);
        ax |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)ax, 0x121, 0xf, 0xf, false);
        ax = (unsigned)(__builtin_amdgcn_readlane((int)ax, 0) | __builtin_amdgcn_readlane((int)ax, 16) |
                        __builtin_amdgcn_readlane((int)ax, 32) | __builtin_amdgcn_readlane((int)ax, 48));
        if (lane == 0) wave_axes[wave] = ax;
    }
    any_active = __syncthreads_or(any_active);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;
    const int split = blockIdx.y;

    if (any_active) {
        const int nk_total = p.Kpad >> 6;
        const int ks0 = split * p.steps_per_split;
        const int ks1 = min(nk_total, ks0 + p.steps_per_split);
        const v4i rsrc = make_rsrc_words(p.in, p.in_bytes);
        const v4i wrsrc = make_rsrc_words((const uint16_t*)p.w_hi + (size_t)n0 * p.Kpad, (unsigned)(BN * p.Kpad * 2));
        const unsigned lds0 = lds_addr(smem) + wave * 1024;       // this wave's 1 KiB slice of every 4 KiB DMA group
        int woff[BN / 32];
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            int n = (t >> 3) + 32 * i;
            woff[i] = (n * p.Kpad + (slot ^ ((n >> 1) & 7)) * 8) * 2;
        }

        auto issue = [&](int ks, int buf) {
            const unsigned base = lds0 + buf * STAGE;
            const int kb = ks * BK;
            int tap, c0;
            if (p.cin_shift >= 0) { tap = kb >> p.cin_shift; c0 = kb & ((1 << p.cin_shift) - 1); }
            else { tap = (int)fdiv((uint32_t)kb, p.dCin); c0 = kb - tap * p.Cin; }
            const bool tv = tap < p.ntaps;
            const int toff = (lut_off[tv ? tap : 0] + c0) * 2;
            const int sh = lut_sh[tv ? tap : 0];
            const int sx = sh & 255, sy = (sh >> 8) & 255, sz = (sh >> 16) & 255;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned mk = rmask[i];
                bool ok = tv && (((mk >> sx) & (mk >> sy) & (mk >> sz)) & 1u);
                int voff = ok ? rowoff[i] + toff : (int)0x80000000;         // out of range -> zeros land in LDS, branch-free
                dma16_async(rsrc, base + i * 4096, voff);
            }
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) {
                dma16_async(wrsrc, base + A_BYTES + i * 4096, woff[i] + kb * 2);
            }
        };
        auto compute = [&](int buf) {
            const char* base = smem + buf * STAGE;
            const char* bb = base + A_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                v8 ah[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) ah[a] = *(const v8*)(base + dma_off(wm * WM + a * 16 + fr, kk * 4 + fq));
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    v8 bhf = *(const v8*)(bb + dma_off(wn * WN + b * 16 + fr, kk * 4 + fq));
#pragma unroll
                    for (int a = 0; a < TM; ++a) acc[a][b] = MM::mma(bhf, ah[a], acc[a][b]);
                }
            }
        };

        int nl = ks1 - ks0;
        if (p.row_pos) {
            if (wave == 0) {
                const unsigned axes = wave_axes[0] | wave_axes[1] | wave_axes[2] | wave_axes[3];
                int count = 0;
                for (int base = 0; base < ks1 - ks0; base += 64) {
                    const int ks = ks0 + base + lane;
                    bool ok = false;
                    if (ks < ks1) {
                        const int kb = ks * BK;
                        const int tap = p.cin_shift >= 0 ? (kb >> p.cin_shift) : (int)fdiv((uint32_t)kb, p.dCin);
                        if (tap < p.ntaps) {
                            const int sh = lut_sh[tap];
                            ok = ((axes >> (sh & 255)) & (axes >> ((sh >> 8) & 255)) & (axes >> ((sh >> 16) & 255))) & 1u;
                        }
                    }
                    const unsigned long long bal = __ballot(ok);
                    if (ok && count + __popcll(bal & ((1ull << lane) - 1ull)) < 256)
                        live_steps[count + __popcll(bal & ((1ull << lane) - 1ull))] = ks;
                    count += __popcll(bal);
                }
                if (lane == 0) nlive_s = count;
            }
            __syncthreads();
            nl = nlive_s;
        }
        auto step_at = [&](int j) { return p.row_pos ? live_steps[j] : ks0 + j; };
        if (nl > 0) {
            // NST - 1 stages in flight ahead of the MFMAs.  The DMAs are asm-issued (common.h) and complete in order, so
            // "stage ks has landed" == "at most the PER instructions of each younger stage are still outstanding".
            constexpr int PER = 4 + BN / 32;
#pragma unroll
            for (int d = 0; d < NST - 1; ++d)
                if (d < nl) issue(step_at(d), d);
            int buf = 0;
            for (int j = 0; j < nl; ++j) {
                const int younger = min(nl - 1 - j, NST - 2);
                if (NST == 2 || younger == 0) wait_vmcnt(0);
                else if (younger == 1) wait_vmcnt(PER);
                else wait_vmcnt(2 * PER);
                __builtin_amdgcn_s_barrier();                            // everybody's stage j is visible; all reads of stage j - 1 are done
                asm volatile("" ::: "memory");
                if (j + NST - 1 < nl) {
                    int nb = buf + NST - 1;
                    if (nb >= NST) nb -= NST;
                    issue(step_at(j + NST - 1), nb);                     // refills the buffer stage j - 1 was computed from
                }
                compute(buf);
                if (++buf == NST) buf = 0;
            }
        }
        __syncthreads();
    }
    conv_epilogue<AT, BN, TM, TN, WAVES_M, WM, WN>(p, acc, m0, n0, mtile, wm, wn, fr, fq, t, split, any_active, red, Meff);
}


// ================================================================================================ halo kernel (2D, 3x3 / 1 / pad 1)
// Every BasicBlock conv of ResNet-18 that keeps its resolution (16 of the 20 convs, and their data gradients) is a 3x3,
// stride-1, pad-1 convolution.  conv_dma_kernel gathers their im2col rows tap by tap: each input pixel crosses the L2 -> LDS
// path 9 times and every 128 x 64 tile re-streams the whole weight panel (24 KB per MFLOP).  Here a PERSISTENT workgroup
// (up to three per CU: the slab of a 128-position tile is ~25 KB) walks tiles of TR whole image rows (64 * TM positions) x 64
// output channels:
//   * the input rows a tile needs (TR rows + a halo row above and below every image segment, a zero pixel left and right of
//     every row) are DMA'd ONCE per 64-channel chunk into an LDS slab - borders, image boundaries inside the tile and rows past
//     the tensor are out-of-range fetches, i.e. zeros in LDS, so the MFMA loop has no masks: tap (dy, dx) is the same slab read
//     shifted by dy * P + dx pixels (P = W + 2); ONE slab buffer - its refill between chunks is exposed to this workgroup and
//     hidden by the co-resident ones; the next TILE's slab is fetched under this tile's epilogue;
//   * weights stream through a ring of three [64 x 64] (tap, chunk) units, two in flight, one barrier per unit; the stream
//     runs across tile boundaries, so the next tile's first units are in flight under this tile's epilogue.
// L2 -> LDS bytes per MFLOP drop 2.5-4x (layer1: 6.1 KB against 24 KB).  What the first two versions taught (ablations in
// profiles/r2/README.md): at these tile sizes (9 units = 4,600 MFMA cycles on layer1) the kernel is bound by INSTRUCTION ISSUE,
// not by bytes - a 2,085-instruction prologue (fast divisions for the slab map) and a 4,452-instruction generic epilogue per
// tile cost 13 + 10 us of a 34 us launch, and two co-resident workgroups run them in lockstep, not against each other's MFMAs.
// So: everything that depends only on the geometry is computed once per workgroup (slab source offsets, fragment pixels),
// every cursor is incremental (no division / modulo in the loop), ring waits are immediates, and the conv + BatchNorm epilogue
// is a dedicated lean one.
// Geometries: H % TR == 0 (a tile stays inside one image; its halo rows exist unless it touches the image's top / bottom)
// or TR % H == 0 (whole images per tile, every halo row is zero); anything else stays on conv_dma_kernel.
// Slab: pixel (slab row s, column x) at byte (s * P + x) * 128, its 16-byte chunks XOR-swizzled by (x + h_swz * s) & 7 (round 6: the
// round-2 swizzle (pixel >> 1) & 7 is conflict-free for 16 CONSECUTIVE pixels, but a fragment's 16 positions are rows of W pixels with
// the slab's two border pixels between them, and a ds_read_b128 cycle serves lanes {0-3, 12-15, 20-27} etc.: 2.0 LDS cycles per
// lane group on 4- and 8-wide maps, 1.67 on 16- and 32-wide ones - the 17-25 % of bank-conflict cycles the round-2 PMC pass counted;
// halo_swizzle() picks the per-row multiplier that makes every tap's fragment read conflict-free); one extra
// always-zero pixel after the last row serves the MFMA rows past the end of a partial last tile.
#define HALO_MAX_ROUNDS 10
#ifndef HALO_NR
#define HALO_NR 3
#endif
template <int N>
__device__ __forceinline__ void wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// column sums of the lanes' (cs, cq) -> one [2][64] record: DPP row reduction, cross-wave through LDS
template <int TN, int BN>
__device__ __forceinline__ void halo_store_stats(const f32x4* cs, const f32x4* cq, float* red, float* stats, int Cout, int rec, int ntile, int wave,
                                                 int fr, int fq, int t, bool raw_barrier) {
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s_ = cs[b][r], q_ = cq[b][r];
            s_ += row_ror<8>(s_); q_ += row_ror<8>(q_);
            s_ += row_ror<4>(s_); q_ += row_ror<4>(q_);
            s_ += row_ror<2>(s_); q_ += row_ror<2>(q_);
            s_ += row_ror<1>(s_); q_ += row_ror<1>(q_);
            if (fr == 0) {
                const int col = b * 16 + fq * 4 + r;
                red[(wave * BN + col) * 2 + 0] = s_;
                red[(wave * BN + col) * 2 + 1] = q_;
            }
        }
    if (raw_barrier) {                          // inside the tile loop: LDS-DMA in flight must not be waited for (no vmcnt wait)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
        __syncthreads();
    }
    if (t < BN) {
        float s_ = 0.f, q_ = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s_ += red[(w * BN + t) * 2]; q_ += red[(w * BN + t) * 2 + 1]; }
        stats[((size_t)rec * 2 + 0) * Cout + ntile * BN + t] = s_;
        stats[((size_t)rec * 2 + 1) * Cout + ntile * BN + t] = q_;
        // (per-tile use: the next write to `red` comes after at least nine more barriers)
    }
}

__device__ __forceinline__ v4i lds_read16(unsigned addr) {
    v4i r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
template <int OFF>
__device__ __forceinline__ v4i lds_read16_off(unsigned addr) {
    v4i r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// wait until at most N LDS operations are outstanding; the fragments named become "defined here" for the compiler, so no MFMA that
// consumes them can be scheduled above the wait
template <int N>
__device__ __forceinline__ void frag_wait(v4i (&a)[2], v4i (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void frag_wait(v4i (&a)[3], v4i (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
// In-kernel phase stamps of the halo kernels (-DHALO_STAMPS builds only: tools/probes/halo_probe.hip).  Wave 0 of every workgroup writes
// (id << 48 | s_memtime) into a 2 KiB LDS area `stl` of the kernel and copies it to p.h_dbg at the end; ~150 cycles per stamp.
#ifdef HALO_STAMPS
#define HSTAMP(id) do { if (wave == 0 && n_stamp < 255) { const long long c_ = __builtin_readcyclecounter(); if (lane == 0) stl[n_stamp] = ((long long)(id) << 48) | (c_ & 0xFFFFFFFFFFFFll); ++n_stamp; } } while (0)
#else
#define HSTAMP(id) do { } while (0)
#endif
// WGREC: BatchNorm records per persistent workgroup (launches with many tiles per workgroup) instead of per tile
template <int TM, bool WGREC, typename AT>
__global__ __launch_bounds__(256, WGREC ? 3 : 1) void conv_halo2d_kernel(const ConvArgs p) {     // WGREC: <= 170 VGPRs (three workgroups per CU)
    typedef Mma<typename OpOf<AT>::E> MM;
    typedef typename MM::v8 v8;
    constexpr int BN = 64, TN = 4, WM = 16 * TM, W_BYTES = BN * 128, NR = HALO_NR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int slab_bytes = p.h_slab_bytes;
    char* const ring = smem + slab_bytes;
    float* const red = (float*)(ring + NR * W_BYTES);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fq = lane >> 4;
#ifdef HALO_STAMPS
    int n_stamp = 0;
    long long* const stl = (long long*)(smem + p.h_slab_bytes + HALO_NR * 8192 + 2048);      // wave 0's stamps (2 KiB past `red`)
#endif
    HSTAMP(1);
    const int H = p.IH, W = p.IW, P = W + 2, TR = p.h_tr, NH = p.B * H;
    const int NT = p.Cout / BN, nchunks = p.Cin >> 6;
    const int items = p.h_mtiles * NT, G = gridDim.x;
    const int S = slab_bytes >> 12;                                     // DMA rounds (4 waves x 1 KiB) per slab
    const int zero_pix = p.h_rows * P;
    const v4i in_rsrc = make_rsrc_words(p.in, p.in_bytes);              // rows past the tensor are out of range = zeros
    const v4i w_rsrc = make_rsrc_words(p.w_hi, (unsigned)((size_t)p.Cout * p.Kpad * 2));
    const unsigned lds_slab = lds_addr(smem) + wave * 1024, lds_ring = lds_addr(ring) + wave * 1024;
    const int whole = (TR % H == 0) ? 1 : 0;                            // whole images per tile (else H % TR == 0: one image segment)

    // ---- geometry-only lane constants (once per workgroup) -------------------------------------------------------------
    // slab source of this lane per DMA round: byte offset relative to the tile's first row (-1: always zero) and whether the
    // pixel lies in the halo row above / below an image segment (then it exists only if the tile does not touch that border)
    int soff[HALO_MAX_ROUNDS];
    unsigned stop = 0, sbot = 0;
    {
        const int spix = wave * 8 + (lane >> 3), slot = lane & 7;
#pragma unroll
        for (int r = 0; r < HALO_MAX_ROUNDS; ++r) {
            const int sp = r * 32 + spix;
            const int srow = (int)fdiv((unsigned)sp, p.dP), sx = sp - srow * P;
            bool ok = srow < p.h_rows && sx >= 1 && sx <= W;
            int grel;
            if (whole) {
                const int i = (int)fdiv((unsigned)srow, p.dH2), b = srow - i * (H + 2);
                ok = ok && b >= 1 && b <= H;
                grel = i * H + b - 1;
            } else {
                grel = srow - 1;
                if (ok && grel < 0) stop |= 1u << r;
                if (ok && grel >= TR) sbot |= 1u << r;
            }
            soff[r] = ok ? ((grel * W + sx - 1) * p.Cin + ((slot ^ ((sx + p.h_swz * srow) & 7)) << 3)) * 2 : -1;
        }
    }
    // fragment rows of this lane: slab pixel of the centre tap (and the swizzle term of that pixel: additive in row and column)
    int pixc[TM], gswc[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const unsigned pa = wave * WM + a * 16 + fr;
        const int j = (int)fdiv(pa, p.dOW), x = (int)pa - j * W;
        const int i = whole ? (int)fdiv((unsigned)j, p.dOH) : 0;
        pixc[a] = (j + 1 + 2 * i) * P + x + 1;
        gswc[a] = x + 1 + p.h_swz * (j + 1 + 2 * i);
    }
    const unsigned lds_slab0 = lds_addr(smem), lds_ring0 = lds_addr(ring);
    unsigned boff[2];                                                   // weight fragment of this lane inside a ring slot, per k-step
    boff[0] = fr * 128 + (((fq ^ (fr >> 1)) & 7) << 4);
    boff[1] = boff[0] ^ 64;
    int woff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int n = r * 32 + wave * 8 + (lane >> 3);
        woff[r] = (n * p.Kpad + (((lane & 7) ^ ((n >> 1) & 7)) << 3)) * 2;
    }

    auto issue_slab = [&](int g0_, int chunk) {
        int top_ok = 1, bot_ok = 1;
        if (!whole) {
            const int h0_ = g0_ - (int)fdiv((unsigned)g0_, p.dOH) * H;
            top_ok = h0_ > 0;
            bot_ok = h0_ + TR < H;
        }
        const unsigned kill = (top_ok ? 0u : stop) | (bot_ok ? 0u : sbot);
        const int cb = g0_ * W * p.Cin * 2 + chunk * 128;
#pragma unroll
        for (int r = 0; r < HALO_MAX_ROUNDS; ++r)
            if (r < S) dma16_async(in_rsrc, lds_slab + r * 4096, (soff[r] == -1 || ((kill >> r) & 1u)) ? (int)0x80000000 : soff[r] + cb);
    };
    // weight stream cursor: item, (chunk, tap) -> k offset, ring slot; runs NR - 1 units ahead, across tile boundaries
    int w_item = blockIdx.x, w_nb = (w_item % NT) * BN * p.Kpad, w_k = 0, w_tap = 0, w_chunk = 0, w_slot = 0, w_ahead = 0;
    auto issue_w = [&]() {
        const unsigned dst = lds_ring + w_slot * W_BYTES;
        const int kb = (w_nb + w_k) * 2;
        if (!(p.h_abl & 128)) {
        dma16_async(w_rsrc, dst, woff[0] + kb);
        dma16_async(w_rsrc, dst + 4096, woff[1] + kb);
        }
        ++w_ahead;
        if (++w_slot == NR) w_slot = 0;
        w_k += p.Cin;
        if (++w_tap == 9) {
            w_tap = 0;
            if (++w_chunk == nchunks) {
                w_chunk = 0;
                w_item += G;
                if (w_item < items) w_nb = (w_item - (w_item / NT) * NT) * BN * p.Kpad;
            }
            w_k = w_chunk * 64;
        }
    };

    int item = blockIdx.x;
    if (item >= items) return;
    HSTAMP(2);                                                          // geometry constants done
    {
        issue_slab((item / NT) * TR, 0);
#pragma unroll
        for (int d = 0; d < NR - 1; ++d) issue_w();                       // NR - 1 units ahead (a tile has >= 9 units)
    }
    int c_slot = 0;
    // WGREC: BatchNorm column sums of every tile this workgroup computes (all of them belong to output-channel tile blockIdx.x % NT:
    // the grid is a multiple of NT) in ONE record per workgroup, written after the tile loop - a 12 x 224^2 batch has 18,816 row
    // tiles on layer1, whose per-tile records cost bn_finalize 33 us per layer.  (Launches with only a few tiles per workgroup keep
    // per-tile records: carrying the sums across the tile loop cost layer3 / layer4 of the bench shape 4-7 us.)
    f32x4 cs[TN], cq[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) { cs[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; cq[b] = cs[b]; }
    for (; item < items; item += G) {
        const int mtile = item / NT, ntile = item - mtile * NT;
        if (!WGREC) {
#pragma unroll
            for (int b = 0; b < TN; ++b) { cs[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; cq[b] = cs[b]; }
        }
        const int g0 = mtile * TR;
        const int npos = min(TR, NH - g0) * W;
        int pix0[TM], tmul[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const bool ok = wave * WM + a * 16 + fr < npos;
            pix0[a] = ok ? pixc[a] : zero_pix;                            // (the zero pixel holds zeros in every chunk: any swizzle)
            tmul[a] = ok ? 1 : 0;
        }
        f32x4 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int chunk = 0; chunk < nchunks; ++chunk) {
            if (chunk > 0) {                                              // refill the (single) slab buffer between two chunks of a tile
                __builtin_amdgcn_s_barrier();                             // every wave is done reading the previous chunk
                asm volatile("" ::: "memory");
                issue_slab(g0, chunk);
            }
            HSTAMP(3);                                                    // slab issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the slab is the youngest DMA: everything has landed
            HSTAMP(4);                                                    // slab landed
            // (rolled tap loop: unrolled, the nine taps' fragment addresses were hoisted and spilled to scratch - whose reloads
            //  are VMEM operations that wait for every LDS-DMA in flight)
            int shift = p.transposed ? (P + 1) : -(P + 1), kx = 0;        // tap (ky, kx) reads the slab shifted by (ky - 1) * P + (kx - 1); negated for the data gradient
            int gshift = p.transposed ? (p.h_swz + 1) : -(p.h_swz + 1);   // ... and its pixel's swizzle term moves by (kx - 1) + h_swz * (ky - 1)
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                // this unit's weights were issued NR - 1 units ago; the NR - 2 units issued after them may still be in flight
                --w_ahead;
                if (tap > 0 && !(p.h_abl & 64)) {
                    if (w_ahead == NR - 2) wait_vmcnt_c<2 * (NR - 2)>();
                    else wait_vmcnt(2 * w_ahead);                         // the last units of the stream
                }
                HSTAMP(5);                                                // this unit's weights landed
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                HSTAMP(6);                                                // past the barrier
                if (w_item < items) issue_w();
                HSTAMP(7);                                                // next unit's weights issued
                if (!(p.h_abl & 1)) {
                    // all twelve fragment reads of the unit are issued before its first MFMA (inline asm: left to itself the compiler
                    // re-uses four registers per operand and waits for the LDS before every MFMA pair - eight exposed LDS latencies
                    // per unit, which at one to three waves per SIMD is most of the unit's time)
                    const unsigned wb = lds_ring0 + c_slot * W_BYTES;
                    unsigned aaddr[TM];
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const int pix = pix0[a] + tmul[a] * shift;
                        aaddr[a] = lds_slab0 + pix * 128 + ((fq ^ ((gswc[a] + gshift) & 7)) << 4);
                    }
                    v4i ar[2][TM], br[2][TN];
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                        for (int a = 0; a < TM; ++a) ar[kk][a] = lds_read16(aaddr[a] ^ (kk << 6));
#pragma unroll
                        for (int b = 0; b < TN; ++b) br[kk][b] = lds_read16_off<0>(wb + boff[kk] + b * 2048);
                    }
                    frag_wait<TM + TN>(ar[0], br[0]);                                  // the first k-step's fragments have landed
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int a = 0; a < TM; ++a) acc[a][b] = MM::mma(__builtin_bit_cast(v8, br[0][b]), __builtin_bit_cast(v8, ar[0][a]), acc[a][b]);
                    frag_wait<0>(ar[1], br[1]);
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int a = 0; a < TM; ++a) acc[a][b] = MM::mma(__builtin_bit_cast(v8, br[1][b]), __builtin_bit_cast(v8, ar[1][a]), acc[a][b]);
                }
                HSTAMP(8);                                                // MFMAs issued
                if (++c_slot == NR) c_slot = 0;
                const int step = (++kx == 3) ? (kx = 0, P - 2) : 1;       // next tap: one pixel right, or down a row and two left
                const int gstep = kx == 0 ? p.h_swz - 2 : 1;
                shift += p.transposed ? -step : step;
                gshift += p.transposed ? -gstep : gstep;
            }
        }
        // the next tile's slab is fetched under this tile's epilogue
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (item + G < items) issue_slab(((item + G) / NT) * TR, 0);

        HSTAMP(9);                                                        // epilogue starts
        // ---- lean epilogue (conv + BatchNorm statistics, or accumulate for the data gradient): rows = positions [g0 W, g0 W + npos)
        if (!(p.h_abl & 4)) {
            AT* const out = (AT*)p.out + ((size_t)g0 * W) * p.Cout + ntile * BN;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int m = wave * WM + a * 16 + fr;
                if (m < npos) {
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        AT* o = out + (size_t)m * p.Cout + b * 16 + fq * 4;
                        f32x4 v = acc[a][b];
                        if (p.accumulate) { float4 e = Act<AT>::ld4(o); v[0] += e.x; v[1] += e.y; v[2] += e.z; v[3] += e.w; }
                        Act<AT>::st4(o, make_float4(v[0], v[1], v[2], v[3]));
                        if (p.stats) {                                    // statistics of what BatchNorm will read back
                            v[0] = Act<AT>::rnd(v[0]); v[1] = Act<AT>::rnd(v[1]); v[2] = Act<AT>::rnd(v[2]); v[3] = Act<AT>::rnd(v[3]);
                            cs[b] += v;
                            cq[b] += v * v;
                        }
                    }
                }
            }
            HSTAMP(10);                                                   // output stores issued
            if (!WGREC && p.stats && !(p.h_abl & 16)) halo_store_stats<TN, BN>(cs, cq, red, p.stats, p.Cout, mtile, ntile, wave, fr, fq, t, true);
        }
        HSTAMP(11);                                                       // tile done
    }
#ifdef HALO_STAMPS
    if (wave == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 255; i += 64) p.h_dbg[(size_t)blockIdx.x * 256 + i] = i < n_stamp ? stl[i] : 0;
        if (lane == 0) p.h_dbg[(size_t)blockIdx.x * 256 + 255] = n_stamp;
    }
#endif
    if (WGREC && p.stats && !(p.h_abl & 16)) halo_store_stats<TN, BN>(cs, cq, red, p.stats, p.Cout, blockIdx.x / NT, blockIdx.x % NT, wave, fr, fq, t, false);
}


// ================================================================================================ halo kernel, row-unit pipeline
// Same tiles, slab and operand layouts as conv_halo2d_kernel, restructured around what its in-kernel stamps showed
// (tools/probes/halo_probe.hip): per 8 KiB tap unit a wave spent ~130 cycles waiting for the weights, ~220 issuing the next two
// DMA pieces and ~540 in the block of 16 MFMAs (256 cycles of matrix work): the compiler re-used four registers per operand and
// waited for the LDS before every MFMA pair, the tap's fragment addresses were recomputed in front of the MFMAs, and the epilogue
// spent ~4k cycles per tile in register copies around its branches.  Here
//   * one ring slot is a kernel ROW of one 64-channel chunk (3 taps x 64 x 64 = 24 KiB): one barrier and one DMA wait per 48 MFMAs;
//   * the fragments of k-step t + 2 are read (inline asm, three register sets) between the MFMAs of k-step t - across kernel rows,
//     chunks and tiles: the stream of kernel rows of a workgroup is one software pipeline, the epilogue of a tile runs with the
//     next tile's first fragments already in registers; every fragment address is a per-workgroup constant (no VALU in the k-steps);
//   * the slab of the next (tile, chunk) is issued in the k-steps right after the barrier that freed its buffer (two slab buffers);
//   * weights, three ways (NTL):
//       NTL = 0  streamed: the row after the next one's six pieces are issued four k-steps before the barrier that needs them
//                (launches with about one tile per workgroup: layer4 of the bench shape, 27 against 45 us);
//       NTL >= 1 resident: the ring holds the three kernel rows of ONE 64-channel chunk while the workgroup runs that chunk for a
//                group of up to NTL of its tiles, each with its own accumulators - a workgroup fetches every weight once per
//                group instead of once per tile (a DMA piece costs the issuing wave ~115 cycles, and with one workgroup per CU
//                nothing else runs meanwhile); with 64 input channels the ring IS the filter bank and is loaded once;
//   * one workgroup per CU (<= 160 KiB of LDS), BatchNorm sums carried in registers: one record per workgroup.
#define HROWS_SLOT (3 * 8192)
// DRIP: the tile's output waits in a 4 KiB LDS tile per wave and leaves as one full-row store per k-step of the NEXT element's first
// kernel row: with every workgroup of the launch in step, stores issued in the epilogue all hit the memory system at once.  Launches
// whose slabs leave no room for the 16 KiB (DRIP = false) store from the epilogue through a 2 KiB tile per wave.
// TMT: 16-position fragments per wave (2: 128-position tiles; 3: 192-position tiles, streamed weights only, no drip - for launches
// whose 128-position tiles need one more round of workgroups than the 192-position ones: layer3 of the bench shape, 384 tiles
// on 256 CUs = two rounds, 256 tiles of 192 = one round of 1.5 x the MFMAs per barrier / DMA piece: 25.9 -> see DESIGN.md)
// PROD (round 4, streamed 128-position tiles only): 512 threads - waves 4..7 are PRODUCERS that issue every LDS-DMA piece of the
// row schedule and wait for it (vmcnt) in front of the row barrier; waves 0..3 run the k-steps, fragment reads and the epilogue and never
// touch the vector-memory path inside the loop (their part of the kernel must fit 256 registers: two waves per SIMD).
template <typename AT, bool DRIP, bool ACCUM, int NTL, int TMT = 2, bool PROD = false>
__global__ __launch_bounds__(PROD ? 512 : 256, 1) void conv_halo_rows_kernel(const ConvArgs p) {
    typedef Mma<typename OpOf<AT>::E> MM;
    typedef typename MM::v8 v8;
    constexpr int BN = 64, TN = 4, TM = TMT, WM = 16 * TMT, TAPB = BN * 128, SLOT = HROWS_SLOT;
    static_assert(TMT == 2 || (TMT == 3 && !DRIP && NTL == 0), "192-position tiles: streamed weights, epilogue stores");
    static_assert(!PROD || (TMT == 2 && NTL == 0), "producer waves: streamed 128-position tiles");
    constexpr int NTLE = NTL < 1 ? 1 : NTL;                             // tiles per group
    constexpr bool STREAM = NTL == 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int slab_bytes = p.h_slab_bytes;
    char* const ring = smem + 2 * slab_bytes;
    float* const red = (float*)(ring + 3 * SLOT);

    const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;      // (producers: wave = the 1 KiB quarter of every piece they move)
    const bool producer = PROD && threadIdx.x >= 256;
    const int fr = lane & 15, fq = lane >> 4;
#ifdef HALO_STAMPS
    int n_stamp = 0;
    long long* const stl = (long long*)(smem + 2 * p.h_slab_bytes + 3 * HROWS_SLOT + (DRIP ? 16384 : 8192));
#endif
    HSTAMP(1);
    const int H = p.IH, W = p.IW, P = W + 2, TR = p.h_tr, NH = p.B * H;
    const int NT = p.Cout / BN, nchunks = p.Cin >> 6;
    const int items = p.h_mtiles * NT, G = gridDim.x;
    const int S = slab_bytes >> 12;
    const int zero_pix = p.h_rows * P;
    const v4i in_rsrc = make_rsrc_words(p.in, p.in_bytes);
    const v4i w_rsrc = make_rsrc_words(p.w_hi, (unsigned)((size_t)p.Cout * p.Kpad * 2));
    const unsigned lds_slab0 = lds_addr(smem), lds_ring0 = lds_addr(ring);
    const unsigned dma_slab = lds_slab0 + wave * 1024, dma_ring = lds_ring0 + wave * 1024;
    const int whole = (TR % H == 0) ? 1 : 0;
    // XCD-aware tile order (round 6, launches with one tile per workgroup and h_xcg set by the host): hardware workgroup b runs on XCD b % 8,
    // and in linear order the NT channel tiles of a row tile land on NT different XCDs - every XCD streamed a quarter of layer3's rows through
    // its L2 for ONE channel tile (PMC: 27.9 MB fetched for 7.5 MB of operands), every XCD all of layer4's rows.  Here XCD x owns a block of
    // (h_xcg channel tiles) x (row tiles): cg = NT for layer3 (rows fetched once, the 1.2 MB of weights by every XCD), 2 for layer4.
    int bid = blockIdx.x;
    if (p.h_xcg) {
        const int cg = p.h_xcg, x = bid & 7, s = bid >> 3, ngrp = NT / cg;
        const int cgi = x % ngrp, rgi = x / ngrp;
        bid = (rgi * ((G >> 3) / cg) + s / cg) * NT + cgi * cg + s % cg;
    }
    if (bid >= items) return;
    const int n_my = (items - bid + G - 1) / G;             // tiles of this workgroup: items bid + k G (one output-channel tile)
    const bool w_once = !STREAM && nchunks == 1;                        // resident filter bank

    int woff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int n = r * 32 + wave * 8 + (lane >> 3);
        woff[r] = (n * p.Kpad + (((lane & 7) ^ ((n >> 1) & 7)) << 3)) * 2;
    }
    // ---- weights: kernel row ky of a chunk goes to ring slot ky; six 1 KiB pieces per wave and row
    const int w_nb = (bid % NT) * BN * p.Kpad;
    auto w_row_pieces = [&](int chunk, int ky, int j0, int j1) {
        // the data gradient is the same correlation with the taps taken in reverse order (tap' = 8 - tap)
        const int tap0 = p.transposed ? 8 - ky * 3 : ky * 3, tstep = p.transposed ? -1 : 1;
        const int kb = (w_nb + chunk * 64) * 2;
        const unsigned dst = dma_ring + ky * SLOT;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j >= j0 && j < j1)
                dma16_async(w_rsrc, dst + (j >> 1) * TAPB + (j & 1) * 4096, woff[j & 1] + kb + (tap0 + tstep * (j >> 1)) * p.Cin * 2);
    };
    // streamed mode: cursor over (tile, chunk) elements, three rows each
    int w_k = 0, w_chunk = 0;                                           // tile index in this workgroup's list, chunk
    auto p_w_pieces = [&](int ky, int j0, int j1) {                      // (the issuing side: every wave, or the producer waves)
#ifndef HR_ABL_NODMA
        if (STREAM && w_k < n_my) w_row_pieces(w_chunk, ky, j0, j1);
#endif
    };
    auto w_pieces = [&](int ky, int j0, int j1) {                        // (as called from the k-step schedule)
        if constexpr (!PROD) p_w_pieces(ky, j0, j1);
    };
    auto p_w_advance = [&]() {
        if (++w_chunk == nchunks) { w_chunk = 0; ++w_k; }
    };
    auto w_advance = [&]() {
        if constexpr (!PROD) p_w_advance();
    };
    // issued first: the weight DMAs are in flight while the slab source table and the fragment addresses are computed
    if (!PROD || producer) {
        w_row_pieces(0, 0, 0, 6);
        if (STREAM) {
            w_row_pieces(0, 1, 0, PROD ? 6 : 4);
        } else {
            w_row_pieces(0, 1, 0, 6);
            w_row_pieces(0, 2, 0, 6);
        }
    }

    // L2 warm-up (round 6): in the training step every launch finds its weights cold (packed at the start of the step, megabytes of
    // activations ago) and the ring runs one kernel row ahead - far less than a trip to HBM.  The workgroups of an XCD share the weight rows
    // of the XCD's channel tiles: each touches its 1 / n of them (one dword per 128-byte line and lane, right behind the first pieces), so
    // the XCD's L2 holds the whole slice by the time the prologue (~3 us) is over.  Waited for with the first pieces (vmcnt(0) below).
    // The touches are LDS-DMA loads into a dummy kilobyte per wave of `red` (unused until the epilogue): a load with a register destination
    // that nothing waits for would leave the compiler free to hand that register to another value before the data lands in it.
    int touch4 = 0;
    if (p.h_xcg > 0 && p.h_touch && (!PROD || producer)) {
        const unsigned dummy = lds_addr(red) + wave * 1024;
        const int cg = p.h_xcg, n = G >> 3, s = (int)blockIdx.x >> 3;
        const int lines = cg * BN * p.Kpad / 64, per = (lines + n - 1) / n;      // 128-byte lines of the XCD's cg channel tiles; this workgroup's share
        const int l0 = (bid % NT / cg * cg) * BN * p.Kpad / 64 + s * per;
        // (lanes past the share read out of range: zeros, no traffic; the branch is uniform per wave only up to the share's last wave)
        if (wave * 64 < per) dma16_async(w_rsrc, dummy, t < per ? (l0 + t) * 128 : (int)0x80000000);
        if (wave * 64 + 256 < per) dma16_async(w_rsrc, dummy, t + 256 < per ? (l0 + t + 256) * 128 : (int)0x80000000);
        // ... and the tile's input rows (written by the previous kernel of the chain: in another XCD's L2 or in memory by now), split over
        // the cg workgroups of this XCD that share the row tile; the accumulating data gradient also warms the output tile its epilogue
        // reads (those loads sit at the very end of the kernel, with nothing left to hide them)
        const int g0t = (bid / NT) * TR;
        const int ilines = min(TR, NH - g0t) * W * p.Cin / 64, ishare = (ilines + cg - 1) / cg, il0 = (bid % NT % cg) * ishare;
        const int ibase = g0t * W * p.Cin * 2;
        if (wave * 64 < ishare) dma16_async(in_rsrc, dummy, (t < ishare && il0 + t < ilines) ? ibase + (il0 + t) * 128 : (int)0x80000000);
        if (wave * 64 + 256 < ishare)
            dma16_async(in_rsrc, dummy, (t + 256 < ishare && il0 + t + 256 < ilines) ? ibase + (il0 + t + 256) * 128 : (int)0x80000000);
        if (ACCUM && t < min(TR, NH - g0t) * W) touch4 = *(const volatile int*)((const char*)p.out + ((size_t)(g0t * W + t) * p.Cout + (bid % NT) * BN) * sizeof(AT));
    }

    // ---- geometry-only lane constants (as conv_halo2d_kernel) ----------------------------------------------------------
    int soff[HALO_MAX_ROUNDS];
    unsigned stop = 0, sbot = 0;
    {
        const int spix = wave * 8 + (lane >> 3), slot = lane & 7;
#pragma unroll
        for (int r = 0; r < HALO_MAX_ROUNDS; ++r) {
            soff[r] = -1;
            if (r >= S) continue;                                       // (uniform: rounds past the slab are never issued)
            const int sp = r * 32 + spix;
            const int srow = (int)fdiv((unsigned)sp, p.dP), sx = sp - srow * P;
            bool ok = srow < p.h_rows && sx >= 1 && sx <= W;
            int grel;
            if (whole) {
                const int i = (int)fdiv((unsigned)srow, p.dH2), b = srow - i * (H + 2);
                ok = ok && b >= 1 && b <= H;
                grel = i * H + b - 1;
            } else {
                grel = srow - 1;
                if (ok && grel < 0) stop |= 1u << r;
                if (ok && grel >= TR) sbot |= 1u << r;
            }
            soff[r] = ok ? ((grel * W + sx - 1) * p.Cin + ((slot ^ ((sx + p.h_swz * srow) & 7)) << 3)) * 2 : -1;
        }
    }
    // ---- slab stream: one element per (group, chunk, tile of the group), alternating between the two slab buffers -------------
    int s_g = 0, s_chunk = 0, s_tl = 0, s_buf = 0, s_cb = 0;
    unsigned s_kill = 0;
    auto grp_n = [&](int g) { return min(NTLE, n_my - g); };
    auto slab_setup = [&]() {                                           // source constants of the element the cursor points at
        const int item_ = bid + (s_g + s_tl) * G;
        const int g0_ = (item_ / NT) * TR;
        int top_ok = 1, bot_ok = 1;
        if (!whole) {
            const int h0_ = g0_ - (int)fdiv((unsigned)g0_, p.dOH) * H;
            top_ok = h0_ > 0;
            bot_ok = h0_ + TR < H;
        }
        s_kill = (top_ok ? 0u : stop) | (bot_ok ? 0u : sbot);
        s_cb = g0_ * W * p.Cin * 2 + s_chunk * 128;
    };
    bool slab_on = true;                                                // (probe builds switch the in-loop slab DMAs off)
    auto p_slab_pieces = [&](int r0, int r1) {
        if (s_g < n_my && slab_on) {
            const unsigned dst = dma_slab + s_buf * slab_bytes;
#pragma unroll
            for (int r = 0; r < HALO_MAX_ROUNDS; ++r)
                if (r >= r0 && r < r1 && r < S)
                    dma16_async(in_rsrc, dst + r * 4096, (soff[r] == -1 || ((s_kill >> r) & 1u)) ? (int)0x80000000 : soff[r] + s_cb);
        }
    };
    auto p_slab_advance = [&]() {
        s_buf ^= 1;
        if (++s_tl == grp_n(s_g)) {
            s_tl = 0;
            if (++s_chunk == nchunks) { s_chunk = 0; s_g += NTLE; }
        }
        if (s_g < n_my) slab_setup();
    };
    auto slab_pieces = [&](int r0, int r1) {
        if constexpr (!PROD) p_slab_pieces(r0, r1);
    };
    auto slab_advance = [&]() {
        if constexpr (!PROD) p_slab_advance();
    };
    if (!PROD || producer) {
        slab_setup();                                                   // the first slab, before the remaining constants
        p_slab_pieces(0, HALO_MAX_ROUNDS);
        p_slab_advance();
    }
#ifdef HR_ABL_NODMA
    slab_on = false;
#endif
    if (PROD && producer) {
        // ---- producer waves: the DMA side of the row schedule below, one barrier per kernel row -----------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(touch4));
        __builtin_amdgcn_s_barrier();                                   // (the consumers' prologue barrier)
        asm volatile("" ::: "memory");
        int c_g = 0, c_chunk = 0;
#pragma unroll 1
        for (;;) {
#pragma unroll
            for (int KY = 0; KY < 3; ++KY) {
                // Everything this barrier promises the consumers was issued a whole kernel row ago (the consumers' own schedule issued
                // the last two pieces of a row in the k-step before its barrier, to spread the issue cost over their MFMAs): all six
                // pieces of the row after the next go out right after the barrier that frees their ring slot, the next element's whole
                // slab after the barrier that frees its buffer.
                const int PKY = (KY + 2) % 3;
                if (KY == 1) p_w_advance();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                p_w_pieces(PKY, 0, 6);
                if (KY == 0) { p_slab_pieces(0, HALO_MAX_ROUNDS); p_slab_advance(); }
            }
            int n_g = c_g, n_chunk = c_chunk + 1;
            if (n_chunk == nchunks) { n_chunk = 0; n_g += NTLE; }
            if (n_g >= n_my) break;
            c_g = n_g; c_chunk = n_chunk;
        }
        __syncthreads();                                                // (the consumers' barrier in front of the statistics)
        if (p.stats) __syncthreads();                                   // (halo_store_stats)
        return;
    }

    int pixc[TM], gswc[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const unsigned pa = wave * WM + a * 16 + fr;
        const int j = (int)fdiv(pa, p.dOW), x = (int)pa - j * W;
        const int i = whole ? (int)fdiv((unsigned)j, p.dOH) : 0;
        pixc[a] = (j + 1 + 2 * i) * P + x + 1;
        gswc[a] = x + 1 + p.h_swz * (j + 1 + 2 * i);                    // swizzle term of the centre pixel (additive in row and column)
    }
    // ---- fragment addresses: every tap's slab address of this lane's two fragment rows, both 64-byte halves (buffer 0; the
    //      other slab buffer is + slab_bytes), and the weight fragment bases of the three ring slots - no address arithmetic
    //      is left in the k-steps
    unsigned arel[9][TM][2], bbase[3][2];
    {
        const unsigned boff = fr * 128 + (((fq ^ (fr >> 1)) & 7) << 4);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            bbase[ky][0] = lds_ring0 + ky * SLOT + boff;
            bbase[ky][1] = lds_ring0 + ky * SLOT + (boff ^ 64);
        }
    }
    int cur_npos = -1;
    auto set_rows = [&](int item_) {                                    // rows past the end of the last tile read the zero pixel
        const int npos_ = min(TR, NH - (item_ / NT) * TR) * W;
        if (npos_ == cur_npos) return;
        cur_npos = npos_;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const bool ok = wave * WM + a * 16 + fr < npos_;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int pix = ok ? pixc[a] + (tap / 3 - 1) * P + (tap % 3 - 1) : zero_pix;
                const unsigned rel = (pix << 7) | ((fq ^ ((gswc[a] + (tap % 3 - 1) + p.h_swz * (tap / 3 - 1)) & 7)) << 4);
                arel[tap][a][0] = lds_slab0 + rel;
                arel[tap][a][1] = lds_slab0 + (rel ^ 64);
            }
        }
    };
    set_rows(bid);

    // ---- fragments: three register sets, set of k-step t = t % 3 (six k-steps per kernel row) -----------------------------
    v4i fa[3][TM], fb[3][TN];
    f32x4 acc[NTLE][TM][TN], cs[TN], cq[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        cs[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; cq[b] = cs[b];
#pragma unroll
        for (int q = 0; q < NTLE; ++q)
#pragma unroll
            for (int a = 0; a < TM; ++a) acc[q][a][b] = cs[b];
    }
    int c_g = 0, c_chunk = 0, c_sbuf = 0;
    // drip state: staged tile of the previous epilogue (rows m0 + 8 d, d = 0..3, 16 bytes per lane)
    const unsigned d_lds0 = lds_addr(red) + wave * 4096 + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 4) & 3)) << 4);
    const unsigned d_lds1 = lds_addr(red) + wave * 4096 + (8 + (lane >> 3)) * 128 + (((lane & 7) ^ (4 + ((lane >> 4) & 3))) << 4);
    const int d_m0 = wave * WM + (lane >> 3);
    const size_t d_step = (size_t)8 * p.Cout * sizeof(AT);
    char* d_ptr = nullptr;
    int d_npos = 0;
    bool d_on = false;
    v4i d_q;
#define HR_DRIP_READ(D) if (DRIP && d_on) d_q = ((D) & 1) ? lds_read16_off<((D) >> 1) * 2048>(d_lds1) : lds_read16_off<((D) >> 1) * 2048>(d_lds0);
#define HR_DRIP_STORE(D)                                                                  \
    if (DRIP && d_on) {                                                                   \
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(d_q));                                 \
        if (d_m0 + (D) * 8 < d_npos) *(v4i*)(d_ptr + (D) * d_step) = d_q;                  \
    }

#ifdef HR_ABL_NOMMA                                                     /* probe builds only (tools/probes/halo_probe.hip): timing ablations */
#define HR_MMA(TL, SET, A, B) asm volatile("" : "+v"(acc[TL][A][B]) : "v"(fb[SET][B]), "v"(fa[SET][A]))
#else
#define HR_MMA(TL, SET, A, B) acc[TL][A][B] = MM::mma(__builtin_bit_cast(v8, fb[SET][B]), __builtin_bit_cast(v8, fa[SET][A]), acc[TL][A][B])
#endif
#ifdef HR_ABL_NOREAD
#define HR_RD(dst, expr) asm volatile("" : "+v"(dst))
#else
#define HR_RD(dst, expr) dst = expr
#endif
    // k-step: wait for set CUR, then its eight MFMAs with the six reads of set NX between them (the fragments of the k-step after
    // the next: tap TAP, half KK of the 64 channels, slab buffer offset SB, ring slot KYR)
#define HR_KSTEP(TL, CUR, NX, SB, TAP, KYR, KK)                                                                       \
    {                                                                                                                  \
        frag_wait<TM + TN>(fa[CUR], fb[CUR]);                                                                          \
        if constexpr (TM == 2) {                                                                                       \
        HR_MMA(TL, CUR, 0, 0); HR_RD(fa[NX][0], lds_read16(arel[TAP][0][KK] + (SB)));                                    \
        HR_MMA(TL, CUR, 1, 0); HR_RD(fa[NX][1], lds_read16(arel[TAP][1][KK] + (SB)));                                    \
        HR_MMA(TL, CUR, 0, 1); HR_RD(fb[NX][0], lds_read16_off<((TAP) % 3) * TAPB>(bbase[KYR][KK]));                     \
        HR_MMA(TL, CUR, 1, 1); HR_RD(fb[NX][1], lds_read16_off<((TAP) % 3) * TAPB + 2048>(bbase[KYR][KK]));              \
        HR_MMA(TL, CUR, 0, 2); HR_RD(fb[NX][2], lds_read16_off<((TAP) % 3) * TAPB + 4096>(bbase[KYR][KK]));              \
        HR_MMA(TL, CUR, 1, 2); HR_RD(fb[NX][3], lds_read16_off<((TAP) % 3) * TAPB + 6144>(bbase[KYR][KK]));              \
        HR_MMA(TL, CUR, 0, 3);                                                                                          \
        HR_MMA(TL, CUR, 1, 3);                                                                                          \
        } else {                                  /* twelve MFMAs, seven reads */                                       \
        HR_MMA(TL, CUR, 0, 0); HR_RD(fa[NX][0], lds_read16(arel[TAP][0][KK] + (SB)));                                    \
        HR_MMA(TL, CUR, 1, 0); HR_RD(fa[NX][1], lds_read16(arel[TAP][1][KK] + (SB)));                                    \
        HR_MMA(TL, CUR, TM - 1, 0); HR_RD(fa[NX][TM - 1], lds_read16(arel[TAP][TM - 1][KK] + (SB)));                     \
        HR_MMA(TL, CUR, 0, 1); HR_RD(fb[NX][0], lds_read16_off<((TAP) % 3) * TAPB>(bbase[KYR][KK]));                     \
        HR_MMA(TL, CUR, 1, 1); HR_RD(fb[NX][1], lds_read16_off<((TAP) % 3) * TAPB + 2048>(bbase[KYR][KK]));              \
        HR_MMA(TL, CUR, TM - 1, 1); HR_RD(fb[NX][2], lds_read16_off<((TAP) % 3) * TAPB + 4096>(bbase[KYR][KK]));         \
        HR_MMA(TL, CUR, 0, 2); HR_RD(fb[NX][3], lds_read16_off<((TAP) % 3) * TAPB + 6144>(bbase[KYR][KK]));              \
        HR_MMA(TL, CUR, 1, 2);                                                                                          \
        HR_MMA(TL, CUR, TM - 1, 2);                                                                                     \
        HR_MMA(TL, CUR, 0, 3);                                                                                          \
        HR_MMA(TL, CUR, 1, 3);                                                                                          \
        HR_MMA(TL, CUR, TM - 1, 3);                                                                                     \
        }                                                                                                              \
    }
    // kernel row KY of the current element: six k-steps; NKY = (KY + 1) % 3, SBN = slab buffer of the row after this one
#define HR_ROW(TL, KY, NKY, SBN)                                                                                      \
    {                                                                                                                  \
        if ((KY) == 0) { HR_DRIP_READ(0) }                                                                              \
        HR_KSTEP(TL, 0, 2, sb, (KY) * 3 + 1, KY, 0)                                                                     \
        if ((KY) == 0) { HR_DRIP_STORE(0) }                                                                             \
        w_pieces(NKY, 4, 6);                          /* streamed: the last two pieces of the next kernel row */        \
        if ((KY) == 1) { if (STREAM) w_advance(); slab_pieces(7, HALO_MAX_ROUNDS); slab_advance(); }                    \
        if ((KY) == 0) { HR_DRIP_READ(1) }                                                                              \
        HR_KSTEP(TL, 1, 0, sb, (KY) * 3 + 1, KY, 1)                                                                     \
        if ((KY) == 0) { HR_DRIP_STORE(1) HR_DRIP_READ(2) }                                                             \
        HR_KSTEP(TL, 2, 1, sb, (KY) * 3 + 2, KY, 0)                                                                     \
        if ((KY) == 0) { HR_DRIP_STORE(2) HR_DRIP_READ(3) }                                                             \
        HR_KSTEP(TL, 0, 2, sb, (KY) * 3 + 2, KY, 1)                                                                     \
        if ((KY) == 0) { HR_DRIP_STORE(3) d_on = false; }                                                               \
        if ((KY) == 2) set_rows(n_item);                                                                               \
        HSTAMP(4);                                                                                                     \
        /* the next row's weights (streamed) and, before a new element, its slab have landed for every wave; the row before this is free */ \
        if constexpr (!PROD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
        HSTAMP(5);                                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
        HSTAMP(6);                                                                                                     \
        HR_KSTEP(TL, 1, 0, SBN, (NKY) * 3, NKY, 0)                                                                      \
        w_pieces(((KY) + 2) % 3, 0, 2);                                                                                 \
        if ((KY) == 0) slab_pieces(0, 4);                                                                               \
        HR_KSTEP(TL, 2, 1, SBN, (NKY) * 3, NKY, 1)                                                                      \
        w_pieces(((KY) + 2) % 3, 2, 4);                                                                                 \
        if ((KY) == 0) slab_pieces(4, 7);                                                                               \
        HSTAMP(8);                                                                                                     \
    }
    // the first fragments of an element (k-steps 0 and 1 of its kernel row 0) from slab buffer offset SB
#define HR_FIRST_READS(SB)                                                                                            \
    {                                                                                                                  \
        /* in k-step order: the first k-step waits for all but the TM + TN youngest reads */                           \
        fa[0][0] = lds_read16(arel[0][0][0] + (SB)); fa[0][1] = lds_read16(arel[0][1][0] + (SB));                        \
        if constexpr (TM == 3) fa[0][TM - 1] = lds_read16(arel[0][TM - 1][0] + (SB));                                    \
        fb[0][0] = lds_read16_off<0>(bbase[0][0]); fb[0][1] = lds_read16_off<2048>(bbase[0][0]);                        \
        fb[0][2] = lds_read16_off<4096>(bbase[0][0]); fb[0][3] = lds_read16_off<6144>(bbase[0][0]);                     \
        fa[1][0] = lds_read16(arel[0][0][1] + (SB)); fa[1][1] = lds_read16(arel[0][1][1] + (SB));                        \
        if constexpr (TM == 3) fa[1][TM - 1] = lds_read16(arel[0][TM - 1][1] + (SB));                                    \
        fb[1][0] = lds_read16_off<0>(bbase[0][1]); fb[1][1] = lds_read16_off<2048>(bbase[0][1]);                        \
        fb[1][2] = lds_read16_off<4096>(bbase[0][1]); fb[1][3] = lds_read16_off<6144>(bbase[0][1]);                     \
    }
    // element TL of the group: three kernel rows, then (after the last chunk) the tile's epilogue
#define HR_ELEM(TL)                                                                                                   \
    if ((TL) < NTLE && (TL) < ng) {                                                                                    \
        const int c_item = bid + (c_g + (TL)) * G;                                                         \
        /* the element after this one (its first fragments are read in kernel row 2): next tile of the group, else next (chunk, group) */ \
        int n_item = c_item + G;                                                                                       \
        if ((TL) + 1 >= ng) n_item = bid + (last_chunk ? c_g + NTLE : c_g) * G;                            \
        if (n_item >= items) n_item = c_item;                                                                          \
        const unsigned sb = c_sbuf * slab_bytes, sbn = (c_sbuf ^ 1) * slab_bytes;                                      \
        HR_ROW(TL, 0, 1, sb)                                                                                            \
        HR_ROW(TL, 1, 2, sb)                                                                                            \
        HR_ROW(TL, 2, 0, sbn)                                                                                           \
        c_sbuf ^= 1;                                                                                                   \
        if (last_chunk) {                                                                                              \
            epilogue(acc[TL], c_item);                                                                                 \
            HSTAMP(10);                                                                                                \
        }                                                                                                              \
    }

    auto epilogue = [&](f32x4 (&ac)[TM][TN], int c_item) __attribute__((always_inline)) {
        // ---- epilogue of the tile (the next element's first fragments are already in flight / in registers)
        const int mtile = c_item / NT, ntile = c_item - mtile * NT;
        const int g0 = mtile * TR;
        const int npos = min(TR, NH - g0) * W;
        if (!(p.h_abl & 4)) {
            AT* const out = (AT*)p.out + ((size_t)g0 * W) * p.Cout + ntile * BN;
            if (ACCUM) {                                                  // data gradient added to the shortcut's: fp32 sum, rounded once
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int m = wave * WM + a * 16 + fr;
                    if (m < npos) {
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            AT* o = out + (size_t)m * p.Cout + b * 16 + fq * 4;
                            f32x4 v = ac[a][b];
                            float4 e = Act<AT>::ld4(o);
                            v[0] += e.x; v[1] += e.y; v[2] += e.z; v[3] += e.w;
                            Act<AT>::st4(o, make_float4(v[0], v[1], v[2], v[3]));
                        }
                    }
                }
            } else {
                // the accumulators hold 4 channels x 16 positions per register quad: stored directly that is eight 8-byte stores per
                // lane in 32-byte row segments.  Staged through LDS (16 positions x 128 B per fragment row block, 16-byte chunks
                // XOR-swizzled by position pair) every store instruction writes eight full 128-byte rows.
                char* const stg = (char*)red + wave * (DRIP ? 4096 : 2048);   // (`red` itself is only used after the last tile)
#pragma unroll
                for (int a = 0; a < TM; ++a) {
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const f32x4 v = ac[a][b];
                        AT h4[4] = {(AT)v[0], (AT)v[1], (AT)v[2], (AT)v[3]};
                        *(uint2*)(stg + (DRIP ? a * 2048 : 0) + fr * 128 + (((b * 2 + (fq >> 1)) ^ ((fr >> 1) & 7)) << 4) + (fq & 1) * 8) =
                            *(const uint2*)h4;
                        {   // BatchNorm sums of what will be read back (unconditional: a branch here makes the compiler copy the
                            // sum registers at every merge; a launch without statistics just does not write them)
                            f32x4 r = {(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
                            if (wave * WM + a * 16 + fr >= npos) r = (f32x4){0.f, 0.f, 0.f, 0.f};
                            cs[b] += r;
                            cq[b] += r * r;
                        }
                    }
                    if (!DRIP) {
#pragma unroll
                        for (int i2 = 0; i2 < 2; ++i2) {
                            const int pos = i2 * 8 + (lane >> 3), ch = lane & 7;
                            const uint4 q = *(const uint4*)(stg + pos * 128 + ((ch ^ ((pos >> 1) & 7)) << 4));
                            const int m = wave * WM + a * 16 + pos;
                            if (m < npos) *(uint4*)((char*)(out + (size_t)m * p.Cout) + ch * 16) = q;
                        }
                    }
                }
                if (DRIP) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staged (the asm reads of the drip do not wait for plain stores)
                    d_on = true;
                    d_npos = npos;
                    d_ptr = (char*)(out + (size_t)d_m0 * p.Cout) + (lane & 7) * 16;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) ac[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    HSTAMP(2);
    // ---- prologue: the first slab and weights have been issued; fragments of k-steps 0 and 1 --------------------------------
    if constexpr (!PROD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!PROD) asm volatile("" :: "v"(touch4));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    HR_FIRST_READS(0u)
    HSTAMP(3);

#pragma unroll 1
    for (;;) {                                                          // one (group, chunk) per iteration
        const int ng = grp_n(c_g);
        const bool last_chunk = c_chunk == nchunks - 1;
        HR_ELEM(0)
        HR_ELEM(1)
        HR_ELEM(2)
        HR_ELEM(3)
        int n_g = c_g, n_chunk = c_chunk + 1;
        if (n_chunk == nchunks) { n_chunk = 0; n_g += NTLE; }
        if (n_g >= n_my) break;
        if (!STREAM && !w_once) {
            // the ring changes chunk: every wave is done with the old one (its last reads were waited for in the final k-steps; the
            // fragments prefetched for the next element came from the OLD weights and are read again below)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            w_row_pieces(n_chunk, 0, 0, 6);
            w_row_pieces(n_chunk, 1, 0, 6);
            w_row_pieces(n_chunk, 2, 0, 6);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned sb0 = c_sbuf * slab_bytes;
            HR_FIRST_READS(sb0)
        }
        c_g = n_g; c_chunk = n_chunk;
    }
    // drain: the fragment reads issued for a kernel row that does not exist; the last tile's staged output
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (DRIP && d_on) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const v4i q = *(const v4i*)((const char*)red + wave * 4096 + (d >> 1) * 2048 + ((d & 1) * 8 + (lane >> 3)) * 128 +
                                        (((lane & 7) ^ ((((d & 1) * 8 + (lane >> 3)) >> 1) & 7)) << 4));
            if (d_m0 + d * 8 < d_npos) *(v4i*)(d_ptr + d * d_step) = q;
        }
    }
    __syncthreads();                                                    // the staging tiles alias `red`, which the statistics use next
    if (p.stats) halo_store_stats<TN, BN>(cs, cq, red, p.stats, p.Cout, bid / NT, bid % NT, wave, fr, fq, t, false);
    HSTAMP(11);
#ifdef HALO_STAMPS
    if (wave == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 255; i += 64) p.h_dbg[(size_t)blockIdx.x * 256 + i] = i < n_stamp ? stl[i] : 0;
        if (lane == 0) p.h_dbg[(size_t)blockIdx.x * 256 + 255] = n_stamp;
    }
#endif
#undef HR_ELEM
#undef HR_FIRST_READS
#undef HR_ROW
#undef HR_DRIP_READ
#undef HR_DRIP_STORE
#undef HR_KSTEP
#undef HR_MMA
}


// ================================================================================================ stem kernel (2D, 4 stored input channels, stride 2)
// conv1 of the ResNet trunk (7x7 / 2 / pad 3, 3 -> 64 channels, mv_cnn.py:44) through conv_igemm_kernel is a chain of exposed gather
// latencies: 7 k-steps of 8-byte im2col loads per 128-position tile, 57-72 us at the bench shape against ~20 us for its 100 MB
// of output.  With four stored channels one kernel ROW (kw = 0..7, ci = 0..3) is exactly one 32-wide MFMA k-step, so:
//   * the whole filter bank lives in REGISTERS as MFMA A fragments (KH rows x 4 output-channel tiles x 4 VGPRs = 112 for 7x7 -> 64),
//     loaded once per persistent workgroup;
//   * the input rows a tile of TH output rows needs are staged ONCE in an LDS slab (8 B per pixel, borders zero); output column ow,
//     k-group g reads pixels 2 ow - pad + 2 g, + 1 = the 16-byte chunk (ow + g) of slab row 2 r + kh: one conflict-free ds_read_b128 per
//     16 positions and kernel row, no address arithmetic beyond an add;
//   * the next tile's slab rows are prefetched into registers under the MFMAs of the current one (two slab buffers);
//   * the BatchNorm column sums stay in registers across the workgroup's tiles: ONE record per workgroup.
// Geometry (stem_geometry): KD = 1, stored Cin = 4, stride 2, KW <= 8, KH <= 8, Cout = 64, OW % 16 == 0, IW + pad <= 2 OW + 6.
#define STEM_MAX_KH 8
struct StemGeom { int TH, slab_rows, row_bytes, ptiles, tiles_per_img, ntiles, grid, npt; };
// NPT = position tiles (16 positions) per wave and workgroup tile: 4 (256-position tiles, ~490 registers: one workgroup per CU) or
// 2 (128-position tiles, <= 256 registers: two per CU, whose MFMAs overlap each other's epilogue / slab traffic)
template <int KH, int NPT, typename AT>
__global__ __launch_bounds__(256, NPT == 2 ? 2 : 1) void conv_stem_kernel(const ConvArgs p, const StemGeom sg) {
    typedef Mma<typename OpOf<AT>::E> MM;
    typedef typename MM::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
    const int slab_bytes = sg.slab_rows * sg.row_bytes;
    float* const red = (float*)(smem + 2 * slab_bytes);                          // [4 waves][64][2]
    const int OW = p.OW, IW = p.IW, IH = p.IH, OH = p.OH;
    const int pt_per_row = OW >> 4;

    // ---- filter bank -> registers: A fragment (kh, ct): row = output channel 16 ct + fr, k = (kw = 2 fq, 2 fq + 1) x 4 channels
    v8 wf[KH][4];
    {
        const uint16_t* w = (const uint16_t*)p.w_hi;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const uint16_t* src = w + (size_t)(ct * 16 + fr) * p.Kpad + (kh * p.KW + 2 * fq) * 4;
                uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
                if (2 * fq < p.KW) lo = *(const uint2*)src;
                if (2 * fq + 1 < p.KW) hi = *(const uint2*)(src + 4);
                const uint4 raw = make_uint4(lo.x, lo.y, hi.x, hi.y);
                wf[kh][ct] = __builtin_bit_cast(v8, raw);
            }
        // the fragments are complete here; re-define them through an empty asm so that the compiler's waitcnt pass does not carry
        // "pending global loads feed the MFMA operands" into the tile loop, where it became a vmcnt(0) in front of the first MFMA
        // of every tile - i.e. a wait for the NEXT tile's prefetch that is supposed to fly under these MFMAs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) asm volatile("" : "+v"(wf[kh][ct]));
    }
    // zero both slab buffers once: the border chunks are never written again
    for (int i = t * 16; i < 2 * slab_bytes; i += 256 * 16) *(uint4*)(smem + i) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    // slab fill: 16-byte global loads of two pixels (x even) -> LDS at pixel x + pad (8-byte aligned: two ds_write_b64).  Which slab
    // row / pixel pair a thread moves never changes: computed once (the tile loop ran ~1,500 instructions per wave with the divisions
    // inside it - at one wave per SIMD that was 47 of the kernel's 65 us).
    constexpr int MAXLD = 4;                                                     // loads per thread per tile (stem_geometry checks it)
    const int ld_per_row = IW >> 1, nld = sg.slab_rows * ld_per_row;
    int ld_row[MAXLD], ld_goff[MAXLD], ld_loff[MAXLD];
#pragma unroll
    for (int u = 0; u < MAXLD; ++u) {
        const int e = t + u * 256;
        const int srow = e / ld_per_row, xp = e - srow * ld_per_row;
        ld_row[u] = e < nld ? srow : -(1 << 20);                                 // never a valid image row
        ld_goff[u] = (srow * IW + 2 * xp) * 8;
        ld_loff[u] = srow * sg.row_bytes + (2 * xp + p.pw) * 8;
    }
    uint4 pre[MAXLD];
    auto fetch = [&](int tile) {
        const int img = tile / sg.tiles_per_img, oh0 = (tile - img * sg.tiles_per_img) * sg.TH;
        const int iy0 = oh0 * 2 - p.ph;
        const char* base = (const char*)p.in + ((size_t)img * IH + iy0) * IW * 8;          // may point before the image: only valid rows are read
#pragma unroll
        for (int u = 0; u < MAXLD; ++u) {
            pre[u] = make_uint4(0u, 0u, 0u, 0u);
            if ((unsigned)(iy0 + ld_row[u]) < (unsigned)IH) pre[u] = *(const uint4*)(base + ld_goff[u]);
        }
    };
    auto stash = [&](int buf) {
        char* sb = smem + buf * slab_bytes;
#pragma unroll
        for (int u = 0; u < MAXLD; ++u)
            if (ld_row[u] >= 0) {
                char* d = sb + ld_loff[u];
                *(uint2*)d = make_uint2(pre[u].x, pre[u].y);
                *(uint2*)(d + 8) = make_uint2(pre[u].z, pre[u].w);
            }
    };

    f32x4 cs[4], cq[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { cs[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; cq[ct] = cs[ct]; }
    // position tiles of this wave inside a workgroup tile: q = wave, wave + 4, ... (row q / pt_per_row, columns 16 (q % pt_per_row) ..);
    // a tile index past the end recomputes tile 0 (branch-free MFMA loop), only its stores / statistics are skipped
    int boff[NPT], ooff[NPT], orow[NPT];
    bool inside[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int q = wave + 4 * j;
        const int qc = q < sg.ptiles ? q : 0;
        const int r = qc / pt_per_row, c0 = (qc - r * pt_per_row) * 16;
        inside[j] = q < sg.ptiles;
        orow[j] = r;
        boff[j] = (2 * r) * sg.row_bytes + (c0 + fr + fq) * 16;
        ooff[j] = ((r * OW) + c0 + fr) * p.Cout + fq * 4;
    }

    int tile = blockIdx.x, buf = 0;
    if (tile < sg.ntiles) { fetch(tile); stash(0); }
    __syncthreads();
    for (; tile < sg.ntiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        if (nxt < sg.ntiles && !(p.h_abl & 8)) fetch(nxt);                       // in flight under this tile's MFMAs
        const char* sb = smem + buf * slab_bytes;
        const int img = tile / sg.tiles_per_img, oh0 = (tile - img * sg.tiles_per_img) * sg.TH;
        f32x4 acc[NPT][4];
#pragma unroll
        for (int j = 0; j < NPT; ++j)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[j][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(p.h_abl & 1))
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            v8 b[NPT];
#pragma unroll
            for (int j = 0; j < NPT; ++j) b[j] = *(const v8*)(sb + boff[j] + kh * sg.row_bytes);
#pragma unroll
            for (int j = 0; j < NPT; ++j)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[j][ct] = MM::mma(wf[kh][ct], b[j], acc[j][ct]);
        }
        // epilogue: lane holds output channels 16 ct + 4 fq .. + 3 of position (row r, column c0 + fr): 8-byte stores
        AT* const obase = (AT*)p.out + ((size_t)img * OH + oh0) * OW * p.Cout;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            if (inside[j] && oh0 + orow[j] < OH) {                               // wave-uniform
                AT* o = obase + ooff[j];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    f32x4 v = acc[j][ct];
                    if (!(p.h_abl & 4)) Act<AT>::st4(o + ct * 16, make_float4(v[0], v[1], v[2], v[3]));
                    if (p.stats) {
                        v[0] = Act<AT>::rnd(v[0]); v[1] = Act<AT>::rnd(v[1]); v[2] = Act<AT>::rnd(v[2]); v[3] = Act<AT>::rnd(v[3]);
                        cs[ct] += v;
                        cq[ct] += v * v;
                    }
                }
            }
        }
        if (nxt < sg.ntiles) stash(buf ^ 1);
        __syncthreads();                                                         // next slab visible, this one free
        buf ^= 1;
    }
    if (p.stats) {                                                               // one record per workgroup
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s_ = cs[ct][r], q_ = cq[ct][r];
                s_ += row_ror<8>(s_); q_ += row_ror<8>(q_);
                s_ += row_ror<4>(s_); q_ += row_ror<4>(q_);
                s_ += row_ror<2>(s_); q_ += row_ror<2>(q_);
                s_ += row_ror<1>(s_); q_ += row_ror<1>(q_);
                if (fr == 0) {
                    const int col = ct * 16 + fq * 4 + r;
                    red[(wave * 64 + col) * 2 + 0] = s_;
                    red[(wave * 64 + col) * 2 + 1] = q_;
                }
            }
        __syncthreads();
        if (t < 64) {
            float s_ = 0.f, q_ = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s_ += red[(w * 64 + t) * 2]; q_ += red[(w * 64 + t) * 2 + 1]; }
            p.stats[((size_t)blockIdx.x * 2 + 0) * p.Cout + t] = s_;
            p.stats[((size_t)blockIdx.x * 2 + 1) * p.Cout + t] = q_;
        }
    }
}

template <typename AT>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvArgs p) {
    __shared__ float red[16][64][2];
    const int t = threadIdx.x, tc = t & 15, rl = t >> 4;
    const int chunk = blockIdx.x, m0 = chunk * 32, n = blockIdx.y * 64 + tc * 4;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), q4 = s4;
    const float4 bv = p.bias ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    // compact row list: slab row m is LIST row m (position row_pos[m]); chunks past the end of the list only write their empty record
    const int Meff = p.row_count ? min(*p.row_count, p.M) : p.M;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        int m = m0 + rl * 2 + rr;
        if (m >= Meff) continue;
        const size_t pos = p.row_pos ? (size_t)p.row_pos[m] : (size_t)m;
        bool live = p.row_mask ? (p.row_mask[pos] != 0) : true;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            // slab loads four at a time (all in flight before the first add): with one load per iteration the loop was a chain of
            // ksplit dependent round trips - 14.8 us per launch in the round-3 profile, half of a deep voxel level's time.  The sum
            // order (z ascending) is unchanged.
            const float* sp = p.slab + (size_t)m * p.Cout + n;
            const size_t zs = (size_t)p.M * p.Cout;
            int z = 0;
            for (; z + 4 <= p.ksplit; z += 4) {
                const float4 x0 = *(const float4*)(sp + (size_t)z * zs), x1 = *(const float4*)(sp + (size_t)(z + 1) * zs);
                const float4 x2 = *(const float4*)(sp + (size_t)(z + 2) * zs), x3 = *(const float4*)(sp + (size_t)(z + 3) * zs);
                v.x += x0.x; v.y += x0.y; v.z += x0.z; v.w += x0.w;
                v.x += x1.x; v.y += x1.y; v.z += x1.z; v.w += x1.w;
                v.x += x2.x; v.y += x2.y; v.z += x2.z; v.w += x2.w;
                v.x += x3.x; v.y += x3.y; v.z += x3.z; v.w += x3.w;
            }
            for (; z < p.ksplit; ++z) {
                const float4 x = *(const float4*)(sp + (size_t)z * zs);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            else if (p.act == 2) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
        }
        AT* o = (AT*)p.out + pos * p.Cout + n;
        if (p.accumulate) { float4 e = Act<AT>::ld4(o); v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
        Act<AT>::st4(o, v);
        if (sizeof(AT) == 2) v = make_float4(Act<AT>::rnd(v.x), Act<AT>::rnd(v.y), Act<AT>::rnd(v.z), Act<AT>::rnd(v.w));
        s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
        q4.x += v.x * v.x; q4.y += v.y * v.y; q4.z += v.z * v.z; q4.w += v.w * v.w;
    }
    if (p.stats) {
        float* d = &red[rl][tc * 4][0];
        d[0] = s4.x; d[1] = q4.x; d[2] = s4.y; d[3] = q4.y; d[4] = s4.z; d[5] = q4.z; d[6] = s4.w; d[7] = q4.w;
        __syncthreads();
        if (t < 64) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { s += red[w][t][0]; q += red[w][t][1]; }
            p.stats[((size_t)chunk * 2 + 0) * p.Cout + blockIdx.y * 64 + t] = s;
            p.stats[((size_t)chunk * 2 + 1) * p.Cout + blockIdx.y * 64 + t] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight packing
// one packed operand element: bf16 hi (+ bf16 lo = the part hi dropped, 3-product mode), or f16 (f16 storage mode)
__device__ __forceinline__ void prep_store(bool f16, uint16_t* hi, bf16_t* lo, size_t idx, float v) {
    if (f16) { hi[idx] = __builtin_bit_cast(uint16_t, (f16_t)v); return; }
    const bf16_t h = (bf16_t)v;
    hi[idx] = __builtin_bit_cast(uint16_t, h);
    if (lo) lo[idx] = (bf16_t)(v - (float)h);
}

// index of packed element (row, k): row-major [rows][kpad], or (frag) MFMA-fragment-major - the [16 x 32] block (row / 16, k / 32) is
// 512 contiguous elements, lane (row % 16, (k % 32) / 8) of an A fragment owns 8 contiguous ones (include/tricolo_hip.h)
__device__ __forceinline__ size_t prep_index(bool frag, int row, int k, int kpad) {
    if (!frag) return (size_t)row * kpad + k;
    return ((size_t)(row >> 4) * (kpad >> 5) + (k >> 5)) * 512 + ((((k & 31) >> 3) * 16 + (row & 15)) << 3) + (k & 7);
}

// four consecutive packed elements (idx % 4 == 0): one 8-byte store per operand array instead of four 2-byte ones
__device__ __forceinline__ void prep_store4(bool f16, uint16_t* hi, bf16_t* lo, size_t idx, float4 v) {
    if (f16) {
        f16x4 h;
        h[0] = (f16_t)v.x; h[1] = (f16_t)v.y; h[2] = (f16_t)v.z; h[3] = (f16_t)v.w;
        *(f16x4*)(hi + idx) = h;
        return;
    }
    bf16x4 h;
    h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
    *(bf16x4*)(hi + idx) = h;
    if (lo) {
        bf16x4 l;
        l[0] = (bf16_t)(v.x - (float)h[0]); l[1] = (bf16_t)(v.y - (float)h[1]); l[2] = (bf16_t)(v.z - (float)h[2]); l[3] = (bf16_t)(v.w - (float)h[3]);
        *(bf16x4*)(lo + idx) = l;
    }
}

// dst[row][tap * inner_pad + i] (bf16 hi / lo, zero padded to Kpad) from an fp32 tensor addressed by strides.
// forward:  row = co, inner = ci;   dgrad: row = ci, inner = co  (same tensor, swapped strides).
__global__ void weight_prep_kernel(const float* __restrict__ w, long s_row, long s_tap, long s_inner, int rows, int ntaps,
                                   int inner, int inner_pad, int Kpad, uint16_t* __restrict__ hi, bf16_t* __restrict__ lo, int f16, int frag) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)rows * Kpad;
    if (idx >= total) return;
    int row = (int)(idx / Kpad), k = (int)(idx - (long)row * Kpad);
    int tap = k / inner_pad, i = k - tap * inner_pad;
    float v = 0.f;
    if (tap < ntaps && i < inner) v = w[row * s_row + tap * s_tap + i * s_inner];
    prep_store(f16 != 0, hi, lo, prep_index(frag != 0, row, k, Kpad), v);
}

// All layers of a tower in ONE launch: blockIdx.y selects the descriptor, blockIdx.x grid-strides over its work.
// The element-wise form above reads the parameter with whatever stride the layout dictates (36 B between consecutive
// input channels of a torchvision [Cout,Cin,3,3] weight, a whole filter between consecutive output channels for the
// data-gradient operand): every wave touches dozens of cache lines per load.  The three layouts the towers use are
// therefore transposed through LDS - contiguous runs in, contiguous bf16 rows out; anything else takes the generic loop.
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const TriPrepDesc* __restrict__ descs) {
    __shared__ float tile[64 * 73];                              // 18.25 KiB: one filter row, or a 64 x (<= 72) + 1 transpose tile
    const TriPrepDesc d = descs[blockIdx.y];
    uint16_t* hi = (uint16_t*)d.hi;
    bf16_t* lo = (bf16_t*)d.lo;
    const bool f16 = d.fmt == TRI_FMT_F16, frag = d.frag != 0;
    const int t = threadIdx.x, nt = d.ntaps;
    const long span = (long)d.inner * nt;
    if (nt > 1 && d.s_tap == 1 && d.s_inner == nt && d.s_row == span && span <= 64 * 73 && d.inner == d.inner_pad) {
        // forward operand of a [Cout,Cin,taps] weight: one contiguous filter per row; dst[tap * Cin + ci] = src[ci * taps + tap]
        const bool vec = (span & 3) == 0 && (d.inner_pad & 3) == 0 && (d.kpad & 3) == 0 && (((uintptr_t)d.w) & 15) == 0;
        for (int row = blockIdx.x; row < d.rows; row += gridDim.x) {
            const float* src = d.w + (size_t)row * d.s_row;
            if (vec) {                                             // 16-byte loads, 8-byte stores (the element-wise form: 2-byte stores, 128 B per wave)
                for (int e = t * 4; e < span; e += 1024) *(float4*)(tile + e) = *(const float4*)(src + e);
                __syncthreads();
                for (int k = t * 4; k < d.kpad; k += 1024) {
                    const int tap = k / d.inner_pad, i = k - tap * d.inner_pad;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (tap < nt) v = make_float4(tile[i * nt + tap], tile[(i + 1) * nt + tap], tile[(i + 2) * nt + tap], tile[(i + 3) * nt + tap]);
                    prep_store4(f16, hi, lo, prep_index(frag, row, k, d.kpad), v);
                }
            } else {
                for (int e = t; e < span; e += 256) tile[e] = src[e];
                __syncthreads();
                for (int k = t; k < d.kpad; k += 256) {
                    const int tap = k / d.inner_pad, i = k - tap * d.inner_pad;
                    prep_store(f16, hi, lo, prep_index(frag, row, k, d.kpad), tap < nt ? tile[i * nt + tap] : 0.f);
                }
            }
            __syncthreads();
        }
        return;
    }
    if (d.s_inner == 1 && d.s_tap == d.inner && d.s_row == span && d.inner == d.inner_pad && (span & 3) == 0 && (d.kpad & 3) == 0 &&
        (((uintptr_t)d.w) & 15) == 0) {
        // (round 5) the packed row IS the source row ([Cout][taps][Cin] spconv weights, forward operand; linear layers): a cast, four
        // elements per thread, 16-byte loads and 8-byte stores in either order.  These operands took the element-wise loop below -
        // 64-bit divisions and 2-byte stores per element, most of the voxel tower's 41 us packing launch.
        const unsigned q4 = (unsigned)(d.kpad >> 2), total4 = (unsigned)d.rows * q4, used = (unsigned)span;
        for (unsigned e = blockIdx.x * 256u + t; e < total4; e += gridDim.x * 256u) {
            const unsigned row = e / q4, k = (e - row * q4) * 4u;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < used) v = *(const float4*)(d.w + (size_t)row * span + k);
            prep_store4(f16, hi, lo, prep_index(frag, (int)row, (int)k, d.kpad), v);
        }
        return;
    }
    const bool torch_t = nt > 1 && nt <= 9 && d.s_tap == 1 && d.s_row == nt && d.s_inner == (long)d.rows * nt;   // [inner][rows][taps]
    const bool plane_t = d.s_row == 1 && d.s_tap == d.rows && d.s_inner == (long)d.rows * nt;                   // [inner][taps][rows]
    if ((torch_t || plane_t) && d.inner == d.inner_pad) {
        // data-gradient operand: dst[row = ci][tap * Cout + co].  Tiles of RT rows x 64 inner; a source run is RT * taps (torch_t)
        // or RT (plane_t, one tap at a time) contiguous floats per inner index.
        const int RT = torch_t ? 8 : 32;
        const int run = torch_t ? RT * nt : RT, ld = run + 1;     // odd leading dimension: conflict-free column reads
        const int planes = torch_t ? 1 : nt;
        const int rtiles = (d.rows + RT - 1) / RT, itiles = (d.inner + 63) / 64;
        const long ntile = (long)rtiles * itiles * planes;
        for (long tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
            const int plane = (int)(tl / ((long)rtiles * itiles));
            const int rem = (int)(tl - (long)plane * rtiles * itiles);
            const int r0 = (rem / itiles) * RT, i0 = (rem % itiles) * 64;
            const int rows_here = min(RT, d.rows - r0);
            const int run_here = torch_t ? rows_here * nt : rows_here;
            for (int e = t; e < 64 * run; e += 256) {
                const int il = e / run, r = e - il * run;
                float v = 0.f;
                if (i0 + il < d.inner && r < run_here)
                    v = d.w[(size_t)(i0 + il) * d.s_inner + (torch_t ? (size_t)r0 * nt + r : (size_t)plane * d.s_tap + r0 + r)];
                tile[il * ld + r] = v;
            }
            __syncthreads();
            if ((d.inner & 3) == 0 && (d.inner_pad & 3) == 0 && (d.kpad & 3) == 0) {
                for (int e = t; e < 16 * run; e += 256) {         // four inner indices per thread: 8-byte stores
                    const int il = (e & 15) * 4, r = e >> 4;      // r = row_l * taps + tap (torch_t) or row_l (plane_t)
                    const int row_l = torch_t ? r / nt : r, tap = torch_t ? r - row_l * nt : plane;
                    if (row_l < rows_here && i0 + il < d.inner)
                        prep_store4(f16, hi, lo, prep_index(frag, r0 + row_l, tap * d.inner_pad + i0 + il, d.kpad),
                                    make_float4(tile[il * ld + r], tile[(il + 1) * ld + r], tile[(il + 2) * ld + r], tile[(il + 3) * ld + r]));
                }
            } else
            for (int e = t; e < 64 * run; e += 256) {
                const int il = e & 63, r = e >> 6;                // r = row_l * taps + tap (torch_t) or row_l (plane_t)
                const int row_l = torch_t ? r / nt : r, tap = torch_t ? r - row_l * nt : plane;
                if (row_l < rows_here && i0 + il < d.inner)
                    prep_store(f16, hi, lo, prep_index(frag, r0 + row_l, tap * d.inner_pad + i0 + il, d.kpad), tile[il * ld + r]);
            }
            __syncthreads();
        }
        // zero the K padding of every row (kpad is the tap * inner extent rounded up to 32)
        const int kused = nt * d.inner_pad, padw = d.kpad - kused;
        if (padw > 0)
            for (long e = (long)blockIdx.x * 256 + t; e < (long)d.rows * padw; e += (long)gridDim.x * 256) {
                const int row = (int)(e / padw), k = kused + (int)(e - (long)row * padw);
                prep_store(f16, hi, lo, prep_index(frag, row, k, d.kpad), 0.f);
            }
        return;
    }
    const long total = (long)d.rows * d.kpad;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int row = (int)(idx / d.kpad), k = (int)(idx - (long)row * d.kpad);
        int tap = k / d.inner_pad, i = k - tap * d.inner_pad;
        float v = 0.f;
        if (tap < d.ntaps && i < d.inner) v = d.w[row * d.s_row + tap * d.s_tap + i * d.s_inner];
        prep_store(f16, hi, lo, prep_index(frag, row, k, d.kpad), v);
    }
}

extern "C" int tri_weight_prep_multi(const TriPrepDesc* descs_dev, int n, void* stream) {
    if (n <= 0) return TRI_OK;
    static int gx = 0;                                               // workgroups per descriptor (tuning aid: TRICOLO_PREP_GRID)
    if (!gx) gx = 512;   // (256 -> 512: the three packing launches of a step 81 -> 68 us; 1024: 75)
    weight_prep_multi_kernel<<<dim3(gx, n), 256, 0, (hipStream_t)stream>>>(descs_dev);
    return tri_check_launch("tri_weight_prep_multi");
}

extern "C" int tri_weight_prep(const float* w, long s_row, long s_tap, long s_inner, int rows, int ntaps, int inner,
                               int inner_pad, void* w_hi, void* w_lo, int fmt, int frag, void* stream) {
    if (fmt == TRI_FMT_F16 && w_lo) { tri_set_error("tri_weight_prep: the f16 operand format has no lo part"); return TRI_ERR_ARG; }
    if (inner_pad % 4 != 0 || inner > inner_pad) { tri_set_error("tri_weight_prep: inner_pad must be a multiple of 4 >= inner"); return TRI_ERR_ARG; }
    if (frag && rows % 16 != 0) { tri_set_error("tri_weight_prep: the fragment-major order needs rows % 16 == 0"); return TRI_ERR_ARG; }
    int Kpad = (ntaps * inner_pad + 31) / 32 * 32;
    long total = (long)rows * Kpad;
    int blocks = (int)((total + 255) / 256);
    weight_prep_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(w, s_row, s_tap, s_inner, rows, ntaps, inner, inner_pad, Kpad,
                                                                (uint16_t*)w_hi, (bf16_t*)w_lo, fmt == TRI_FMT_F16 ? 1 : 0, frag ? 1 : 0);
    return tri_check_launch("tri_weight_prep");
}

// ------------------------------------------------------------------------------------------------------ launcher

static int ilog2_exact(int v) {
    for (int s = 0; s < 31; ++s) if ((1 << s) == v) return s;
    return -1;
}

template <int BN, int NSPLIT, typename AT>
static int launch_conv(const ConvArgs& a, hipStream_t stream) {
    constexpr int STAGE = NSPLIT * (128 * 64 + BN * 64);
    constexpr int WAVES_M = (BN >= 64) ? 2 : 4;
    size_t smem = 2 * STAGE + 512 + WAVES_M * BN * 2 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_igemm_kernel<BN, NSPLIT, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    int mt = (a.M + 127) / 128, nt = a.Cout / BN;
    conv_igemm_kernel<BN, NSPLIT, AT><<<dim3(mt * nt, a.ksplit), 256, smem, stream>>>(a);
    int rc = tri_check_launch("tri_conv");
    if (rc || a.ksplit == 1) return rc;
    conv_splitk_finish_kernel<AT><<<dim3((a.M + 31) / 32, a.Cout / 64), 256, 0, stream>>>(a);
    return tri_check_launch("tri_conv_splitk_finish");
}

static int conv_bn(int cout) { return cout % 128 == 0 ? 128 : (cout % 64 == 0 ? 64 : 32); }

// One plan per (geometry, direction), used by the launchers AND by the workspace / statistics-size queries.
struct ConvPlan {
    int bn;               // output-channel tile
    int halo;             // 0, or TM (4 / 2) of conv_halo2d_kernel: 2D 3x3 / 1 / pad 1, 16-bit storage, Cin % 64 == 0, Cout % 64 == 0
    int h_tr, h_rows, h_slab_bytes, h_nr, h_mtiles, h_dbuf;
    int h_v5;             // conv_halo_rows_kernel (row-unit pipeline, one workgroup per CU) instead of conv_halo2d_kernel
    int h_grid;           // persistent workgroups of the halo launch (a multiple of the output-channel tiles)
    int h_wgrec;          // 1: one BatchNorm record per workgroup (h_grid / (Cout / 64) records), 0: one per row tile (h_mtiles)
    int stem;             // 1: conv_stem_kernel (2D, 4 stored input channels, stride 2, Cout 64, 16-bit storage); records = stem_grid
    int stem_grid;
    int vox0;             // 1: conv_vox0_kernel (conv_vox.hip: level 0 of the voxel tower, 16-bit storage); records = vox0_grid
    int vox0_grid;
    int voxb, voxb_grid;  // 1: conv_voxb_kernel (conv_voxg.hip: level 1 of the voxel tower, ranked active rows over bricks); records = voxb_grid
    int s2g, s2g_units;   // 1: conv_s2g_kernel (conv_s2g.hip: forward of the 3x3 / 2 layers that open layer3 / layer4, 16-bit storage); records = s2g_units
    int voxg;             // 1: conv_voxg_kernel (conv_voxg.hip: SubMConv3d on 2^3 / 4^3 / 8^3 grids, forward and data gradient, 16-bit storage);
    int voxg_units, voxg_ct, voxg_spu;        // records = units
    int c64;              // 1: conv_c64_kernel (conv_c64.hip: 64 -> 64 channels, 2D 3x3 / 1 / pad 1, 16-bit storage); records = c64_grid
    int c64_grid;
    int s2d;              // 1: conv_s2d_kernel (conv_c64.hip: data gradient of a 64 -> 128 channel 3x3 / 2 layer, 16-bit storage)
    int pw, pw_tr;        // 1: conv_pw_kernel (conv_pw.hip: 1x1 / 2 shortcut convolution), pw_tr: the plan is the data gradient's; records = row tiles
    int dma;              // 1: LDS-DMA kernel (16-bit activation storage, Cin % 64 == 0), 64-wide k-steps
    int nunits;           // k-steps (32 wide, or 64 wide for the DMA kernel)
    int ksplit, per_split;
};

// Split-K target: workgroups a small-M layer is split up to.  Every split costs M x Cout fp32 of slab write + re-read,
// so the sweep (profiles/r1/README.md) favours ~one workgroup per CU over the 3 per CU the fp32-staging kernel liked.
static int conv_target_blocks() {                               // tuning aid: TRICOLO_CONV_BLOCKS overrides it
    static int v = -1;
    if (v < 0) v = 0;
    return v;
}

static bool halo_disabled() {                                   // A/B switch: TRICOLO_NO_HALO=1 keeps conv_dma_kernel for the 3x3 layers
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_HALO"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

static int num_cus() {
    static int v = 0;
    if (!v) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) v = pr.multiProcessorCount;
        else v = 256;                                                // MI355X; also the answer on a box without a GPU (plan queries)
    }
    return v;
}

int tri_internal_num_cus() { return num_cus(); }                     // for conv_wgrad.hip (not part of the C ABI header)

// geometry of conv_halo2d_kernel for row tiles of 64 * TM positions; false when the layer does not fit
static bool halo_geometry(int B, int H, int W, int cin, int cout, int TM, ConvPlan* pl) {
    const int BM = 64 * TM, P = W + 2;
    if (W > BM) return false;
    const int TR = BM / W;
    int k;                                                           // image segments of a tile
    if (H % TR == 0) k = 1;                                          // a tile stays inside one image
    else if (TR % H == 0) k = TR / H;                                // whole images per tile
    else return false;                                               // tiles would cross image boundaries at arbitrary rows
    const int rows = TR + 2 * k;
    const int slab = ((rows * P + 1) * 128 + 4095) / 4096 * 4096;
    if (slab / 4096 > 10) return false;                              // HALO_MAX_ROUNDS
    (void)cout;
    pl->halo = TM; pl->h_tr = TR; pl->h_rows = rows; pl->h_slab_bytes = slab; pl->h_nr = 3; pl->h_dbuf = 0;
    pl->h_mtiles = (B * H + TR - 1) / TR;
    {
        const int NT = cout / 64, items = pl->h_mtiles * NT;
        const size_t smem = (size_t)slab + (size_t)HALO_NR * 8192 + 2048;
        int per_cu = (int)(163840 / (smem + 256));
        per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);
        int slots = num_cus() * per_cu / NT * NT;
        if (slots < NT) slots = NT;
        pl->h_grid = items < slots ? items : slots;
        pl->h_wgrec = items >= 4 * slots ? 1 : 0;
        static int rows_kernel = -1;                                  // A/B switch: TRICOLO_HALO_ROWS=0 keeps conv_halo2d_kernel
        if (rows_kernel < 0) { const char* e = getenv("TRICOLO_HALO_ROWS"); rows_kernel = e ? atoi(e) : 1; }
        pl->h_v5 = 0;
        // where the row-unit pipeline runs (TRICOLO_HALO_ROWS: 0 nowhere, 2 everywhere it fits): 64 input channels (resident filter
        // bank, only slabs stream: layer1 27 against 34 us) and launches with at most 1.5 tiles per workgroup (layer4 of the bench
        // shape 27 against 45 us, layer3 29 / 27 against 30 / 28).  With more weight-streaming tiles per workgroup its one workgroup
        // per CU pays every DMA piece in issue time and the three co-resident workgroups of conv_halo2d_kernel are faster (layer2
        // 24 / 24 against 26 / 25 us; steps with the row kernel everywhere: config 4 / 5 3.30-3.32 / 24.16 ms against 3.28-3.29 / 24.01)
        int g = num_cus() / NT * NT;
        if (g < NT) g = NT;
        static int half_tiles = -1;                                   // tuning aid: TRICOLO_HALO_ROWS_HALFTILES (tiles per workgroup x 2)
        if (half_tiles < 0) half_tiles = 3;
        // (round 4: with >= 256 input channels - four or more weight chunks per tile - the row-unit kernel also wins launches of several
        //  rounds: 384 images of 8 x 8 x 256 47 -> 32 us, 768 images of 4 x 4 x 512 85 -> 65 us; TRICOLO_HALO_ROWS_DEEP=0 restores the tile rule)
        static int deep = -1;
        if (deep < 0) deep = 1;
        if (rows_kernel && (cin == 64 || (deep && cin >= 256) || 2 * items <= half_tiles * g || rows_kernel == 2) && (TM == 2 || (TM == 3 && cin != 64)) &&
            2 * (size_t)slab + 3 * (size_t)HROWS_SLOT + 8192 <= 163840) {
            pl->h_v5 = 1;
            pl->h_grid = items < g ? items : g;
            pl->h_wgrec = 1;                                          // one BatchNorm record per workgroup
        }
    }
    return true;
}

static bool stem_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_STEM"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
static bool stem_geometry(int B, int IH, int IW, int cin, int OH, int OW, int cout, int KD, int KH, int KW, int stride, int pd, int ph, int pw,
                          StemGeom* g) {
    if (KD != 1 || pd != 0 || cin != 4 || stride != 2 || cout != 64 || (KH != 3 && KH != 5 && KH != 7) || KW > 8 || OW % 16 || IW % 2) return false;
    if (pw < 0 || ph < 0 || IW + pw > 2 * OW + 6 || (long)B * IH * IW * 8 >= ((long)1 << 31)) return false;
    const int per_row = OW / 16;
    static int npt = -1;                                             // position tiles per wave: 2 (two workgroups per CU) unless overridden
    if (npt < 0) npt = 2;
    if (per_row > 4 * npt) return false;
    int TH = 4 * npt / per_row;                                      // <= 4 * npt position tiles per workgroup tile
    if (TH > OH) TH = OH;
    g->npt = npt;
    g->TH = TH;
    g->slab_rows = (TH - 1) * 2 + KH;
    g->row_bytes = (OW + 3) * 16;
    g->ptiles = TH * per_row;
    g->tiles_per_img = (OH + TH - 1) / TH;
    g->ntiles = B * g->tiles_per_img;
    if (g->slab_rows * (IW / 2) > 256 * 4) return false;             // MAXLD of the kernel
    if (2 * g->slab_rows * g->row_bytes + 4 * 64 * 2 * 4 > 160 * 1024 / 2) return false;
    const int slots = num_cus() * (npt == 4 ? 1 : 2);
    g->grid = g->ntiles < slots ? g->ntiles : slots;
    return true;
}

static bool dma_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_DMA"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

// split_mode: 0 bf16 operands / fp32 activations, 1 bf16x3 (hi + lo operands), 2 16-bit operands AND activation storage
// (bf16 or f16: same kernels, same plans)
// row_list: the launch walks a compact active-row list (voxel levels): the workgroups past ceil(count / 128) leave at once, so the
// dense tile count overstates the launch.  The count lives on the device; the plan assumes the usual occupancy of the voxel grids
// (<= 25 % at every level of the synthetic and ShapeNet-like shapes) when it sizes split-K - a denser list just runs more
// workgroups than planned.
static ConvPlan conv_make_plan(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                               int pd, int ph, int pw, int split_mode, int row_list = 0) {
    ConvPlan pl{};
    long M = (long)B * OD * OH * OW;
    int ntaps = KD * KH * KW;
    int kpad = (ntaps * cin + 31) / 32 * 32;
    int bn = conv_bn(cout);
    int blocks = (int)((M + 127) / 128) * (cout / bn);
    const int blocks_dense = blocks;                                          // the tile-width rule below keeps using the dense count
    int occ_div = 1;
    if (row_list) {
        // measured occupancies of the voxel levels (synthetic shapes, 32^3 and 64^3 inputs): 13-18 % on 16^3 and finer grids, ~25 % at 8^3,
        // ~45 % at 4^3, ~95 % at 2^3 (coarse levels fill up)
        const int side = OD > OH ? (OD > OW ? OD : OW) : (OH > OW ? OH : OW);
        occ_div = side >= 8 ? 4 : (side >= 4 ? 2 : 1);
        blocks = (blocks + occ_div - 1) / occ_div;
    }
    // (round 3: 32-wide output tiles exist too - built for the data gradient of voxel level 1, 64 -> 32 channels, the slowest kernel of
    // the voxel backward on the register-staged gather: 43-57 -> 24 us.  Config 2's step does not move (1.050-1.063 against 1.058-1.061 ms),
    // config 4's - where the voxel backward is the tail since the image tower got shorter - gains 0.02 ms in three of three pairs
    // (2.896-2.922 against 2.920-2.944): on by default, TRICOLO_DMA32=0 switches them off)
    static int no32 = -1;
    if (no32 < 0) no32 = 0;
    if (split_mode == 2 && cin % 64 == 0 && (cout % 64 == 0 || (cout % 32 == 0 && !no32)) && !dma_disabled()) pl.dma = 1;
    pl.bn = bn;
    {
        TriVox0Geom vg;
        if (split_mode == 2 && tri_internal_vox0_geometry(B, ID, IH, IW, cin, OD, OH, OW, cout, KD, KH, KW, stride, pd, ph, pw, &vg)) {
            pl.vox0 = 1; pl.vox0_grid = vg.grid; pl.bn = 32; pl.nunits = 4; pl.ksplit = 1; pl.per_split = 4;
            return pl;
        }
        TriVoxbGeom vb;
        if (split_mode == 2 && tri_internal_voxb_geometry(B, ID, IH, IW, cin, OD, OH, OW, cout, KD, KH, KW, stride, pd, ph, pw, &vb)) {
            pl.voxb = 1; pl.voxb_grid = vb.grid; pl.bn = 64; pl.nunits = 27; pl.ksplit = 1; pl.per_split = 27;
            return pl;
        }
        TriVoxgGeom vgg;
        if (split_mode == 2 && tri_internal_voxg_geometry(B, ID, IH, IW, cin, OD, OH, OW, cout, KD, KH, KW, stride, pd, ph, pw, &vgg)) {
            // coarse voxel grids: activations stationary in LDS, weights straight into MFMA registers (forward and data gradient;
            // the launch walks the grid by the site mask, one BatchNorm record per unit of samples)
            pl.voxg = 1; pl.voxg_units = vgg.nunits; pl.voxg_ct = vgg.ct; pl.voxg_spu = vgg.spu;
            pl.bn = vgg.ct; pl.nunits = kpad / 32; pl.ksplit = 1; pl.per_split = pl.nunits;
            return pl;
        }
    }
    {
        StemGeom sgm;
        if (split_mode == 2 && !stem_disabled() && ID == 1 && OD == 1 &&
            stem_geometry(B, IH, IW, cin, OH, OW, cout, KD, KH, KW, stride, pd, ph, pw, &sgm)) {
            pl.stem = 1; pl.stem_grid = sgm.grid; pl.bn = 64; pl.nunits = KH; pl.ksplit = 1; pl.per_split = KH;
            return pl;
        }
    }
    {
        TriPwGeom pg;
        if (pl.dma && split_mode == 2 && !row_list && tri_internal_pw_geometry(B, ID, IH, IW, cin, OD, OH, OW, cout, KD, KH, KW, stride, pd, ph, pw, &pg)) {
            // 1x1 / 2 shortcut convolution: conv_pw_kernel; a call it does not take (row mask / bias / activation / accumulate) runs
            // conv_dma_kernel without split-K (the fields below are its plan; same record count: one per 128-row tile)
            pl.pw = 1; pl.pw_tr = pg.transposed; pl.bn = pg.bn; pl.nunits = kpad / 64; pl.ksplit = 1; pl.per_split = pl.nunits;
            return pl;
        }
    }
    {
        TriC64Geom cg;
        if (pl.dma && split_mode == 2 && tri_internal_c64_geometry(B, ID, IH, IW, cin, OD, OH, OW, cout, KD, KH, KW, stride, pd, ph, pw, &cg)) {
            // 64 -> 64 channels: conv_c64_kernel (filter bank in registers; conv_c64.hip), forward and data gradient; a call with a
            // row mask / bias / activation takes conv_dma_kernel (nunits etc. below are its plan)
            pl.c64 = 1; pl.c64_grid = cg.grid; pl.bn = 64; pl.nunits = kpad / 64; pl.ksplit = 1; pl.per_split = pl.nunits;
            return pl;
        }
    }
    {
        TriS2gGeom g2;
        if (pl.dma && split_mode == 2 && tri_internal_s2g_geometry(B, ID, IH, IW, cin, OD, OH, OW, cout, KD, KH, KW, stride, pd, ph, pw, &g2)) {
            // 3x3 / 2 layers with >= 128 input channels: conv_s2g_kernel (space-to-depth slab, weights straight into MFMA registers); a call with a
            // row mask / list / bias / activation / accumulate is refused (the layer's operand is packed fragment-major).  Planned whether or
            // not the call carries a row list: the family query packs the operand without knowing, so a row-list call must reach
            // conv_dispatch's refusal instead of reading the fragment-major operand row-major in conv_dma_kernel
            pl.s2g = 1; pl.s2g_units = g2.nunits; pl.bn = 64; pl.nunits = kpad / 32; pl.ksplit = 1; pl.per_split = pl.nunits;
            return pl;
        }
    }
    {
        TriC64Geom sg;
        // GEMM view of a data-gradient call (the output grid is twice the input grid: no forward layer looks like this)
        if (pl.dma && split_mode == 2 && tri_internal_s2d_geometry(B, ID, IH, IW, cin, OD, OH, OW, cout, KD, KH, KW, stride, pd, ph, pw, &sg))
            pl.s2d = 1;                                                        // (a call with a row mask / list takes conv_dma_kernel: its plan below)
    }
    if (pl.dma && cout % 64 == 0 && !halo_disabled() && KD == 1 && KH == 3 && KW == 3 && stride == 1 && pd == 0 && ph == 1 && pw == 1 && ID == 1 && OD == 1 &&
        IH == OH && IW == OW && (long)B * IH * IW * cin * 2 < ((long)1 << 31)) {
        // 128-position tiles: a ~25 KB slab + 24 KB ring lets three workgroups share a CU (the kernel is bound by instruction
        // issue and latency, not by operand bytes: occupancy matters more than the bigger tile's reuse)
        ConvPlan t2{}, t3{};
        const ConvPlan* pick = halo_geometry(B, IH, IW, cin, cout, 2, &t2) ? &t2 : nullptr;
        // 192-position tiles on the row-unit kernel where they save a ROUND of workgroups: a tile of 1.5 x the positions costs ~1.2 x the
        // time (1.5 x the MFMAs per barrier, weight piece and DMA wait), so they win when 1.2 x their rounds < the 128-position rounds
        // (layer3 of the bench shape: 384 tiles on 256 CUs -> 256 tiles, 26.5 -> 19.7 us forward, 24.9 -> 18.1 us data gradient; 384 images:
        // 768 tiles in three rounds -> 512 in two, 47 (conv_halo2d_kernel) -> 32 us; round 4, TRICOLO_HALO_TM3=0 switches them off)
        static int tm3 = -1;
        if (tm3 < 0) { const char* e = getenv("TRICOLO_HALO_TM3"); tm3 = (e && e[0] == '0') ? 0 : 1; }
        if (pick && pick->h_v5 && tm3 && cin != 64 && halo_geometry(B, IH, IW, cin, cout, 3, &t3) && t3.h_v5) {
            const int nt = cout / 64, g = t2.h_grid > t3.h_grid ? t2.h_grid : t3.h_grid;
            const int r2 = (t2.h_mtiles * nt + g - 1) / g, r3 = (t3.h_mtiles * nt + g - 1) / g;
            if (12 * r3 < 10 * r2) pick = &t3;
        }
        // (round 6: 192 x 32 tiles - TN = 2 fragments per wave - for launches whose 64-channel tiles leave CUs idle were built and measured:
        //  layer4 of the bench shape, 256 tiles instead of 192, ran 26 / 25 us against 24 / 23: six MFMAs per five fragment reads and the
        //  same barriers per k-step cost more than the idle quarter of the GPU; dropped)
        if (pick) {
            pl.halo = pick->halo; pl.h_tr = pick->h_tr; pl.h_rows = pick->h_rows; pl.h_slab_bytes = pick->h_slab_bytes;
            pl.h_nr = pick->h_nr; pl.h_mtiles = pick->h_mtiles; pl.h_dbuf = pick->h_dbuf; pl.h_grid = pick->h_grid; pl.h_wgrec = pick->h_wgrec; pl.h_v5 = pick->h_v5;
            pl.bn = 64; pl.nunits = 9 * (cin / 64); pl.ksplit = 1; pl.per_split = pl.nunits;
            return pl;
        }
    }
    {   // The DMA kernel is latency-bound, not MFMA-bound, at this workload's layer sizes: 128x64 tiles (24 KiB stages, three
        // workgroups per CU, twice the workgroups) beat 128x128 on every layer measured up to 384 wide tiles (sweep in
        // profiles/r1/README.md).  Wide tiles are kept for launches that fill the GPU several times over anyway.
        static int narrow = -1;
        if (narrow < 0) narrow = 1024;
        // (decided on the DENSE tile count also for row-list launches: 64^3 level 2 - 2,048 dense tiles, 365 live - takes 42 us with
        // 128-wide tiles against 54 with 64-wide ones; the occupancy discount is for the split-K decision only)
        if (pl.dma && bn == 128 && blocks_dense < narrow) { pl.bn = 64; blocks *= 2; }
    }
    // (round 5) dense layers with a short K on the register-staged kernel (fp32 storage: the GRU's input projection, 3,072 x 256 x 768 at the
    // bench shape): 64-wide tiles that fill the GPU once instead of 128-wide tiles split over K with a finish launch (two launches, a
    // 19 MB round trip of partial sums for 8 k-steps of work)
    static int lin64 = -1;
    if (lin64 < 0) lin64 = 1;
    const bool short_linear = lin64 && !pl.dma && !row_list && ntaps == 1 && kpad <= 256 && cout % 64 == 0;
    if (short_linear && bn == 128 && blocks < num_cus() && 2 * blocks >= num_cus()) { pl.bn = 64; blocks *= 2; }
    pl.nunits = pl.dma ? kpad / 64 : kpad / 32;
    int ks = 1;
    int min_per = pl.dma ? 2 : 4;                                              // at least this many units per split
    if (blocks < 384 && pl.nunits >= 2 * min_per && cout % 64 == 0 && !(short_linear && blocks >= num_cus())) {
        const int target = conv_target_blocks() ? conv_target_blocks() : (split_mode == 2 ? 256 : 768);
        ks = (target + blocks - 1) / blocks;
        if (ks > pl.nunits / min_per) ks = pl.nunits / min_per;
        if (ks > 32) ks = 32;
        if (ks < 1) ks = 1;
    }
    if (row_list) {
        // the fp32 slabs of a row-list launch are sized for the DENSE row count (the live count is a device value): keep them under 48 MB
        const long per_split = M * cout * (long)sizeof(float);
        while (ks > 1 && ks * per_split > (48L << 20)) --ks;
    }
    pl.per_split = (pl.nunits + ks - 1) / ks;
    pl.ksplit = (pl.nunits + pl.per_split - 1) / pl.per_split;
    return pl;
}

// Pipeline depth of conv_dma_kernel (stages of [128 x 64] + [BN x 64] operand tiles in LDS, NST - 1 of them in flight under the
// MFMAs).  Launches that fill the GPU several times over run 2 stages x 3 workgroups per CU (profiles/r1/README.md: 2.01 vs 2.24 ms
// over the ResNet layers).  Launches that cannot - the deep voxel levels over a compact row list, split-K layers, small data
// gradients - are bound by the operand bytes a CU can pull per microsecond with few workgroups resident; a third stage keeps two
// k-steps in flight and still fits TWO workgroups per CU with 64-wide tiles (72 KiB each), which matters as much: round 3 measured
// 64^3 level 3 (268 live workgroups) at 56 / 47 / 85 us with 2 / 3 / 4 stages - four stages leave one workgroup per CU and the
// launch a second, nearly empty round.  128-wide tiles (32 KiB stages) keep 2.  TRICOLO_DMA_STAGES = 2 | 3 | 4 forces one depth
// everywhere, TRICOLO_DMA_STAGES_ROWS the depth of the row-list launches.
static int dma_stages_env(const char* name) {
    const char* e = getenv(name);
    int v = e ? atoi(e) : 0;
    return (v >= 2 && v <= 4) ? v : 0;
}
static int dma_stages_for(const ConvArgs& a, int bn) {
    static int forced = -1, rows = -1;
    if (forced < 0) { forced = dma_stages_env("TRICOLO_DMA_STAGES"); rows = dma_stages_env("TRICOLO_DMA_STAGES_ROWS"); }
    if (forced) return forced;
    if (a.row_count && rows) return rows;
    const long wgs = (long)((a.M + 127) / 128) * (a.Cout / bn) * a.ksplit;
    const bool small = a.row_count != nullptr || wgs <= 2L * num_cus();
    return (small && bn <= 64) ? 3 : 2;
}

template <int BN, int NST, typename AT>
static int launch_dma(const ConvArgs& a, hipStream_t stream) {
    constexpr size_t smem = NST * (128 * 128 + BN * 128) + 512 + (size_t)4 * BN * 2 * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv_dma_kernel<BN, NST, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr = true;
    }
    int mt = (a.M + 127) / 128, nt = a.Cout / BN;
    conv_dma_kernel<BN, NST, AT><<<dim3(mt * nt, a.ksplit), 256, smem, stream>>>(a);
    int rc = tri_check_launch("tri_conv(dma)");
    if (rc || a.ksplit == 1) return rc;
    conv_splitk_finish_kernel<AT><<<dim3((a.M + 31) / 32, a.Cout / 64), 256, 0, stream>>>(a);
    return tri_check_launch("tri_conv_splitk_finish");
}

// Per-row multiplier of the slab swizzle (ConvArgs::h_swz): the value whose A-fragment reads cost the fewest LDS cycles, counted with the
// gfx950 lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table: one LDS cycle per group of 16 lanes when their 16-byte slots differ
// mod 16) over every wave, fragment row block, tap and 64-byte half of a tile.  1.0 = conflict-free.
static int halo_swizzle(int W, int H, int TM) {
    static std::mutex mu;
    static std::map<long, int> memo;
    if (W < 1 || H < 1 || TM < 1) return 0;
    const long key = ((long)TM << 40) | ((long)W << 20) | (long)H;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
    }
    static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                   {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int P = W + 2, TR = 64 * TM / W, whole = (TR % H == 0);
    long best = -1;
    int best_c = 0;
    for (int c1 = 0; c1 < 8; ++c1) {
        long cyc = 0;
        for (int wave = 0; wave < 4; ++wave)
            for (int a = 0; a < TM; ++a)
                for (int tap = 0; tap < 9; ++tap)
                    for (int kk = 0; kk < 2; ++kk)
                        for (int g = 0; g < 4; ++g) {
                            int cnt[16] = {0}, mx = 0;
                            for (int l = 0; l < 16; ++l) {
                                const int lane = grp[g][l], fr = lane & 15, fq = lane >> 4;
                                const int pa = wave * 16 * TM + a * 16 + fr, j = pa / W, x = pa - j * W, i = whole ? j / H : 0;
                                const int row = j + 1 + 2 * i + tap / 3 - 1, col = x + 1 + tap % 3 - 1;
                                const int slot = (((row * P + col) & 1) << 3) | (((fq + 4 * kk) ^ (col + c1 * row)) & 7);
                                if (++cnt[slot] > mx) mx = cnt[slot];
                            }
                            cyc += mx;
                        }
        if (best < 0 || cyc < best) { best = cyc; best_c = c1; }
    }
    std::lock_guard<std::mutex> lk(mu);
    memo[key] = best_c;
    return best_c;
}

#ifdef HALO_STAMPS
static long long* g_halo_dbg = nullptr;
#endif
template <int TM, typename AT>
static int launch_halo(ConvArgs& a, const ConvPlan& pl, hipStream_t stream) {
    a.h_tr = pl.h_tr; a.h_rows = pl.h_rows; a.h_slab_bytes = pl.h_slab_bytes; a.h_nr = pl.h_nr; a.h_mtiles = pl.h_mtiles;
    a.h_dbuf = pl.h_dbuf;
    a.h_abl = tri_probe_ablation();
    a.dP = make_fastdiv(a.IW + 2); a.dH2 = make_fastdiv(a.IH + 2);
    a.h_swz = halo_swizzle(a.IW, a.IH, pl.halo);
#ifdef HALO_STAMPS
    a.h_dbg = g_halo_dbg;
    const size_t smem = (size_t)pl.h_slab_bytes + (size_t)HALO_NR * 8192 + 2048 + 2048;
#else
    const size_t smem = (size_t)pl.h_slab_bytes + (size_t)HALO_NR * 8192 + 2048;
#endif
    a.h_xcg = 0; a.h_touch = 0;
    if (pl.h_v5) {
        // XCD blocks + L2 warm-up of the weights (one tile per workgroup, a whole number of workgroups per XCD): the channel-group width cg that
        // moves the fewest bytes into the eight L2s - weights x (row groups = 8 cg / NT) + input rows x (channel groups = NT / cg)
        // (TRICOLO_HALO_XCG=0 keeps the linear tile order, TRICOLO_HALO_TOUCH=0 the cold start: A/B partners)
        static int xcg_on = -1, touch_on = -1;
        if (xcg_on < 0) { const char* e = getenv("TRICOLO_HALO_XCG"); xcg_on = e ? atoi(e) : -1; if (xcg_on < 0) xcg_on = 99; }
        if (touch_on < 0) { const char* e = getenv("TRICOLO_HALO_TOUCH"); touch_on = (e && e[0] == '0') ? 0 : 1; }
        const int NT = a.Cout / 64, G = pl.h_grid;
        if (xcg_on && G == pl.h_mtiles * NT && G % 8 == 0 && a.Cin != 64) {
            const double wb = (double)a.Cout * a.Kpad * 2, ib = (double)a.M * a.Cin * 2;
            double best = 0;
            for (int cg = 1; cg <= NT; ++cg) {
                if (NT % cg || 8 % (NT / cg) || (G / 8) % cg) continue;
                if (xcg_on != 99 && cg != xcg_on) continue;
                const double bytes = wb * (8.0 * cg / NT) + ib * (NT / cg);
                if (!a.h_xcg || bytes < best) { a.h_xcg = cg; best = bytes; }
            }
            a.h_touch = a.h_xcg ? touch_on : 0;
        }
#ifdef HALO_STAMPS
        a.h_dbg = g_halo_dbg;
        const size_t extra = 2048;
#else
        const size_t extra = 0;
#endif
        const bool drip = 2 * (size_t)pl.h_slab_bytes + 3 * (size_t)HROWS_SLOT + 16384 + extra <= 163840 && !a.accumulate && pl.halo != 3;
        const size_t smem5 = 2 * (size_t)pl.h_slab_bytes + 3 * (size_t)HROWS_SLOT + (drip ? 16384 : 8192) + extra;
        // weights: resident filter bank (64 input channels: NTL 1) or streamed (NTL 0).  The kernel also runs groups of 2 / 4 tiles per
        // resident chunk (NTL 2 / 4, TRICOLO_HALO_NTL): measured no better than streaming on any configuration (profiles/r2/NOTES), so
        // those instantiations are only built with -DHALO_NTL_EXPERIMENT
        int ntl = a.Cin == 64 ? 1 : 0;
#ifdef HALO_NTL_EXPERIMENT
        { static int f = -1; if (f < 0) f = 0;
          const int items = pl.h_mtiles * (a.Cout / 64), per_wg = (items + pl.h_grid - 1) / pl.h_grid;
          if (f && a.Cin != 64 && per_wg > 1) ntl = per_wg == 2 ? 2 : 4; }
#endif
#define TRI_ROWS_LAUNCH(DRIP_, ACC_, NTL_, ...)                                                                                     \
        do {                                                                                                                        \
            static size_t attr5 = 0;                                                                                                \
            if (smem5 > attr5) {                                                                                                    \
                hipFuncSetAttribute((const void*)conv_halo_rows_kernel<AT, DRIP_, ACC_, NTL_, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)smem5);                                                                                    \
                attr5 = smem5;                                                                                                      \
            }                                                                                                                       \
            conv_halo_rows_kernel<AT, DRIP_, ACC_, NTL_, ##__VA_ARGS__><<<pl.h_grid, 256, smem5, stream>>>(a);                       \
        } while (0)
#define TRI_ROWS_MODE(NTL_)                                                                                                         \
        do {                                                                                                                        \
            if (a.accumulate) TRI_ROWS_LAUNCH(false, true, NTL_);                                                                   \
            else if (drip) TRI_ROWS_LAUNCH(true, false, NTL_);                                                                      \
            else TRI_ROWS_LAUNCH(false, false, NTL_);                                                                               \
        } while (0)
        // producer waves for the streamed 128-position tiles (layer4 of the bench shape: 23.0 -> 20.5 us forward, 21.8 -> 19.1 us data gradient;
        // config 3 3.771 -> 3.748 ms, three of three alternating pairs); TRICOLO_HALO_PROD=0 is the A/B partner
        static int prod = -1;
        if (prod < 0) { const char* e = getenv("TRICOLO_HALO_PROD"); prod = (e && e[0] == '0') ? 0 : 1; }
        if (prod && pl.halo == 2 && ntl == 0) {
#define TRI_ROWS_LAUNCH_P(DRIP_, ACC_)                                                                                              \
        do {                                                                                                                        \
            static size_t attrp = 0;                                                                                                \
            if (smem5 > attrp) {                                                                                                    \
                hipFuncSetAttribute((const void*)conv_halo_rows_kernel<AT, DRIP_, ACC_, 0, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)smem5);                                                                                    \
                attrp = smem5;                                                                                                      \
            }                                                                                                                       \
            conv_halo_rows_kernel<AT, DRIP_, ACC_, 0, 2, true><<<pl.h_grid, 512, smem5, stream>>>(a);                                \
        } while (0)
            if (a.accumulate) TRI_ROWS_LAUNCH_P(false, true);
            else if (drip) TRI_ROWS_LAUNCH_P(true, false);
            else TRI_ROWS_LAUNCH_P(false, false);
#undef TRI_ROWS_LAUNCH_P
            return tri_check_launch("tri_conv(halo rows, producer waves)");
        }
        if (pl.halo == 3) {                                           // 192-position tiles (streamed weights, epilogue stores)
            if (a.accumulate) TRI_ROWS_LAUNCH(false, true, 0, 3);
            else TRI_ROWS_LAUNCH(false, false, 0, 3);
        }
        else if (ntl == 0) TRI_ROWS_MODE(0);
        else if (ntl == 1) TRI_ROWS_MODE(1);
#ifdef HALO_NTL_EXPERIMENT
        else if (ntl == 2) TRI_ROWS_MODE(2);
        else TRI_ROWS_MODE(4);
#endif
#undef TRI_ROWS_MODE
#undef TRI_ROWS_LAUNCH
        return tri_check_launch("tri_conv(halo rows)");
    }
    static size_t attr = 0;
    if (smem > attr) {
        hipFuncSetAttribute((const void*)conv_halo2d_kernel<TM, false, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute((const void*)conv_halo2d_kernel<TM, true, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr = smem;
    }
    if (pl.h_wgrec && a.stats) conv_halo2d_kernel<TM, true, AT><<<pl.h_grid, 256, smem, stream>>>(a);   // persistent workgroups
    else conv_halo2d_kernel<TM, false, AT><<<pl.h_grid, 256, smem, stream>>>(a);
    return tri_check_launch("tri_conv(halo)");
}

template <typename AT>
static int launch_stem(ConvArgs& a, hipStream_t stream) {
    StemGeom sg;
    a.h_abl = tri_probe_ablation();
    if (!stem_geometry(a.B, a.IH, a.IW, a.Cin, a.OH, a.OW, a.Cout, a.KD, a.KH, a.KW, a.stride, a.pd, a.ph, a.pw, &sg)) {
        tri_set_error("conv(stem): geometry changed between plan and launch"); return TRI_ERR_ARG;
    }
    const size_t smem = (size_t)2 * sg.slab_rows * sg.row_bytes + 4 * 64 * 2 * sizeof(float);
#define TRI_STEM(KH_)                                                                                                     \
    case KH_: {                                                                                                           \
        static bool attr = false;                                                                                         \
        if (!attr) {                                                                                                      \
            hipFuncSetAttribute((const void*)conv_stem_kernel<KH_, 2, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
            hipFuncSetAttribute((const void*)conv_stem_kernel<KH_, 4, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
            attr = true;                                                                                                  \
        }                                                                                                                 \
        if (sg.npt == 4) conv_stem_kernel<KH_, 4, AT><<<sg.grid, 256, smem, stream>>>(a, sg);                             \
        else conv_stem_kernel<KH_, 2, AT><<<sg.grid, 256, smem, stream>>>(a, sg);                                         \
        break;                                                                                                            \
    }
    switch (a.KH) {
        TRI_STEM(7) TRI_STEM(3) TRI_STEM(5)
        default: tri_set_error("conv(stem): kernel height not instantiated (3, 5, 7)"); return TRI_ERR_UNSUPPORTED;
    }
#undef TRI_STEM
    return tri_check_launch("tri_conv(stem)");
}

template <typename AT>
static int launch_dma_any(const ConvArgs& a, int bn, hipStream_t stream) {
    const int nst = dma_stages_for(a, bn);
    if (bn == 32) return nst == 2 ? launch_dma<32, 2, AT>(a, stream) : launch_dma<32, 3, AT>(a, stream);
    switch (nst) {
        case 4: return bn == 128 ? launch_dma<128, 4, AT>(a, stream) : launch_dma<64, 4, AT>(a, stream);
        case 3: return bn == 128 ? launch_dma<128, 3, AT>(a, stream) : launch_dma<64, 3, AT>(a, stream);
        default: return bn == 128 ? launch_dma<128, 2, AT>(a, stream) : launch_dma<64, 2, AT>(a, stream);
    }
}

static int conv_dispatch(ConvArgs& a, int act_fmt, void* workspace, size_t workspace_bytes, hipStream_t stream, const TriConvBnSums* bs = nullptr) {
    if (a.Cin % 4 != 0) { tri_set_error("conv: stored input channels must be a multiple of 4"); return TRI_ERR_ARG; }
    if (a.Cout % 32 != 0) { tri_set_error("conv: output channels must be a multiple of 32"); return TRI_ERR_ARG; }
    if (a.ntaps > 64) { tri_set_error("conv: more than 64 taps unsupported"); return TRI_ERR_UNSUPPORTED; }
    if (a.stride != 1 && a.stride != 2) { tri_set_error("conv: stride must be 1 or 2"); return TRI_ERR_UNSUPPORTED; }
    if (a.KD > 8 || a.KH > 8 || a.KW > 8) { tri_set_error("conv: kernel extent > 8 unsupported"); return TRI_ERR_UNSUPPORTED; }
    if (act_fmt < 0 || act_fmt > 2) { tri_set_error("conv: act_fmt must be 0 (fp32), 1 (bf16) or 2 (f16)"); return TRI_ERR_ARG; }
    a.Kpad = (a.ntaps * a.Cin + 31) / 32 * 32;
    a.cin_shift = ilog2_exact(a.Cin);
    size_t in_bytes = (size_t)a.B * a.ID * a.IH * a.IW * a.Cin * (act_fmt ? 2 : 4);
    if (in_bytes >= ((size_t)1 << 31)) { tri_set_error("conv: input tensor >= 2 GiB (32-bit buffer offsets)"); return TRI_ERR_UNSUPPORTED; }
    a.in_bytes = (unsigned)in_bytes;
    const bool split = a.w_lo != nullptr;
    if (act_fmt && split) { tri_set_error("conv: 16-bit activation storage takes single operands (no lo part)"); return TRI_ERR_ARG; }
    ConvPlan pl = conv_make_plan(a.B, a.ID, a.IH, a.IW, a.Cin, a.OD, a.OH, a.OW, a.Cout, a.KD, a.KH, a.KW, a.stride, a.pd, a.ph, a.pw, act_fmt ? 2 : (split ? 1 : 0),
                                 a.row_count ? 1 : 0);
    a.ksplit = pl.ksplit;
    a.steps_per_split = pl.per_split;
    a.nunits = pl.nunits;
    if (a.ksplit > 1) {
        size_t need = (size_t)a.ksplit * a.M * a.Cout * sizeof(float);
        if (workspace == nullptr || workspace_bytes < need) {
            tri_set_error("conv: this layer runs split-K; pass tri_conv_workspace() bytes of scratch");
            return TRI_ERR_ARG;
        }
        a.slab = (float*)workspace;
    }
    a.dOW = make_fastdiv(a.OW); a.dOH = make_fastdiv(a.OH); a.dOD = make_fastdiv(a.OD); a.dCin = make_fastdiv(a.Cin);
    if (a.row_count) {
        // compact row list (row_pos[0 .. *row_count) are the rows to compute): not an order hint, it changes WHICH rows run, so
        // the kernel must honour it - any layer without split-K does (the split-K finish kernel walks positions, not the list)
        if (!a.row_pos) { tri_set_error("conv: row_count needs row_pos"); return TRI_ERR_ARG; }
        if (a.row_mask) { tri_set_error("conv: pass either row_mask or a compact row list"); return TRI_ERR_ARG; }
        if (pl.dma && a.Kpad / 64 > 256) {
            tri_set_error("conv: more than 256 k-steps with a row list (the live-step table of conv_dma_kernel)");
            return TRI_ERR_UNSUPPORTED;
        }
    } else if (a.row_pos && !(pl.dma && pl.ksplit == 1 && !a.row_mask && !a.stats && a.Kpad / 64 <= 256)) {
        a.row_pos = nullptr;          // a pure visiting-order hint: honoured by the DMA kernel without split-K, dropped elsewhere
    }
    if (pl.vox0 && !a.transposed) {
        // the brick kernel walks the dense grid by the SITE MASK (rows of inactive sites are neither computed nor written); its
        // BatchNorm records are one per workgroup, which is what tri_conv_num_mtiles reports for this layer
        if (a.row_count || a.bias || a.act != 0 || a.accumulate) {
            tri_set_error("conv: this layer runs the brick kernel (tri_conv_kernel_family == 6): pass the site mask as row_mask, no row list, "
                          "bias, activation or accumulate");
            return TRI_ERR_ARG;
        }
        TriVox0Geom vg;
        tri_internal_vox0_geometry(a.B, a.ID, a.IH, a.IW, a.Cin, a.OD, a.OH, a.OW, a.Cout, a.KD, a.KH, a.KW, a.stride, a.pd, a.ph, a.pw, &vg);
        return tri_internal_vox0_launch(vg, a.B, a.in, a.w_hi, a.Kpad, a.out, a.row_mask, a.stats, act_fmt, stream);
    }
    if (pl.voxb && !a.transposed) {
        if (a.row_count || a.bias || a.act != 0 || a.accumulate) {
            tri_set_error("conv: this layer runs conv_voxb_kernel (tri_conv_kernel_family == 14): pass the site mask as row_mask, no row list, "
                          "bias, activation or accumulate");
            return TRI_ERR_ARG;
        }
        TriVoxbGeom vb;
        tri_internal_voxb_geometry(a.B, a.ID, a.IH, a.IW, a.Cin, a.OD, a.OH, a.OW, a.Cout, a.KD, a.KH, a.KW, a.stride, a.pd, a.ph, a.pw, &vb);
        return tri_internal_voxb_launch(vb, a.B, a.in, a.w_hi, a.out, a.row_mask, a.stats, act_fmt, stream);
    }
    if (pl.voxg) {
        if (a.row_count || a.bias || a.act != 0 || a.accumulate) {
            tri_set_error("conv: this layer runs conv_voxg_kernel (tri_conv_kernel_family == 13): pass the site mask as row_mask, no row list, "
                          "bias, activation or accumulate");
            return TRI_ERR_ARG;
        }
        TriVoxgGeom vgg;
        tri_internal_voxg_geometry(a.B, a.ID, a.IH, a.IW, a.Cin, a.OD, a.OH, a.OW, a.Cout, a.KD, a.KH, a.KW, a.stride, a.pd, a.ph, a.pw, &vgg);
        return tri_internal_voxg_launch(vgg, a.B, a.Cin, a.Cout, a.Kpad, a.in, a.w_hi, a.out, a.row_mask, a.stats, a.transposed, act_fmt, stream);
    }
    if (pl.c64) {
        if (!a.row_mask && !a.row_count && !a.bias && a.act == 0) {
            TriC64Geom cg;
            tri_internal_c64_geometry(a.B, a.ID, a.IH, a.IW, a.Cin, a.OD, a.OH, a.OW, a.Cout, a.KD, a.KH, a.KW, a.stride, a.pd, a.ph, a.pw, &cg);
            return tri_internal_c64_launch(cg, a.B, a.IH, a.in, a.w_hi, a.out, bs ? bs->partial : a.stats, a.transposed, a.accumulate, act_fmt, bs, stream);
        }
        if (a.stats) {                                                // (the record count of this layer is conv_c64_kernel's)
            tri_set_error("conv: this layer runs conv_c64_kernel (tri_conv_kernel_family == 9): statistics only without row mask / bias / activation");
            return TRI_ERR_ARG;
        }
        a.row_pos = a.row_count ? a.row_pos : nullptr;
    }
    if (pl.s2g) {
        if (a.transposed || a.row_mask || a.row_count || a.bias || a.act != 0 || a.accumulate) {
            tri_set_error("conv: this layer runs conv_s2g_kernel (tri_conv_kernel_family == 15): forward only, no row mask / list, bias, activation "
                          "or accumulate");
            return TRI_ERR_ARG;
        }
        TriS2gGeom g2;
        tri_internal_s2g_geometry(a.B, a.ID, a.IH, a.IW, a.Cin, a.OD, a.OH, a.OW, a.Cout, a.KD, a.KH, a.KW, a.stride, a.pd, a.ph, a.pw, &g2);
        return tri_internal_s2g_launch(g2, a.B, a.IH, a.IW, a.Cin, a.Cout, a.Kpad, a.in, a.w_hi, a.out, a.stats, act_fmt, stream);
    }
    if (pl.s2d && a.transposed && !bs && !a.row_mask && !a.row_count && !a.bias && a.act == 0 && !a.stats) {
        TriC64Geom sg;
        tri_internal_s2d_geometry(a.B, a.ID, a.IH, a.IW, a.Cin, a.OD, a.OH, a.OW, a.Cout, a.KD, a.KH, a.KW, a.stride, a.pd, a.ph, a.pw, &sg);
        return tri_internal_s2d_launch(sg, a.B, a.IH, a.in, a.w_hi, a.out, a.accumulate, act_fmt, stream);
    }
    if (bs) {
        tri_set_error("conv: this layer's data-gradient kernel takes no BatchNorm-backward sums (tri_conv_dgrad_bn_records == 0)");
        return TRI_ERR_UNSUPPORTED;
    }
    if (pl.stem && !a.transposed && !a.row_mask && !a.row_count && !a.bias && a.act == 0 && !a.accumulate) {
        a.row_pos = nullptr;
        return act_fmt == TRI_FMT_F16 ? launch_stem<f16_t>(a, stream) : launch_stem<bf16_t>(a, stream);
    }
    if (pl.halo && !a.row_mask && !a.row_count && !a.bias && a.act == 0) {
        a.row_pos = nullptr;
        return act_fmt == TRI_FMT_F16 ? launch_halo<2, f16_t>(a, pl, stream) : launch_halo<2, bf16_t>(a, pl, stream);
    }
    if (pl.pw && pl.pw_tr == (a.transposed ? 1 : 0) && !a.row_mask && !a.row_count && !a.bias && a.act == 0 && !a.accumulate) {
        TriPwGeom pg;
        tri_internal_pw_geometry(a.B, a.ID, a.IH, a.IW, a.Cin, a.OD, a.OH, a.OW, a.Cout, a.KD, a.KH, a.KW, a.stride, a.pd, a.ph, a.pw, &pg);
        return tri_internal_pw_launch(pg, a.B, a.in, a.w_hi, a.Cin, a.Cout, a.Kpad, a.out, a.stats, a.transposed ? 1 : 0, act_fmt, stream);
    }
    if (pl.dma) return act_fmt == TRI_FMT_F16 ? launch_dma_any<f16_t>(a, pl.bn, stream) : launch_dma_any<bf16_t>(a, pl.bn, stream);
#define TRI_CONV(BN_)                                                                                     \
    (act_fmt == TRI_FMT_F16 ? launch_conv<BN_, 1, f16_t>(a, stream)                                       \
     : act_fmt == TRI_FMT_BF16 ? launch_conv<BN_, 1, bf16_t>(a, stream)                                   \
     : (split ? launch_conv<BN_, 2, float>(a, stream) : launch_conv<BN_, 1, float>(a, stream)))
    if (a.Cout % 128 == 0 && pl.bn != 64) return TRI_CONV(128);
    if (a.Cout % 64 == 0) return TRI_CONV(64);
    return TRI_CONV(32);
#undef TRI_CONV
}

extern "C" int tri_conv_kpad(int ntaps, int cin_stored) { return (ntaps * cin_stored + 31) / 32 * 32; }

// number of [2][Cout] statistic records tri_conv_fwd writes for this layer: one per 128-row tile, or one per 32-row
// chunk when the layer runs split-K (the finish kernel produces them).  tri_bn_finalize just sums all records.
// split3: 0 bf16 operands / fp32 activations, 1 bf16x3 (hi + lo operands), 2 16-bit operands and activation storage.
extern "C" int tri_conv_num_records(const TriConvDesc* d, int split3, int row_list) {
    long M = (long)d->B * d->OD * d->OH * d->OW;
    ConvPlan pl = conv_make_plan(d->B, d->ID, d->IH, d->IW, d->Cin, d->OD, d->OH, d->OW, d->Cout, d->KD, d->KH, d->KW, d->stride, d->pad_d,
                                 d->pad_h, d->pad_w, split3, row_list);
    if (pl.stem) return pl.stem_grid;
    if (pl.vox0) return pl.vox0_grid;
    if (pl.voxb) return pl.voxb_grid;
    if (pl.voxg) return pl.voxg_units;
    if (pl.c64) return pl.c64_grid;
    if (pl.s2g) return pl.s2g_units;
    if (pl.halo) return pl.h_wgrec ? pl.h_grid / (d->Cout / 64) : pl.h_mtiles;
    return pl.ksplit > 1 ? (int)((M + 31) / 32) : (int)((M + 127) / 128);
}
extern "C" int tri_conv_num_mtiles(const TriConvDesc* d, int split3) { return tri_conv_num_records(d, split3, 0); }

// kernel family the dispatch picks (for profilers): low byte 0 conv_igemm_kernel / 2 conv_dma_kernel, bits 8.. = channel tile
extern "C" int tri_conv_kernel_family(const TriConvDesc* d, int transposed, int split3) {
    ConvPlan pl = transposed ? conv_make_plan(d->B, d->OD, d->OH, d->OW, d->Cout, d->ID, d->IH, d->IW, d->Cin, d->KD, d->KH, d->KW, d->stride,
                                              d->pad_d, d->pad_h, d->pad_w, split3)
                             : conv_make_plan(d->B, d->ID, d->IH, d->IW, d->Cin, d->OD, d->OH, d->OW, d->Cout, d->KD, d->KH, d->KW, d->stride,
                                              d->pad_d, d->pad_h, d->pad_w, split3);
    if (pl.stem && !transposed) return 4 | (64 << 8);
    if (pl.vox0 && !transposed) return 6 | (32 << 8);
    if (pl.voxb && !transposed) return 14 | (64 << 8);
    if (pl.voxg) return 13 | (pl.voxg_ct << 8) | (pl.voxg_spu << 24);      // (bits 24..: samples per unit)
    if (pl.c64) return 9 | (64 << 8);
    if (pl.s2d && transposed) return 10 | (64 << 8);
    if (pl.s2g && !transposed) return 15 | (64 << 8);
    if (pl.halo) return (pl.h_v5 ? 5 : 3) | (pl.halo << 8);
    if (pl.pw) return 12 | (pl.bn << 8);
    return (pl.dma ? 2 : 0) | (pl.bn << 8) | ((pl.ksplit > 1 || (pl.dma && tri_conv_kpad(d->KD * d->KH * d->KW, transposed ? d->Cout : d->Cin) / 64 > 256)) ? (1 << 16) : 0);
}

// out[B,OD,OH,OW,Cout] = conv(in[B,ID,IH,IW,Cin], W) (+bias, act 0 none / 1 relu / 2 tanh); rows with row_mask==0 are
// written as zeros (submanifold rule); stats != NULL receives per-128-row-tile column sums and sums of squares.
// bytes of split-K scratch tri_conv_fwd (transposed = 0) / tri_conv_dgrad (transposed = 1) can use for this layer
// (0 when the layer already fills the GPU) and therefore REQUIRES.
extern "C" size_t tri_conv_workspace(const TriConvDesc* d, int transposed) {
    size_t need = 0;
    for (int mode = 0; mode < 6; ++mode) {                                   // three operand modes x (dense | compact row list)
        ConvPlan pl;
        long M;
        int cout;
        const int rl = mode / 3;
        if (rl && (d->KD != 3 || d->stride != 1)) continue;                  // row lists are a voxel-level (3D submanifold) thing
        if (transposed) {
            M = (long)d->B * d->ID * d->IH * d->IW; cout = d->Cin;
            pl = conv_make_plan(d->B, d->OD, d->OH, d->OW, d->Cout, d->ID, d->IH, d->IW, d->Cin, d->KD, d->KH, d->KW, d->stride, d->pad_d,
                                d->pad_h, d->pad_w, mode % 3, rl);
        } else {
            M = (long)d->B * d->OD * d->OH * d->OW; cout = d->Cout;
            pl = conv_make_plan(d->B, d->ID, d->IH, d->IW, d->Cin, d->OD, d->OH, d->OW, d->Cout, d->KD, d->KH, d->KW, d->stride, d->pad_d,
                                d->pad_h, d->pad_w, mode % 3, rl);
        }
        size_t n = pl.ksplit > 1 ? (size_t)pl.ksplit * M * cout * sizeof(float) : 0;
        if (n > need) need = n;
    }
    return need;
}

extern "C" int tri_conv_fwd(const TriConvDesc* d, const void* in, const void* w_hi, const void* w_lo, void* out,
                            const uint8_t* row_mask, const float* bias, int act, int accumulate, float* stats, int act_fmt,
                            void* workspace, size_t workspace_bytes, const int* row_pos, const int* row_count, void* stream) {
    ConvArgs a{};
    a.in = in; a.w_hi = w_hi; a.w_lo = w_lo; a.out = out;
    a.row_mask = row_mask; a.bias = bias; a.stats = stats; a.row_pos = row_pos; a.row_count = row_count;
    a.B = d->B; a.ID = d->ID; a.IH = d->IH; a.IW = d->IW; a.Cin = d->Cin;
    a.OD = d->OD; a.OH = d->OH; a.OW = d->OW; a.Cout = d->Cout;
    a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pd = d->pad_d; a.ph = d->pad_h; a.pw = d->pad_w;
    a.transposed = 0; a.act = act; a.accumulate = accumulate;
    a.ntaps = d->KD * d->KH * d->KW;
    a.M = d->B * d->OD * d->OH * d->OW;
    return conv_dispatch(a, act_fmt, workspace, workspace_bytes, (hipStream_t)stream);
}

// din[B,ID,IH,IW,Cin] (+)= conv_transpose(dout[B,OD,OH,OW,Cout], Wt), Wt packed [Cin][taps*Cout] by tri_weight_prep
// with swapped strides.  `d` is the FORWARD descriptor of the layer.
extern "C" int tri_conv_dgrad(const TriConvDesc* d, const void* dout, const void* wt_hi, const void* wt_lo, void* din,
                              const uint8_t* row_mask, int accumulate, int act_fmt, void* workspace, size_t workspace_bytes,
                              const int* row_pos, const int* row_count, void* stream) {
    ConvArgs a{};
    a.in = dout; a.w_hi = wt_hi; a.w_lo = wt_lo; a.out = din;
    a.row_mask = row_mask; a.bias = nullptr; a.stats = nullptr; a.row_pos = row_pos; a.row_count = row_count;
    a.B = d->B; a.ID = d->OD; a.IH = d->OH; a.IW = d->OW; a.Cin = d->Cout;       // gather source = dout grid
    a.OD = d->ID; a.OH = d->IH; a.OW = d->IW; a.Cout = d->Cin;                   // rows = input positions
    a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pd = d->pad_d; a.ph = d->pad_h; a.pw = d->pad_w;
    a.transposed = 1; a.act = 0; a.accumulate = accumulate;
    a.ntaps = d->KD * d->KH * d->KW;
    a.M = d->B * d->ID * d->IH * d->IW;
    return conv_dispatch(a, act_fmt, workspace, workspace_bytes, (hipStream_t)stream);
}

// Data gradient + the BatchNorm-backward sums of the pass that would read din next (include/tricolo_hip.h, TriConvBnSums)
extern "C" int tri_conv_dgrad_bn_records(const TriConvDesc* d, int accumulate, int act_fmt) {
    (void)accumulate;
    if (act_fmt != TRI_FMT_BF16 && act_fmt != TRI_FMT_F16) return 0;
    ConvPlan pl = conv_make_plan(d->B, d->OD, d->OH, d->OW, d->Cout, d->ID, d->IH, d->IW, d->Cin, d->KD, d->KH, d->KW, d->stride, d->pad_d,
                                 d->pad_h, d->pad_w, 2, 0);
    if (pl.c64 && d->IW <= 32) return pl.c64_grid;              // (conv_c64.hip: the sums forms exist for 16- and 32-wide images)
    return 0;
}
extern "C" int tri_conv_dgrad_bn(const TriConvDesc* d, const void* dout, const void* wt_hi, const void* wt_lo, void* din, int accumulate,
                                 int act_fmt, void* workspace, size_t workspace_bytes, const int* row_pos, const TriConvBnSums* sums,
                                 void* stream) {
    if (!sums || !sums->y || !sums->partial || (!sums->relu_scale != !sums->relu_shift) || (sums->relu_scale && sums->relu_out)) {
        tri_set_error("tri_conv_dgrad_bn: sums needs y, partial and at most one of (relu_scale + relu_shift) / relu_out");
        return TRI_ERR_ARG;
    }
    if (tri_conv_dgrad_bn_records(d, accumulate, act_fmt) == 0) {
        tri_set_error("tri_conv_dgrad_bn: no fused form for this layer (tri_conv_dgrad_bn_records == 0)");
        return TRI_ERR_UNSUPPORTED;
    }
    ConvArgs a{};
    a.in = dout; a.w_hi = wt_hi; a.w_lo = wt_lo; a.out = din;
    a.row_pos = row_pos;
    a.B = d->B; a.ID = d->OD; a.IH = d->OH; a.IW = d->OW; a.Cin = d->Cout;
    a.OD = d->ID; a.OH = d->IH; a.OW = d->IW; a.Cout = d->Cin;
    a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pd = d->pad_d; a.ph = d->pad_h; a.pw = d->pad_w;
    a.transposed = 1; a.act = 0; a.accumulate = accumulate;
    a.ntaps = d->KD * d->KH * d->KW;
    a.M = d->B * d->ID * d->IH * d->IW;
    return conv_dispatch(a, act_fmt, workspace, workspace_bytes, (hipStream_t)stream, sums);
}
