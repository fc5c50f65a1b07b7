// 64 -> 64 channel 3x3 / 1 / pad 1 convolution with the filter bank STATIONARY IN REGISTERS (16-bit storage modes, gfx950): layer1 of
// the ResNet trunk (mv_cnn.py:44, torchvision BasicBlock 64 -> 64), forward and data gradient.
//
// conv_halo_rows_kernel keeps layer1's filter bank in LDS and every wave re-reads four weight fragments per k-step next to its two
// activation fragments; at one wave per SIMD (372 registers) nothing covers the slab DMA pieces, the epilogue and the 7.4 k-cycle
// prologue: a 128-position tile takes 5.4 k cycles for 2.3 k cycles of MFMA.  This kernel is the 2D twin of round 3's conv_vox1_kernel (voxel level 1; dropped in round 6 for conv_voxb_kernel):
//   * a persistent workgroup (two per CU) walks bricks of TY image rows x W columns = 8 runs of 16 pixels, stages a brick + one-pixel halo
//     once in an LDS slab (128 B per pixel, 16-byte chunks XOR-swizzled by the pixel pair: conflict-free ds_read_b128 for all three kx);
//   * wave (c, h) holds the A fragments of output channels 32 c .. 32 c + 31 for all 9 taps x 2 k-steps (36 fragments, 144 registers,
//     loaded once per workgroup) and takes the runs of parity h: every activation fragment it reads feeds TWO MFMAs (the 1 : 1 form of
//     conv_vox1_kernel is LDS-bound at half the matrix rate), no weight traffic at all after the prologue;
//   * fragment addresses: per-lane offsets for (kx, k-step), immediates / one add for (ky, run);
//   * forward: BatchNorm sums of the values as stored stay in registers over all bricks, one record per workgroup; data gradient: the
//     same correlation over the transposed operand rows with the taps in reverse order, optionally added to what `out` holds (the
//     shortcut's gradient), fp32 sum rounded once.
#include "common.h"
#include <stdlib.h>
#include "../../include/tricolo_hip.h"
#include "conv_vox.h"

int tri_internal_num_cus();                                                   // conv_igemm.hip

template <int N>
__device__ __forceinline__ float c64_row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}

struct ConvC64Args {
    const void* in;            // [N, H, W, 64] 16-bit
    const void* w;             // packed operand rows [64][576] (k = tap * 64 + channel); data gradient: the transposed rows
    void* out;                 // [N, H, W, 64]
    float* stats;              // [grid][2][64] or NULL
    int N, H, nbricks, transposed, accumulate;
    unsigned in_bytes;
    // BatchNorm-backward sums of the NEXT pass over `out` taken here instead (data gradient; TriConvBnSums): the records then hold
    // sum g', sum g' * y per channel with g' = the stored value where the ReLU mask passes (mode 1: y * scale + shift > 0, mode 2: ro > 0)
    const void* bs_y;          // [N, H, W, 64] the BatchNorm's input, same storage as out
    const void* bs_ro;         // mode 2: the saved block output
    const float* bs_scale;     // mode 1: the forward's folded scale / shift
    const float* bs_shift;
    int bs_mode;               // 0 none (forward statistics: sum v, sum v^2)
};

template <int W, int TY>
struct C64Cfg {
    static constexpr int XOFF = 4;                                             // pad pixels left of a slab row
    static constexpr int RPR = (W + 15) / 16;                                  // runs per image row (56-wide rows: the last run is half used)
    static constexpr int P = RPR * 16 + 16;                                    // pixels per slab row: a multiple of 16 (the swizzle then depends on x only)
    static constexpr int PITCH = P * 128;
    static constexpr int SLAB = (TY + 2) * PITCH;
    static constexpr int RUNS = TY * RPR;
    static constexpr int CPR = W * 8;                                          // 16-byte chunks per image row
    static constexpr int ITEMS = (TY + 2) * CPR;
    static constexpr int MAXC = (ITEMS + 255) / 256;
    static constexpr int WROW = 9 * 64 * 2 + 16;                               // filter-bank row pitch while it is staged through LDS: 16 rows -> 16 distinct bank quads
    static constexpr size_t SMEM = (size_t)SLAB > (size_t)64 * WROW ? (size_t)SLAB : (size_t)64 * WROW;
    static_assert(RUNS == 8 || RUNS == 4, "a brick is 8 or 4 runs (half of them per wave pair)");
    static_assert(2 * PITCH + 4096 < 65536, "fragment-read immediates");
};

template <typename AT, int W, int TY, bool ACCUM, bool BS>
__global__ __launch_bounds__(256, 2) void conv_c64_kernel(const ConvC64Args p) {
    typedef C64Cfg<W, TY> C;
    typedef typename OpOf<AT>::E E;
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    constexpr int KPAD = 9 * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
    const int c = wave & 1, h = wave >> 1;
    char* const slab = smem;
    const int G = gridDim.x, wg = blockIdx.x;
    const int bpi = p.H / TY;                                                  // bricks per image

    // ---- filter bank of this wave's 32 output channels: A fragment (tap, k-step, ct): row fr = channel 32 c + 16 ct + fr,
    // k = 8 fq .. 8 fq + 7 of the k-step's 32 channels; the data gradient walks the taps backwards (tap' = 8 - tap)
    // (staged through LDS with coalesced loads: read straight from global memory these are 36 scattered 16-byte loads per lane, and with
    // every workgroup of the launch doing it at once the launch paid ~10 us before its first MFMA - the lesson of conv_vox0_kernel)
    v8 wf[9][2][2];
    {
        constexpr int NCH = 64 * KPAD * 2 / 16;                                // 4,608 chunks of the [64][576] bank
        uint4 wld[NCH / 256];
#pragma unroll
        for (int u = 0; u < NCH / 256; ++u) wld[u] = *(const uint4*)((const char*)p.w + (size_t)(t + u * 256) * 16);
#pragma unroll
        for (int u = 0; u < NCH / 256; ++u) {
            const int ch = t + u * 256, row = ch / 72, col = ch - row * 72;    // 72 chunks per row
            *(uint4*)(slab + row * C::WROW + col * 16) = wld[u];
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const char* wrow = slab + (32 * c + 16 * ct + fr) * C::WROW + fq * 16;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int tw = p.transposed ? 8 - tap : tap;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) wf[tap][ks][ct] = *(const v8*)(wrow + (tw * 64 + ks * 32) * 2);
            }
        }
        __syncthreads();                                                       // the slab takes the staging area over
    }
    // per-lane slab offsets for (kx, k-step): pixel XOFF + fr + kx - 1, chunk 4 ks + fq at its swizzled position
    int lofs[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int sx = C::XOFF + fr + kx - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) lofs[kx][ks] = sx * 128 + (((4 * ks + fq) ^ ((sx >> 1) & 7)) << 4);
    }
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    f32x4 cs[2], cq[2];
    cs[0] = cs[1] = cq[0] = cq[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int bsm = BS ? (ACCUM ? 2 : 1) : 0;                              // (the two forms the BasicBlock backward has: conv2's data
                                                                               // gradient feeds relu(bn1), conv1's accumulated one relu(bn2 + x))
    float* const bsc = (float*)(smem + C::SMEM);                               // mode 1: [2][64] scale / shift (kept out of the register file)
    if (bsm == 1 && t < 128) bsc[t] = t < 64 ? p.bs_scale[t] : p.bs_shift[t - 64];
    // the zero pixel left and right of every slab row (written once: the fill below never touches them)
    constexpr int NZ = C::RPR * 16 - W + 2;                                    // zero pixels per slab row: x = -1 and x = W .. 16 RPR (the half-used run reads them)
    for (int i = t; i < (TY + 2) * NZ * 8; i += 256) {
        const int row = i / (NZ * 8), rem = i - row * (NZ * 8), zp = rem >> 3, q = rem & 7;
        const int sx = zp == 0 ? C::XOFF - 1 : C::XOFF + W + zp - 1;
        *(uint4*)(slab + row * C::PITCH + sx * 128 + q * 16) = make_uint4(0u, 0u, 0u, 0u);
    }

#pragma unroll 1
    for (int j = wg; j < p.nbricks; j += G) {
        const int n = j / bpi, y0 = (j - n * bpi) * TY;
        // sums forms: the brick's y (and saved-output) tile is touched now, one 128-byte pixel per thread, so that the epilogue's loads -
        // issued only under the last kernel row's MFMAs, the register file is full - find it in L2 instead of HBM (in the step y was
        // written a whole forward ago: the exposed latency made the fused form as slow as the reduce pass it replaces)
        unsigned warm = 0;
        if (bsm) {
            constexpr int LINES = TY * W;                                      // pixels of the brick (<= 128)
            const size_t b0 = ((size_t)(n * p.H + y0) * W) * 64;
            if (t < LINES) warm = *(const unsigned*)((const AT*)p.bs_y + b0 + (size_t)t * 64);
            else if (bsm == 2 && t < 2 * LINES) warm = *(const unsigned*)((const AT*)p.bs_ro + b0 + (size_t)(t - LINES) * 64);
        }
        __syncthreads();                                                       // every wave is done with the previous slab
        {
            uint4 pre[C::MAXC];
            int dst[C::MAXC];
#pragma unroll
            for (int u = 0; u < C::MAXC; ++u) {
                const int i = t + u * 256;
                const int yy = i / C::CPR, cc = i % C::CPR;
                const int sx = C::XOFF + (cc >> 3), q = cc & 7;
                const int gy = y0 - 1 + yy;
                const bool inside = i < C::ITEMS;
                dst[u] = inside ? yy * C::PITCH + sx * 128 + ((q ^ ((sx >> 1) & 7)) << 4) : -1;
                const bool ok = inside && (unsigned)gy < (unsigned)p.H;
                const unsigned voff = ok ? (unsigned)(((n * p.H + gy) * W) * 128 + cc * 16) : 0x80000000u;
                pre[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(irs, voff, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < C::MAXC; ++u)
                if (dst[u] >= 0) *(uint4*)(slab + dst[u]) = pre[u];
        }
        if (bsm) asm volatile("" :: "v"(warm));                                // (the touch only has to have been issued)
        __syncthreads();

#pragma unroll 1
        for (int r = h; r < C::RUNS; r += 2) {
            const int yl = r / C::RPR, xr = r % C::RPR;
            const char* sb = slab + yl * C::PITCH + xr * 2048;
            f32x4 acc[2];
            acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            typedef E e4 __attribute__((ext_vector_type(4)));
            const size_t eo = (((size_t)(n * p.H + y0 + yl) * W) + xr * 16 + fr) * 64 + 32 * c + fq * 4;
            AT* const o = (AT*)p.out + eo;
            const bool valid = !(W % 16) || xr * 16 + fr < W;                  // (the unused half of a 56-wide row's last run)
            e4 prev[2];                                                        // ACCUM: what `out` holds, requested before the MFMAs
            if (ACCUM && valid) { prev[0] = *(const e4*)o; prev[1] = *(const e4*)(o + 16); }
            e4 bsy[2], bsr[2];                                                 // BatchNorm-backward sums: y (and the saved output), requested
                                                                               // under the last kernel row's MFMAs (the registers of the
                                                                               // finished row's fragments are free then)
            // the six fragments of a kernel row (3 kx x 2 k-steps) are read as one batch, the next row's batch is issued before this
            // row's twelve MFMAs
            v8 bf[2][6];
#pragma unroll
            for (int i = 0; i < 6; ++i) bf[0][i] = *(const v8*)(sb + lofs[i >> 1][i & 1]);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if (ky < 2) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) bf[(ky + 1) & 1][i] = *(const v8*)(sb + lofs[i >> 1][i & 1] + (ky + 1) * C::PITCH);
                }
                if (ky == 2 && bsm && valid) {
                    bsy[0] = *(const e4*)((const AT*)p.bs_y + eo); bsy[1] = *(const e4*)((const AT*)p.bs_y + eo + 16);
                    if (bsm == 2) { bsr[0] = *(const e4*)((const AT*)p.bs_ro + eo); bsr[1] = *(const e4*)((const AT*)p.bs_ro + eo + 16); }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    acc[0] = MM::mma(wf[ky * 3 + (i >> 1)][i & 1][0], bf[ky & 1][i], acc[0]);
                    acc[1] = MM::mma(wf[ky * 3 + (i >> 1)][i & 1][1], bf[ky & 1][i], acc[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!valid) continue;                                              // nothing stored, nothing counted
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x4 v = acc[ct];
                if (ACCUM) {
                    const e4 e = prev[ct];
                    v[0] += (float)e[0]; v[1] += (float)e[1]; v[2] += (float)e[2]; v[3] += (float)e[3];
                }
                const e4 hh = __builtin_convertvector(v, e4);
                *(e4*)(o + 16 * ct) = hh;
                f32x4 rv = {(float)hh[0], (float)hh[1], (float)hh[2], (float)hh[3]};
                if (bsm) {
                    const f32x4 yv = {(float)bsy[ct][0], (float)bsy[ct][1], (float)bsy[ct][2], (float)bsy[ct][3]};
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
                    if (bsm == 1) { sc = *(const f32x4*)(bsc + 32 * c + 16 * ct + 4 * fq); sh = *(const f32x4*)(bsc + 64 + 32 * c + 16 * ct + 4 * fq); }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool keep = bsm == 1 ? __fmaf_rn(yv[r], sc[r], sh[r]) > 0.f : (float)bsr[ct][r] > 0.f;
                        rv[r] = keep ? rv[r] : 0.f;
                    }
                    cs[ct] += rv;
                    cq[ct] += rv * yv;
                } else {
                    cs[ct] += rv;
                    cq[ct] += rv * rv;
                }
            }
        }
    }

    if (p.stats) {                                                             // one record per workgroup: channels 32 c + 16 ct + 4 fq + r, both run parities
        float* const red = (float*)slab;                                       // [4 waves][32 channels][2] (the slab is idle now)
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s_ = cs[ct][r], q_ = cq[ct][r];
                s_ += c64_row_ror<8>(s_); q_ += c64_row_ror<8>(q_);
                s_ += c64_row_ror<4>(s_); q_ += c64_row_ror<4>(q_);
                s_ += c64_row_ror<2>(s_); q_ += c64_row_ror<2>(q_);
                s_ += c64_row_ror<1>(s_); q_ += c64_row_ror<1>(q_);
                if (fr == 0) {
                    red[(wave * 32 + 16 * ct + 4 * fq + r) * 2 + 0] = s_;
                    red[(wave * 32 + 16 * ct + 4 * fq + r) * 2 + 1] = q_;
                }
            }
        __syncthreads();
        if (t < 64) {                                                          // channel t = 32 c' + i: waves c' (h = 0) and c' + 2 (h = 1)
            const int cw = t >> 5, i = t & 31;
            const float s_ = red[(cw * 32 + i) * 2] + red[((cw + 2) * 32 + i) * 2];
            const float q_ = red[(cw * 32 + i) * 2 + 1] + red[((cw + 2) * 32 + i) * 2 + 1];
            p.stats[(size_t)blockIdx.x * 128 + t] = s_;
            p.stats[(size_t)blockIdx.x * 128 + 64 + t] = q_;
        }
    }
}

static bool c64_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_C64_CONV"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

bool tri_internal_c64_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                               int pd, int ph, int pw, TriC64Geom* g) {
    if (c64_disabled()) return false;
    if (ID != 1 || OD != 1 || KD != 1 || KH != 3 || KW != 3 || stride != 1 || pd != 0 || ph != 1 || pw != 1) return false;
    if (cin != 64 || cout != 64 || OH != IH || OW != IW) return false;
    int ty;
    if (IW == 32) ty = 4; else if (IW == 64 || IW == 56) ty = 2; else if (IW == 16) ty = 8; else return false;
    if ((long)B * IH * IW * 128 >= (1L << 31)) return false;                  // 32-bit buffer offsets
    const int slots = 2 * tri_internal_num_cus();
    // bricks of 8 runs when that gives every persistent workgroup >= 3 of them, else bricks of 4 (the bench shape: 768 bricks of 8 runs on
    // 512 workgroups are 2 rounds for 1.5 rounds of work; 1,536 bricks of 4 runs are 3 each)
    if (IH % ty || (long)B * (IH / ty) < 3L * slots) ty /= 2;
    if (IH % ty) return false;
    g->W = IW; g->TY = ty;
    g->nbricks = B * (IH / ty);
    g->grid = g->nbricks < slots ? g->nbricks : slots;
    return true;
}

template <typename AT, int W, int TY, bool ACCUM, bool BS>
static int c64_launch_t(const ConvC64Args& a, int grid, hipStream_t stream) {
    typedef C64Cfg<W, TY> C;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv_c64_kernel<AT, W, TY, ACCUM, BS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM + 512);
        attr = true;
    }
    conv_c64_kernel<AT, W, TY, ACCUM, BS><<<grid, 256, C::SMEM + 512, stream>>>(a);
    return tri_check_launch("tri_conv(c64)");
}

int tri_internal_c64_launch(const TriC64Geom& g, int B, int H, const void* in, const void* w, void* out, float* stats, int transposed,
                            int accumulate, int act_fmt, const TriConvBnSums* bs, hipStream_t stream) {
    ConvC64Args a{};
    a.in = in; a.w = w; a.out = out; a.stats = stats;
    if (bs) {
        a.bs_y = bs->y; a.bs_ro = bs->relu_out; a.bs_scale = bs->relu_scale; a.bs_shift = bs->relu_shift;
        a.bs_mode = accumulate ? 2 : 1;
        if (accumulate ? !bs->relu_out : !bs->relu_scale) {
            tri_set_error("conv(c64): BatchNorm-backward sums come in two forms: relu_scale / relu_shift without accumulate, relu_out with it");
            return TRI_ERR_UNSUPPORTED;
        }
    }
    a.N = B; a.H = H; a.nbricks = g.nbricks; a.transposed = transposed; a.accumulate = accumulate;
    a.in_bytes = (unsigned)((size_t)B * H * g.W * 128);
#define TRI_C64_BS(W_, TY_, BS_)                                                                                           \
    if (g.W == W_ && g.TY == TY_ && (bs != nullptr) == BS_) {                                                              \
        if (act_fmt == TRI_FMT_F16)                                                                                        \
            return accumulate ? c64_launch_t<f16_t, W_, TY_, true, BS_>(a, g.grid, stream) : c64_launch_t<f16_t, W_, TY_, false, BS_>(a, g.grid, stream);   \
        return accumulate ? c64_launch_t<bf16_t, W_, TY_, true, BS_>(a, g.grid, stream) : c64_launch_t<bf16_t, W_, TY_, false, BS_>(a, g.grid, stream);     \
    }
    // (the sums forms only where they fit the register file: the 56- / 64-wide bricks spill with them and config 5 lost 0.2 ms)
#define TRI_C64(W_, TY_) TRI_C64_BS(W_, TY_, false)
    TRI_C64(32, 4)
    TRI_C64(32, 2)
    TRI_C64(64, 2)
    TRI_C64(64, 1)
    TRI_C64(56, 2)
    TRI_C64(56, 1)
    TRI_C64(16, 8)
    TRI_C64(16, 4)
    TRI_C64_BS(32, 4, true)
    TRI_C64_BS(32, 2, true)
    TRI_C64_BS(16, 8, true)
    TRI_C64_BS(16, 4, true)
#undef TRI_C64_BS
#undef TRI_C64
    tri_set_error("conv(c64): brick shape not instantiated (BatchNorm-backward sums: 16- and 32-wide images only)");
    return TRI_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Data gradient of the 64 -> 128 channel 3x3 / 2 / pad 1 layer (layer2's first conv, mv_cnn.py:44 BasicBlock(64, 128, stride 2)):
// dx[2i + a, 2j + b] = sum over the taps (ky, kx) with ky = a + 1, kx = b + 1 (mod 2) of dOut[i + (a + 1 - ky) / 2, j + (b + 1 - kx) / 2] W[ky, kx]:
// every dOut pixel pair (i + di, j + dj), di, dj in {0, 1}, feeds 4 / 2 / 2 / 1 of the nine taps, each into one of the four parity classes
// (a, b) of dx.  conv_dma_kernel runs this as 1,536 row tiles of 128 positions sorted by class, 1-4 live taps each - 4.7 MFLOP tiles
// that re-gather their dOut rows per tap (33 us, 218 TF at the bench shape).  Here, like conv_c64_kernel:
//   * a persistent workgroup (two per CU) walks bricks of TY dOut rows (+ the row below), staged once in LDS as two half-slabs of 64
//     channels (128 B per pixel each: conv_c64_kernel's layout and swizzle; the column right of the image is a zero pixel);
//   * wave w holds the A fragments of dx channels 16 w .. 16 w + 15 for ALL nine taps x 4 k-steps (36 fragments, 144 registers) and
//     takes every run of 16 dOut pixels: the 16 activation fragments of a run (4 pixel offsets x 4 k-steps) feed 36 MFMAs into four
//     accumulators, one per parity class - 2.25 MFMAs per LDS read;
//   * the epilogue scatters the four classes to dx (2 i + a, 2 j + b), optionally added to what dx holds (the shortcut's gradient).
struct ConvS2dArgs {
    const void* in;            // dOut [N, H, W, 128] 16-bit
    const void* w;             // transposed operand rows [64][9 * 128] (k = tap * 128 + output channel)
    void* out;                 // dx [N, 2 H, 2 W, 64]
    int N, H, nbricks;
    unsigned in_bytes;
};

template <int W, int TY>
struct S2dCfg {
    static constexpr int RPR = W / 16;                                         // runs per dOut row
    static constexpr int P = RPR * 16 + 16;                                    // pixels per slab row (a multiple of 16: the swizzle depends on x only)
    static constexpr int PITCH = P * 128;
    static constexpr int HALF = (TY + 1) * PITCH;                              // one half-slab: 64 of the 128 channels
    static constexpr int SLAB = 2 * HALF;
    static constexpr int CPR = W * 16;                                         // 16-byte chunks per dOut row
    static constexpr int ITEMS = (TY + 1) * CPR;
    static constexpr int MAXC = (ITEMS + 255) / 256;
    static constexpr int WROW = 3 * 128 * 2 + 16;                              // filter bank staged three taps at a time: row pitch (16 rows -> 16 bank quads)
    static constexpr size_t SMEM = (size_t)SLAB > (size_t)64 * WROW ? (size_t)SLAB : (size_t)64 * WROW;
    static_assert(W % 16 == 0, "whole runs");
    static_assert(HALF + PITCH + 4096 < 65536, "fragment-read immediates");
};

template <typename AT, int W, int TY, bool ACCUM>
__global__ __launch_bounds__(256, 2) void conv_s2d_kernel(const ConvS2dArgs p) {
    typedef S2dCfg<W, TY> C;
    typedef typename OpOf<AT>::E E;
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    constexpr int KPAD = 9 * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
    char* const slab = smem;
    const int G = gridDim.x, wg = blockIdx.x;
    const int bpi = p.H / TY;                                                  // bricks per image

    // ---- filter bank: A fragment (tap, k-step): row fr = dx channel 16 wave + fr, k = 8 fq .. 8 fq + 7 of the k-step's 32 dOut channels;
    // staged through LDS three taps at a time with coalesced loads (conv_c64_kernel's lesson)
    v8 wf[9][4];
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        constexpr int CH_ROW = 3 * 128 * 2 / 16;                               // 48 chunks per row of a part
        constexpr int NCH = 64 * CH_ROW;                                       // 3,072 chunks
        if (part) __syncthreads();                                             // the previous part's fragments are in registers
#pragma unroll
        for (int u = 0; u < NCH / 256; ++u) {
            const int ch = t + u * 256, row = ch / CH_ROW, col = ch - row * CH_ROW;
            *(uint4*)(slab + row * C::WROW + col * 16) = *(const uint4*)((const char*)p.w + ((size_t)row * KPAD + part * 384) * 2 + col * 16);
        }
        __syncthreads();
        const char* wrow = slab + (16 * wave + fr) * C::WROW + fq * 16;
#pragma unroll
        for (int tp = 0; tp < 3; ++tp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[part * 3 + tp][ks] = *(const v8*)(wrow + (tp * 128 + ks * 32) * 2);
    }
    __syncthreads();                                                           // the slab takes the staging area over
    // per-lane slab offsets for (dj, k-step within a half): pixel fr + dj, chunk 4 kk + fq at its swizzled position
    int lofs[2][2];
#pragma unroll
    for (int dj = 0; dj < 2; ++dj) {
        const int sx = fr + dj;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) lofs[dj][kk] = sx * 128 + (((4 * kk + fq) ^ ((sx >> 1) & 7)) << 4);
    }
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    // the zero pixel right of every slab row, both halves (written once: the fill below never touches it)
    for (int i = t; i < 2 * (TY + 1) * 8; i += 256) {
        const int hf = i / ((TY + 1) * 8), rem = i - hf * ((TY + 1) * 8), row = rem >> 3, q = rem & 7;
        *(uint4*)(slab + hf * C::HALF + row * C::PITCH + W * 128 + q * 16) = make_uint4(0u, 0u, 0u, 0u);
    }

#pragma unroll 1
    for (int j = wg; j < p.nbricks; j += G) {
        const int n = j / bpi, y0 = (j - n * bpi) * TY;
        __syncthreads();                                                       // every wave is done with the previous slab
        {
            uint4 pre[C::MAXC];
            int dst[C::MAXC];
#pragma unroll
            for (int u = 0; u < C::MAXC; ++u) {
                const int i = t + u * 256;
                const int yy = i / C::CPR, cc = i % C::CPR;
                const int sx = cc >> 4, q = cc & 15;
                const int gy = y0 + yy;
                const bool inside = i < C::ITEMS;
                dst[u] = inside ? (q >> 3) * C::HALF + yy * C::PITCH + sx * 128 + (((q & 7) ^ ((sx >> 1) & 7)) << 4) : -1;
                const bool ok = inside && gy < p.H;                            // (the row below the image: zeros)
                const unsigned voff = ok ? (unsigned)(((n * p.H + gy) * W) * 256 + cc * 16) : 0x80000000u;
                pre[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(irs, voff, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < C::MAXC; ++u)
                if (dst[u] >= 0) *(uint4*)(slab + dst[u]) = pre[u];
        }
        __syncthreads();

#pragma unroll 1
        for (int r = 0; r < TY * C::RPR; ++r) {
            const int yl = r / C::RPR, xr = r % C::RPR;
            const char* sb = slab + yl * C::PITCH + xr * 2048;
            typedef E e4 __attribute__((ext_vector_type(4)));
            // dx pixel (2 (y0 + yl) + a, 2 (16 xr + fr) + b), channels 16 wave + 4 fq ..
            const size_t eo = ((((size_t)n * 2 * p.H + 2 * (y0 + yl)) * (2 * W)) + 2 * (xr * 16 + fr)) * 64 + 16 * wave + fq * 4;
            AT* const o = (AT*)p.out + eo;
            constexpr size_t OROW = (size_t)2 * W * 64;                        // one dx image row
            f32x4 acc[2][2];
            acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // fragments of the dOut row itself (di = 0), then of the row below (di = 1): 2 pixel offsets x 4 k-steps each
            v8 bf[2][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) bf[0][i] = *(const v8*)(sb + lofs[i >> 2][i & 1] + ((i >> 1) & 1) * C::HALF);
#pragma unroll
            for (int i = 0; i < 8; ++i) bf[1][i] = *(const v8*)(sb + lofs[i >> 2][i & 1] + ((i >> 1) & 1) * C::HALF + C::PITCH);
            e4 prev[2][2];                                                     // ACCUM: what dx holds, requested before the MFMAs
            if (ACCUM) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) prev[a][b] = *(const e4*)(o + a * OROW + b * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
            // taps by (ky, kx); class a takes ky = 1 (a = 0, di 0), ky = 2 (a = 1, di 0), ky = 0 (a = 1, di 1); the same for b / kx / dj
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const v8 b00 = bf[0][ks], b01 = bf[0][4 + ks];                 // (di, dj) = (0, 0), (0, 1)
                acc[0][0] = MM::mma(wf[1 * 3 + 1][ks], b00, acc[0][0]);
                acc[0][1] = MM::mma(wf[1 * 3 + 2][ks], b00, acc[0][1]);
                acc[1][0] = MM::mma(wf[2 * 3 + 1][ks], b00, acc[1][0]);
                acc[1][1] = MM::mma(wf[2 * 3 + 2][ks], b00, acc[1][1]);
                acc[0][1] = MM::mma(wf[1 * 3 + 0][ks], b01, acc[0][1]);
                acc[1][1] = MM::mma(wf[2 * 3 + 0][ks], b01, acc[1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const v8 b10 = bf[1][ks], b11 = bf[1][4 + ks];                 // (1, 0), (1, 1)
                acc[1][0] = MM::mma(wf[0 * 3 + 1][ks], b10, acc[1][0]);
                acc[1][1] = MM::mma(wf[0 * 3 + 2][ks], b10, acc[1][1]);
                acc[1][1] = MM::mma(wf[0 * 3 + 0][ks], b11, acc[1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 v = acc[a][b];
                    if (ACCUM) {
                        const e4 e = prev[a][b];
                        v[0] += (float)e[0]; v[1] += (float)e[1]; v[2] += (float)e[2]; v[3] += (float)e[3];
                    }
                    *(e4*)(o + a * OROW + b * 64) = __builtin_convertvector(v, e4);
                }
        }
    }
}

static bool s2d_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_S2D_CONV"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

// GEMM view of the data-gradient call: "input" = dOut grid (128 channels), "output" = dx grid (64 channels), twice as large
bool tri_internal_s2d_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                               int pd, int ph, int pw, TriC64Geom* g) {
    if (s2d_disabled()) return false;
    if (ID != 1 || OD != 1 || KD != 1 || KH != 3 || KW != 3 || stride != 2 || pd != 0 || ph != 1 || pw != 1) return false;
    if (cin != 128 || cout != 64 || OH != 2 * IH || OW != 2 * IW) return false;
    if (IW != 16 && IW != 32) return false;
    if ((long)B * OH * OW * 128 >= (1L << 31) || (long)B * IH * IW * 256 >= (1L << 31)) return false;
    const int slots = 2 * tri_internal_num_cus();
    int ty = IW == 16 ? 8 : 4;
    while (ty > 2 && (IH % ty || (long)B * (IH / ty) < 3L * slots)) ty /= 2;   // >= 3 bricks per persistent workgroup where the batch allows
    if (const char* e = getenv("TRICOLO_S2D_TY")) {                            // tests: the larger bricks without a batch of hundreds
        const int v = atoi(e);
        if ((v == 2 || v == 4 || (v == 8 && IW == 16)) && IH % v == 0) ty = v;
    }
    if (IH % ty) return false;
    g->W = IW; g->TY = ty;
    g->nbricks = B * (IH / ty);
    g->grid = g->nbricks < slots ? g->nbricks : slots;
    return true;
}

template <typename AT, int W, int TY, bool ACCUM>
static int s2d_launch_t(const ConvS2dArgs& a, int grid, hipStream_t stream) {
    typedef S2dCfg<W, TY> C;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv_s2d_kernel<AT, W, TY, ACCUM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
        attr = true;
    }
    conv_s2d_kernel<AT, W, TY, ACCUM><<<grid, 256, C::SMEM, stream>>>(a);
    return tri_check_launch("tri_conv(s2d)");
}

int tri_internal_s2d_launch(const TriC64Geom& g, int B, int H, const void* in, const void* w, void* out, int accumulate, int act_fmt,
                            hipStream_t stream) {
    ConvS2dArgs a{};
    a.in = in; a.w = w; a.out = out;
    a.N = B; a.H = H; a.nbricks = g.nbricks;
    a.in_bytes = (unsigned)((size_t)B * H * g.W * 256);
#define TRI_S2D(W_, TY_)                                                                                                   \
    if (g.W == W_ && g.TY == TY_) {                                                                                        \
        if (act_fmt == TRI_FMT_F16)                                                                                        \
            return accumulate ? s2d_launch_t<f16_t, W_, TY_, true>(a, g.grid, stream) : s2d_launch_t<f16_t, W_, TY_, false>(a, g.grid, stream);   \
        return accumulate ? s2d_launch_t<bf16_t, W_, TY_, true>(a, g.grid, stream) : s2d_launch_t<bf16_t, W_, TY_, false>(a, g.grid, stream);     \
    }
    TRI_S2D(16, 8)
    TRI_S2D(16, 4)
    TRI_S2D(16, 2)
    TRI_S2D(32, 4)
    TRI_S2D(32, 2)
#undef TRI_S2D
    tri_set_error("conv(s2d): brick shape not instantiated");
    return TRI_ERR_UNSUPPORTED;
}
