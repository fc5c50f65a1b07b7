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
#include <stdio.h>
#include <stdlib.h>
#include "../../include/tricolo_hip.h"
#include "conv_vox.h"

struct WgradArgs {
    const void* in;              // activations / gradients: fp32 or bf16 (kernel template parameter AT)
    const void* dout;
    const uint8_t* row_mask;
    const int* row_pos;          // optional compact list of the output positions to contract over (submanifold layers: the
    const int* row_count;        // active sites, tri_mask_compact) + its device-side length; then row_mask is not consulted
    float* slab;                 // [splits][Cout][Kpad]
    const int* plan_off;         // optional gather plan: element offset of each output position's origin voxel
    const unsigned* plan_mask;   //                       packed per-axis tap validity bits (8 per axis)
    unsigned in_bytes;
    int B, ID, IH, IW, Cin;
    int OD, OH, OW, Cout;
    int KD, KH, KW, stride, pd, ph, pw;
    int Kpad, M, ntaps, cin_shift, steps_per_split, ntiles, nsplits;
    int plan_ring;               // conv_wgrad_dma_kernel: the gather plan is staged through a ring of 2 x 16 steps (any steps_per_split)
    FastDiv dOW, dOH, dOD, dCin;
#ifdef WGRAD_STAMPS
    long long* dbg;              // tools/probes/wgrad_probe.hip: per-workgroup (id, cycle) stamps of wave 0
#endif
};
// In-kernel phase stamps of conv_wgrad_dma_kernel (-DWGRAD_STAMPS builds only): wave 0 writes (id << 48 | s_memtime) into the last
// 2 KiB of the workgroup's LDS and copies them to p.dbg at the end; ~150 cycles per stamp.
#ifdef WGRAD_STAMPS
static long long* g_wgrad_dbg = nullptr;
#define WSTAMP(id) do { if (wave == 0 && n_stamp < 255) { const long long c_ = __builtin_readcyclecounter(); if (lane == 0) stl[n_stamp] = ((long long)(id) << 48) | (c_ & 0xFFFFFFFFFFFFll); ++n_stamp; } } while (0)
#else
#define WSTAMP(id) do { } while (0)
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));

// LDS operand images are NATURAL layout [32 positions][C channels] bf16 (what the global loads deliver: 8-byte writes,
// consecutive lanes -> consecutive addresses, conflict-free).  The position-major -> channel-major transpose both MFMA
// operands need is done by the LDS itself: ds_read_b64_tr_b16 hands lane i of each 16-lane group column i of a
// 4-row x 16-column block (rows = 4 consecutive positions = 4 consecutive k), two reads = one 8-k operand fragment.
// 32-byte chunk c of row r is stored at chunk c ^ sw(r) so the 8 rows one half-wave touches hit distinct banks.
template <int ROWB>
__device__ __forceinline__ int nat_off(int row, int byte) {
    constexpr int NCH = ROWB / 32;
    int sw = NCH >= 8 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
    return row * ROWB + ((((byte >> 5) ^ sw) & (NCH - 1)) << 5) + (byte & 31);
}
template <int ROWB, typename V8>
__device__ __forceinline__ V8 tr_frag(const char* tile, int col0, int g, int q, int pq) {
    // rows 8g+q and 8g+4+q, columns col0 + 4*pq .. +3 (bf16)
    const int byte = (col0 + 4 * pq) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + nat_off<ROWB>(8 * g + q, byte)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + nat_off<ROWB>(8 * g + 4 + q, byte)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(V8, r);
}

template <int BI, int BJ, int NSPLIT, typename AT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int WI = BI / 2, WJ = BJ / 2, TM = WI / 16, TN = WJ / 16;
    constexpr int XROW = BI * 2, YROW = BJ * 2;                  // bytes per position row
    constexpr int X_BYTES = 32 * XROW, Y_BYTES = 32 * YROW;
    constexpr int STAGE = NSPLIT * (X_BYTES + Y_BYTES);
    constexpr int XL = 8 * BI, YL = 8 * BJ;                      // float4 loads per k-step (32 positions x C/4 quads)
    constexpr int LOADS = (XL + YL) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lut = (int*)(smem + 2 * STAGE);
    int* lut_off = lut + 64;
    int* lplan_off = lut + 128;                                  // [steps_per_split * 32] origin offsets of this block's positions
    unsigned* lplan_mask = (unsigned*)lplan_off + p.steps_per_split * 32;
    int* lrow = (int*)lplan_mask + p.steps_per_split * 32;      // row list launches: the position of every list entry of this block

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int JT = (p.Kpad + BJ - 1) / BJ;
    // XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (private L2 each).  All tiles of one
    // position split read the same dOut / input rows, so split s is pinned to XCD s % 8: within a group of 8 * ntiles
    // consecutive workgroups, (id % 8) picks the split and (id / 8) the tile.  (PMC: the row-major mapping re-fetched
    // the operands once per XCD - 4.8x the algorithmic HBM/MALL bytes on the 64-channel 3x3 layers.)
    // Layers with only a few splits keep the plain (tile-fastest) order: pinning 6 splits to 6 XCDs would idle two.
    const int ntiles = p.ntiles;
    int split, tile;
    if (p.nsplits >= 16) {
        const int grp = blockIdx.x / (8 * ntiles), rem = blockIdx.x - grp * 8 * ntiles;
        split = grp * 8 + (rem & 7);
        tile = rem >> 3;
    } else {
        split = blockIdx.x / ntiles;
        tile = blockIdx.x - split * ntiles;
    }
    if (split >= p.nsplits) return;
    const int it = tile / JT, jt = tile - it * JT;
    const int i0 = it * BI, j0 = jt * BJ;
    // row list: the splits share the *row_count list entries evenly (the static steps_per_split is the bound for a full list)
    const int nrows = p.row_count ? *p.row_count : p.M;
    const int nsteps_total = (nrows + 31) >> 5;
    const int sps = p.row_count ? (nsteps_total + p.nsplits - 1) / p.nsplits : p.steps_per_split;
    const int ks_begin = split * sps;
    const int ks_end = min(nsteps_total, ks_begin + sps);

    if (t < 64) {
        int kd = 0, kh = 0, kw = 0;
        if (t < p.ntaps) {
            kw = t % p.KW;
            int r = t / p.KW;
            kh = r % p.KH;
            kd = r / p.KH;
        }
        lut[t] = kd | (kh << 8) | (kw << 16);
        lut_off[t] = ((kd * p.IH + kh) * p.IW + kw) * p.Cin;
    }
    // the gather plan of this block's position range goes to LDS once: per-load plan reads then cost no VMEM issue
    if (p.row_count) {
        const int n = (ks_end - ks_begin) * 32;
        for (int i = t; i < n; i += 256) {
            const int idx = ks_begin * 32 + i;
            const int m = idx < nrows ? p.row_pos[idx] : -1;
            lrow[i] = m;
            lplan_off[i] = m >= 0 ? p.plan_off[m] : 0;
            lplan_mask[i] = m >= 0 ? p.plan_mask[m] : 0u;      // no valid tap: the gather of a padding entry delivers zeros
        }
    } else {
        const int n4 = (ks_end - ks_begin) * 8;                 // int4 chunks
        const int4* so = (const int4*)(p.plan_off + ks_begin * 32);
        const int4* sm = (const int4*)(p.plan_mask + ks_begin * 32);
        for (int i = t; i < n4; i += 256) { ((int4*)lplan_off)[i] = so[i]; ((int4*)lplan_mask)[i] = sm[i]; }
    }
    __syncthreads();

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wi = wave >> 1, wj = wave & 1;
    const int fr = lane & 15, fg = lane >> 4, fqq = fr >> 2, fp = fr & 3;

    // per-thread constants of the Y (input) gather: the (tap, channel quad) a thread fetches never changes
    int y_toff[LOADS], y_sh[LOADS];
    bool y_tv[LOADS];
#pragma unroll
    for (int u = 0; u < LOADS; ++u) {
        int e = t + u * 256;
        y_toff[u] = 0; y_sh[u] = 0; y_tv[u] = false;
        if (e >= XL) {
            int quad = (e - XL) % (BJ / 4);
            int j = j0 + quad * 4;
            int tap, c;
            if (p.cin_shift >= 0) { tap = j >> p.cin_shift; c = j & ((1 << p.cin_shift) - 1); }
            else { tap = (int)fdiv((uint32_t)j, p.dCin); c = j - tap * p.Cin; }
            bool tv = tap < p.ntaps;
            int code = lut[tv ? tap : 0];
            int kd = code & 255, kh = (code >> 8) & 255, kw = (code >> 16) & 255;
            y_tv[u] = tv;
            y_toff[u] = lut_off[tv ? tap : 0] + c;
            y_sh[u] = kw | ((8 + kh) << 8) | ((16 + kd) << 16);
        }
    }

    float4 v[LOADS];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);

    // a 32-position step is live when any of its output sites is active (dense layers: always)
    auto step_live = [&](int ks) -> bool {
        if (!p.row_mask || p.row_count) return true;
        int m = ks * 32;
        const uint32_t* mp = (const uint32_t*)(p.row_mask + m);          // mask buffers are padded to 32 bytes
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
        for (int u = 0; u < LOADS; ++u) {
            int e = t + u * 256;
            if (e < XL) {
                int quad = e % (BI / 4), pos = e / (BI / 4);
                int m = p.row_count ? lrow[mbase + pos - ks_begin * 32] : mbase + pos, co = i0 + quad * 4;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m >= 0 && m < p.M && co < p.Cout) {
                    if (sizeof(AT) == 4) x = *(const float4*)((const float*)p.dout + (size_t)m * p.Cout + co);
                    else { uint2 h = *(const uint2*)((const uint16_t*)p.dout + (size_t)m * p.Cout + co); x.x = __builtin_bit_cast(float, h.x); x.y = __builtin_bit_cast(float, h.y); }
                }
                v[u] = x;
            } else {
                int pos = (e - XL) / (BJ / 4);
                int m = mbase + pos;                                     // list index (row list) or position: the plan copy in LDS is indexed alike
                int ro = lplan_off[m - ks_begin * 32];
                unsigned rm = lplan_mask[m - ks_begin * 32];
                int sh = y_sh[u];
                bool ok = y_tv[u] && (((rm >> (sh & 255)) & (rm >> ((sh >> 8) & 255)) & (rm >> ((sh >> 16) & 255))) & 1u);
                if (sizeof(AT) == 4) {
                    unsigned voff = ok ? (unsigned)((ro + y_toff[u]) << 2) : 0x80000000u;
                    v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
                } else {                                          // bf16: the 8 loaded bytes ARE the LDS payload (kept in .x/.y)
                    unsigned voff = ok ? (unsigned)((ro + y_toff[u]) << 1) : 0x80000000u;
                    uint2 h = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0));
                    v[u] = make_float4(__builtin_bit_cast(float, h.x), __builtin_bit_cast(float, h.y), 0.f, 0.f);
                }
            }
        }
    };
    auto store_lds = [&](int buf) {
        char* xb = smem + buf * STAGE;
        char* yb = xb + NSPLIT * X_BYTES;
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            int e = t + u * 256;
            char* dst;
            int lo_off;
            if (e < XL) {
                int quad = e % (BI / 4), pos = e / (BI / 4);
                dst = xb + nat_off<XROW>(pos, quad * 8);
                lo_off = X_BYTES;
            } else {
                int quad = (e - XL) % (BJ / 4), pos = (e - XL) / (BJ / 4);
                dst = yb + nat_off<YROW>(pos, quad * 8);
                lo_off = Y_BYTES;
            }
            if (sizeof(AT) == 2) {
                *(uint2*)dst = make_uint2(__builtin_bit_cast(unsigned, v[u].x), __builtin_bit_cast(unsigned, v[u].y));
                if (NSPLIT == 2) *(uint2*)(dst + lo_off) = make_uint2(0, 0);
            } else if (NSPLIT == 2) {
                bf16x4 h, l;
                split_bf16(v[u], h, l);
                *(bf16x4*)dst = h;
                *(bf16x4*)(dst + lo_off) = l;
            } else {
                *(bf16x4*)dst = to_bf16x4(v[u]);
            }
        }
    };
    auto compute = [&](int buf) {
        const char* xb = smem + buf * STAGE;
        const char* yb = xb + NSPLIT * X_BYTES;
        typedef Mma<typename OpOf<AT>::E> MM;
        typedef typename MM::v8 v8;
        v8 ah[TM], al[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            ah[a] = tr_frag<XROW, v8>(xb, wi * WI + a * 16, fg, fqq, fp);
            if (NSPLIT == 2) al[a] = tr_frag<XROW, v8>(xb + X_BYTES, wi * WI + a * 16, fg, fqq, fp);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            v8 bhf = tr_frag<YROW, v8>(yb, wj * WJ + b * 16, fg, fqq, fp);
            v8 blf;
            if (NSPLIT == 2) blf = tr_frag<YROW, v8>(yb + Y_BYTES, wj * WJ + b * 16, fg, fqq, fp);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                if (NSPLIT == 2) {
                    acc[a][b] = MM::mma(al[a], bhf, acc[a][b]);
                    acc[a][b] = MM::mma(ah[a], blf, acc[a][b]);
                }
                acc[a][b] = MM::mma(ah[a], bhf, acc[a][b]);
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
            int co = i0 + wi * WI + a * 16 + fg * 4 + r;
            if (co >= p.Cout) continue;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                int j = j0 + wj * WJ + b * 16 + fr;
                if (j < p.Kpad) slab[(size_t)co * p.Kpad + j] = acc[a][b][r];
            }
        }
}

// ================================================================================================ LDS-DMA variant
// bf16 activation storage: dOut rows and input rows in HBM already are the bf16 LDS image, so both operand tiles go
// global -> LDS with buffer_load_dwordx4 ... lds (16 B per lane, no VGPR staging / ds_write; issued through
// dma16_async so that the copy really stays in flight under the MFMAs), 64 positions per step
// (32 MFMAs per wave per barrier instead of 16) and the DMA of the next live step in flight under the MFMAs.
// One wave-instruction fills 1 KiB = 1024 / ROWB consecutive tile rows; the nat_off bank swizzle is applied on the
// SOURCE side (the lane that owns LDS slot s of row r fetches the chunk nat_off would have stored there).  The slot ->
// chunk map of a lane is the same for every instruction and step, so (tap, channel, validity shifts) are per-lane constants.
template <int ROWB>
__device__ __forceinline__ int nat_sw(int row) {
    constexpr int NCH = ROWB / 32;
    return NCH >= 8 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
}

// Several layers' weight gradients in ONE launch (tri_conv_wgrad_partial_group): the launch's ~448 resident workgroups are shared by
// the jobs, so each layer is cut into a fraction of the splits it would get alone - the fp32 slab traffic (splits x Cout x K written
// here, re-read by the reduce) shrinks by the number of jobs, and a workgroup's prologue / slab store is paid once per longer split.
#define WGRAD_JOBS_MAX 12
#define WGRAD_RING_STEPS 8                                         // plan ring: 2 chunks of this many 64-position steps (8 KB: two 128x128 workgroups per CU still fit)
struct WgradJobs {
    WgradArgs d[WGRAD_JOBS_MAX];
    int first_block[WGRAD_JOBS_MAX + 1];
    int n;
};
// NWI x NWJ waves, each a (BI / NWI) x (BJ / NWJ) = 64 x 64 (32 x 64 for 64-row tiles) block of the tile: 2 x 2 waves for the 128 x 128 and
// 64 x 128 tiles (two, three workgroups per CU), 4 x 2 for the 256 x 128 tile of the grouped launches (one 512-thread workgroup per CU:
// a third fewer operand bytes through the LDS-DMA path per FLOP, which is what bounds these kernels - profiles/r3/NOTES_wgrad.md).
static_assert(sizeof(WgradJobs) <= 4096, "the job table travels in the kernel arguments (4 KB)");
template <int BI, int BJ, typename E, int NWI = 2, int NWJ = 2>
__global__ __launch_bounds__(NWI * NWJ * 64) void conv_wgrad_dma_kernel(const WgradJobs jobs) {
    constexpr int NW = NWI * NWJ, NT = NW * 64;
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[ji + 1]) ++ji;
    const WgradArgs& p = jobs.d[ji];
    const int bid = (int)blockIdx.x - jobs.first_block[ji];
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    constexpr int KB = 64;
    constexpr int WI = BI / NWI, WJ = BJ / NWJ, TM = WI / 16, TN = WJ / 16;
    constexpr int XROW = BI * 2, YROW = BJ * 2;
    constexpr int X_BYTES = KB * XROW, Y_BYTES = KB * YROW, STAGE = X_BYTES + Y_BYTES;
    constexpr int XRPI = 1024 / XROW, YRPI = 1024 / YROW;        // tile rows per wave-instruction
    constexpr int XNI = KB / (NW * XRPI), YNI = KB / (NW * YRPI);  // instructions per wave per step
    // a lane's chunk swizzle depends on (row & 3) and bit 3 of its row: the rows of its successive instructions must differ by 16s
    static_assert((NW * XRPI) % 16 == 0 && (NW * YRPI) % 16 == 0 && XNI >= 1 && YNI >= 1, "lane -> chunk map must not depend on the instruction");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* lut = (int*)(smem + 2 * STAGE);
    int* lut_off = lut + 64;
    constexpr int RING = 2 * WGRAD_RING_STEPS * KB;              // plan entries of the ring form
    const int pcap = p.plan_ring ? RING : p.steps_per_split * KB;
    const int pmask = p.plan_ring ? RING - 1 : 0x7fffffff;
    int* lplan_off = lut + 128;                                  // [steps_per_split * 64], or the ring
    unsigned* lplan_mask = (unsigned*)lplan_off + pcap;
    int* lrow = (int*)lplan_mask + pcap;                         // row list launches, see conv_wgrad_kernel

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef WGRAD_STAMPS
    int n_stamp = 0;
    long long* const stl = (long long*)(smem + 2 * STAGE + 512 + pcap * (p.row_count ? 12 : 8));
#endif
    WSTAMP(1);
    const int JT = (p.Kpad + BJ - 1) / BJ;
    const int ntiles = p.ntiles;
    int split, tile;
    if (p.nsplits >= 16) {                                       // XCD pinning, see conv_wgrad_kernel
        const int grp = bid / (8 * ntiles), rem = bid - grp * 8 * ntiles;
        split = grp * 8 + (rem & 7);
        tile = rem >> 3;
    } else {
        split = bid / ntiles;
        tile = bid - split * ntiles;
    }
    if (split >= p.nsplits) return;
    const int it = tile / JT, jt = tile - it * JT;
    const int i0 = it * BI, j0 = jt * BJ;
    const int nrows = p.row_count ? *p.row_count : p.M;
    const int nsteps_total = (nrows + KB - 1) / KB;
    const int sps = p.row_count ? (nsteps_total + p.nsplits - 1) / p.nsplits : p.steps_per_split;
    const int ks_begin = split * sps;
    const int ks_end = min(nsteps_total, ks_begin + sps);

    if (t < 64) {
        int kd = 0, kh = 0, kw = 0;
        if (t < p.ntaps) {
            kw = t % p.KW;
            int r = t / p.KW;
            kh = r % p.KH;
            kd = r / p.KH;
        }
        lut[t] = kd | (kh << 8) | (kw << 16);
        lut_off[t] = ((kd * p.IH + kh) * p.IW + kw) * p.Cin;
    }
    const int mpad = (p.M + 31) & ~31;
    {
        const int n = min((ks_end - ks_begin) * KB, pcap);         // ring form: the first two chunks
        for (int i = t; i < n; i += NT) {
            int m = ks_begin * KB + i;
            if (p.row_count) {
                m = m < nrows ? p.row_pos[m] : -1;
                lrow[i] = m;
                lplan_off[i] = m >= 0 ? p.plan_off[m] : 0;
                lplan_mask[i] = m >= 0 ? p.plan_mask[m] : 0u;
            } else {
                lplan_off[i] = m < mpad ? p.plan_off[m] : 0;
                lplan_mask[i] = m < mpad ? p.plan_mask[m] : 0u;
            }
        }
    }
    __syncthreads();
    WSTAMP(2);                                                     // tap table + gather plan staged

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wi = wave / NWJ, wj = wave % NWJ;
    const int fr = lane & 15, fg = lane >> 4, fqq = fr >> 2, fp = fr & 3;

    // X (dOut) lane constants: row inside the instruction, 16-byte source chunk, byte offset of (row, chunk)
    const int xrow0 = wave * XRPI + lane / (XROW / 16), xs = lane % (XROW / 16);
    const int xchunk = ((((xs >> 1) ^ nat_sw<XROW>(xrow0)) & (XROW / 32 - 1)) << 1) | (xs & 1);
    const unsigned xoff = (unsigned)((xrow0 * p.Cout + i0 + xchunk * 8) * 2);
    const unsigned xstep = (unsigned)(NW * XRPI * p.Cout * 2);  // bytes between the rows of consecutive instructions
    // Y (input gather) lane constants
    const int yrow0 = wave * YRPI + lane / (YROW / 16), ys = lane % (YROW / 16);
    const int ychunk = ((((ys >> 1) ^ nat_sw<YROW>(yrow0)) & (YROW / 32 - 1)) << 1) | (ys & 1);
    int y_toff, y_sh;
    bool y_tv;
    {
        int j = j0 + ychunk * 8;
        int tap, c;
        if (p.cin_shift >= 0) { tap = j >> p.cin_shift; c = j & ((1 << p.cin_shift) - 1); }
        else { tap = (int)fdiv((uint32_t)j, p.dCin); c = j - tap * p.Cin; }
        y_tv = tap < p.ntaps;
        int code = lut[y_tv ? tap : 0];
        int kd = code & 255, kh = (code >> 8) & 255, kw = (code >> 16) & 255;
        y_toff = (lut_off[y_tv ? tap : 0] + c) * 2;
        y_sh = kw | ((8 + kh) << 8) | ((16 + kd) << 16);
    }
    const int sx = y_sh & 255, sy = (y_sh >> 8) & 255, sz = (y_sh >> 16) & 255;

    // (asm-issued DMA, see common.h: with the builtin the compiler waits vmcnt(0) before the first fragment read of every
    //  step, i.e. the next stage's DMA never overlapped the MFMAs)
    const v4i rsrc = make_rsrc_words(p.in, p.in_bytes);
    // rows m >= M of dOut fall outside this descriptor and arrive as zeros
    const v4i xrsrc = make_rsrc_words(p.dout, (unsigned)p.M * p.Cout * 2);
    const unsigned lds0 = lds_addr(smem) + wave * 1024;

    auto step_live = [&](int ks) -> bool {
        if (!p.row_mask || p.row_count) return true;
        int m = ks * KB;
        const uint32_t* mp = (const uint32_t*)(p.row_mask + m);          // mask buffers are padded to 32 bytes
        uint32_t any = 0;
#pragma unroll
        for (int q = 0; q < KB / 4; ++q) any |= (m + q * 4 < p.M) ? mp[q] : 0u;
        return any != 0;
    };
    auto next_live = [&](int ks) -> int {
        while (ks < ks_end && !step_live(ks)) ++ks;
        return ks;
    };
    auto issue = [&](int ks, int buf) {
        const unsigned xb = lds0 + buf * STAGE;
        const unsigned yb = xb + X_BYTES;
        // every LDS lookup of the step first (gather plan, row list), ONE wait, then the DMA pieces back to back: the pieces are
        // asm volatile statements the compiler orders all memory accesses around, so a lookup placed between two of them costs a
        // full LDS round trip per piece (measured: ~250 cycles per gathered piece, 1.3 k of a step's 2.9 k cycles)
        const int pbase = ((ks - ks_begin) * KB) & pmask;
        int ro[YNI];
        unsigned rm[YNI];
#pragma unroll
        for (int i = 0; i < YNI; ++i) { ro[i] = lplan_off[pbase + yrow0 + NW * YRPI * i]; rm[i] = lplan_mask[pbase + yrow0 + NW * YRPI * i]; }
        int xo[XNI];
        if (p.row_count) {                                           // dOut rows through the list
#pragma unroll
            for (int i = 0; i < XNI; ++i) {
                const int m = lrow[pbase + xrow0 + NW * XRPI * i];
                xo[i] = m >= 0 ? (int)((unsigned)(m * p.Cout + i0 + xchunk * 8) * 2u) : (int)0x80000000;
            }
        } else {
            const unsigned xbase = xoff + (unsigned)(ks * KB) * (unsigned)(p.Cout * 2);
#pragma unroll
            for (int i = 0; i < XNI; ++i) xo[i] = (int)(xbase + i * xstep);
        }
        unsigned yo[YNI];
#pragma unroll
        for (int i = 0; i < YNI; ++i) {
            const unsigned live = ((rm[i] >> sx) & (rm[i] >> sy) & (rm[i] >> sz)) & (y_tv ? 1u : 0u);
            yo[i] = live ? (unsigned)(ro[i] * 2 + y_toff) : 0x80000000u;
        }
#pragma unroll
        for (int i = 0; i < XNI; ++i) dma16_async(xrsrc, xb + i * (NW * 1024), xo[i]);
#pragma unroll
        for (int i = 0; i < YNI; ++i) dma16_async(rsrc, yb + i * (NW * 1024), (int)yo[i]);
    };
    auto compute = [&](int buf) {
        const char* xb = smem + buf * STAGE;
        const char* yb = xb + X_BYTES;
#pragma unroll
        for (int h = 0; h < KB / 32; ++h) {
            v8 ah[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) ah[a] = tr_frag<XROW, v8>(xb + h * 32 * XROW, wi * WI + a * 16, fg, fqq, fp);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                v8 bhf = tr_frag<YROW, v8>(yb + h * 32 * YROW, wj * WJ + b * 16, fg, fqq, fp);
#pragma unroll
                for (int a = 0; a < TM; ++a) acc[a][b] = MM::mma(ah[a], bhf, acc[a][b]);
            }
        }
    };

    int ks = next_live(ks_begin);
    int buf = 0;
    constexpr int RPT = RING / 2 / NT;                             // plan entries of one ring chunk per thread
    static_assert(RPT >= 1, "ring chunk smaller than the workgroup");
    int rrow[RPT];                                                 // row-list launches: the next chunk's positions, fetched one step ahead of their plan entries
#pragma unroll
    for (int u = 0; u < RPT; ++u) rrow[u] = -1;
    WSTAMP(3);                                                     // lane constants
    if (ks < ks_end) {
        issue(ks, 0);
        while (ks < ks_end) {
            int nxt = next_live(ks + 1);
            WSTAMP(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's DMAs of stage `buf` have landed
            WSTAMP(5);                                                 // DMA wait
            __builtin_amdgcn_s_barrier();                                // ... everyone's have, and all reads of buf ^ 1 are done
            asm volatile("" ::: "memory");
            WSTAMP(6);                                                 // barrier
            if (nxt < ks_end) issue(nxt, buf ^ 1);
            WSTAMP(7);                                                 // next stage issued
            // ring form: in the first step(s) of chunk c (c >= 1) the plan of chunk c + 1 replaces chunk c - 1's (whose last step was
            // issued two iterations ago); the loads fly under this step's MFMAs, the entries are first read 6-7 barriers from here
            const int li = ks - ks_begin;
            const int lph = li & (WGRAD_RING_STEPS - 1);
            // row lists: the chunk's positions come from the list (step 0 of the chunk before), their plan entries one step later
            const bool fetch_rows = p.plan_ring && p.row_count && li >= WGRAD_RING_STEPS && lph == 0;
            const bool refill = p.plan_ring && li >= WGRAD_RING_STEPS && lph == (p.row_count ? 1 : 0);
            const int c1 = li / WGRAD_RING_STEPS + 1;
            if (fetch_rows) {
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    const int m = (ks_begin + c1 * WGRAD_RING_STEPS) * KB + t + NT * u;
                    rrow[u] = m < nrows ? p.row_pos[m] : -1;
                }
            }
            int rpo[RPT];
            unsigned rpm[RPT];
            if (refill) {
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    if (p.row_count) {
                        rpo[u] = rrow[u] >= 0 ? p.plan_off[rrow[u]] : 0;
                        rpm[u] = rrow[u] >= 0 ? p.plan_mask[rrow[u]] : 0u;
                    } else {
                        const int m = (ks_begin + c1 * WGRAD_RING_STEPS) * KB + t + NT * u;
                        rpo[u] = m < mpad ? p.plan_off[m] : 0;
                        rpm[u] = m < mpad ? p.plan_mask[m] : 0u;
                    }
                }
            }
            compute(buf);
            if (refill) {
                const int slot = (c1 & 1) * (RING / 2);
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    lplan_off[slot + t + NT * u] = rpo[u];
                    lplan_mask[slot + t + NT * u] = rpm[u];
                    if (p.row_count) lrow[slot + t + NT * u] = rrow[u];
                }
            }
            WSTAMP(8);                                                 // 32 MFMAs
            buf ^= 1;
            ks = nxt;
        }
    }
    WSTAMP(9);

    float* slab = p.slab + (size_t)split * p.Cout * p.Kpad;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int co = i0 + wi * WI + a * 16 + fg * 4 + r;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                int j = j0 + wj * WJ + b * 16 + fr;
                if (j < p.Kpad) slab[(size_t)co * p.Kpad + j] = acc[a][b][r];
            }
        }
    WSTAMP(10);                                                    // slab stored
#ifdef WGRAD_STAMPS
    if (wave == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 255; i += 64) p.dbg[(size_t)blockIdx.x * 256 + i] = i < n_stamp ? stl[i] : 0;
        if (lane == 0) p.dbg[(size_t)blockIdx.x * 256 + 255] = n_stamp;
    }
#endif
}

// ================================================================================================ stem weight gradient
// Weight gradient of a 2D convolution with 4 stored input channels and stride 2 (conv1 of the ResNet trunk, mv_cnn.py:44):
//   dW[co][kh][kw][ci] = sum_pos dOut[pos][co] * In[2 oh - pad + kh][2 ow - pad + kw][ci]
// Through conv_wgrad_kernel the im2col operand is an 8-byte gather per (position, tap): 61-85 us at the bench shape for 20 GFLOP.
// As in conv_stem_kernel (conv_igemm.hip) one kernel ROW (kw = 0..7, ci = 0..3) is one 32-wide slice of K: the input rows of a
// tile of two output rows are staged once in an LDS slab (8 B per pixel, borders zero), the dOut tile next to it in the natural
// [position][channel] layout, and BOTH MFMA operands are read transposed with ds_read_b64_tr_b16 (positions are the contraction
// index): the A fragment from the dOut tile as in the kernels above, the B fragment straight from the slab - row = position ow
// (16 B apart at stride 2), 16 columns = 4 pixels x 4 channels starting at pixel 2 ow + 4 nt.  The 2 KH (kernel row, pixel quad)
// slices are dealt to the four waves; a persistent workgroup keeps its 64 x KH x 32 partial sums in registers over all its tiles and
// writes ONE fp32 slab, which tri_wgrad_reduce_grouped sums (kw padded to 8: TriWgradReduce.kw_real).
struct StemWgradArgs {
    const void* in; const void* dout; float* slab;
    int B, IH, IW, OH, OW, KW, ph, pw;
    int TH, slab_rows, row_bytes, groups, tiles_per_img, ntiles, h_abl;
    int unpiped;                                                   // A/B switch TRICOLO_STEM_WGRAD_PIPE=0: the unpipelined tile loop
    // BNF instantiation (tri_conv_stem_wgrad_bn): dout is not read - the gradient w.r.t. the conv output is formed while the tile is
    // staged, from the conv output y, the max-pool's winning-tap map / pooled gradient and the BatchNorm-backward coefficients
    const void* y; const uint8_t* arg; const void* dpool;
    const float *c1, *c2, *c3, *rs, *rb;
};
// dy of 8 channels (octet `piece`) of the 2x2 block (bh, bw) of stem output positions = what tri_maxpool_bn_bwd_apply stores there
// (bn_pool.hip stem_route_2x2 + stem_bwd_apply_kernel: same routing and summation order, same rounding of the routed gradient, same
// ReLU mask expression, same FMA chain), packed in the storage type: store(k, h, 4 channels) receives channels 4 h .. 4 h + 3 of
// position (2 bh + k / 2, 2 bw + k % 2).  The block's pixels
// can only have won in the four windows (bh..bh+1, bw..bw+1): 4 + 8 loads, all issued before the first use.
// co = [5][64] floats in LDS: c1, c2, c3, relu scale, relu shift.
__device__ __forceinline__ bool stem_wgrad_unpiped(const StemWgradArgs& p) { return p.unpiped != 0; }
template <typename AT>
__device__ __forceinline__ void stem_dy_load(const StemWgradArgs& p, int img, int bh, int bw, int piece, uint4 (&yr)[4], uint4 (&dr)[2][2],
                                             uint2 (&ar)[2][2]) {
    const int Ho = p.OH >> 1, Wo = p.OW >> 1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        yr[k] = *(const uint4*)((const AT*)p.y + (((size_t)img * p.OH + 2 * bh + (k >> 1)) * p.OW + 2 * bw + (k & 1)) * 64 + piece * 8);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            ar[u][v] = make_uint2(0xffffffffu, 0xffffffffu);
            dr[u][v] = make_uint4(0u, 0u, 0u, 0u);
            if (bh + u < Ho && bw + v < Wo) {
                const size_t o = (((size_t)img * Ho + bh + u) * Wo + bw + v) * 64 + piece * 8;
                ar[u][v] = *(const uint2*)(p.arg + o);
                dr[u][v] = *(const uint4*)((const AT*)p.dpool + o);
            }
        }
}
template <typename AT, typename STORE>
__device__ __forceinline__ void stem_dy_form(const uint4 (&yr)[4], const uint4 (&dr)[2][2], const uint2 (&ar)[2][2], int piece, const float* co,
                                             STORE&& store) {
    // two halves of four channels, each stored before the next is formed (all eight at once do not fit the kernel's registers)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        AT res[4][4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int ch = h * 4 + c4;
            float d[2][2];
            unsigned a[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const AT* dp = (const AT*)&dr[u][v];
                    d[u][v] = (float)dp[ch];
                    a[u][v] = ((h ? ar[u][v].y : ar[u][v].x) >> (8 * c4)) & 255u;
                }
#define TRI_PK(U, V, TAP) (a[U][V] == (TAP) ? d[U][V] : 0.f)
            float g[4];
            g[0] = TRI_PK(0, 0, 4u);
            g[1] = TRI_PK(0, 0, 5u) + TRI_PK(0, 1, 3u);
            g[2] = TRI_PK(0, 0, 7u) + TRI_PK(1, 0, 1u);
            g[3] = (TRI_PK(0, 0, 8u) + TRI_PK(0, 1, 6u)) + (TRI_PK(1, 0, 2u) + TRI_PK(1, 1, 0u));
#undef TRI_PK
            const float* cc = co + piece * 8 + ch;
            const float k1 = cc[0], k2 = cc[64], k3 = cc[128], rs = cc[192], rb = cc[256];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float yv = (float)((const AT*)&yr[k])[ch];
                float gv = Act<AT>::rnd(g[k]);
                gv = __fmaf_rn(yv, rs, rb) > 0.f ? gv : 0.f;
                res[k][c4] = (AT)fp32_rounded(__fmaf_rn(k3, yv, __fmaf_rn(k1, gv, k2)));
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) store(k, h, *(const uint2*)res[k]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <typename AT, typename STORE>
__device__ __forceinline__ void stem_dy_block(const StemWgradArgs& p, int img, int bh, int bw, int piece, const float* co, STORE&& store) {
    uint4 yr[4], dr[2][2];
    uint2 ar[2][2];
    stem_dy_load<AT>(p, img, bh, bw, piece, yr, dr, ar);
    stem_dy_form<AT>(yr, dr, ar, piece, co, store);
}
#define STEM_WG_MAXG 8                                             // 32-position groups per tile (2 rows of <= 128 outputs)
template <int KH, typename AT, bool BNF = false>
__global__ __launch_bounds__(256, 2) void conv_stem_wgrad_kernel(const StemWgradArgs p) {
    typedef Mma<typename OpOf<AT>::E> MM;
    typedef typename MM::v8 v8;
    constexpr int NPAIR = 2 * KH;                                  // (kernel row, pixel quad) slices of K, 16 columns each
    constexpr int PPW = (NPAIR + 3) / 4;                           // slices per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fg = lane >> 4, fqq = fr >> 2, fp = fr & 3;
    const int OW = p.OW, OH = p.OH, IW = p.IW, IH = p.IH;
    const int npos = p.TH * OW;                                    // positions per tile (multiple of 32)
    char* const ytile = smem;                                      // [npos][64] 16-bit, nat_off<128> per 32-row group
    char* const slab = smem + (size_t)npos * 128;
    const int slab_bytes = p.slab_rows * p.row_bytes;

    // slab byte offset of this lane's two position rows (8 fg + fqq, + 4) in every 32-position group, pixel quarter fp included
    int poff[STEM_WG_MAXG][2];
#pragma unroll
    for (int g = 0; g < STEM_WG_MAXG; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pos = g * 32 + 8 * fg + 4 * h + fqq;
            const int r = pos / OW, ow = pos - r * OW;
            poff[g][h] = (2 * r) * p.row_bytes + (2 * ow + fp) * 8;
        }
    // loads of a tile: dOut rows (16 B pieces) and slab pixel pairs; which piece a thread moves never changes
    constexpr int YLD = STEM_WG_MAXG * 32 * 8 / 256, XLD = 4;      // per-thread maxima
    const int ld_per_row = IW >> 1, nxl = p.slab_rows * ld_per_row;
    int x_row[XLD], x_goff[XLD], x_loff[XLD];
#pragma unroll
    for (int u = 0; u < XLD; ++u) {
        const int e = t + u * 256;
        const int srow = e / ld_per_row, xp = e - srow * ld_per_row;
        x_row[u] = e < nxl ? srow : -(1 << 20);
        x_goff[u] = (srow * IW + 2 * xp) * 8;
        x_loff[u] = srow * p.row_bytes + (2 * xp + p.pw) * 8;
    }
    for (int i = t * 16; i < slab_bytes; i += 256 * 16) *(uint4*)(slab + i) = make_uint4(0u, 0u, 0u, 0u);   // border chunks stay zero
    float* const co = (float*)(slab + slab_bytes);                 // BNF: [5][64] BatchNorm-backward / ReLU coefficients
    if (BNF) {
        const float* src[5] = {p.c1, p.c2, p.c3, p.rs, p.rb};
        for (int i = t; i < 5 * 64; i += 256) co[i] = src[i >> 6][i & 63];
    }

    f32x4 acc[PPW][4];
#pragma unroll
    for (int j = 0; j < PPW; ++j)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[j][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (round 4: XCD-contiguous runs of tiles - xcd_remap - would keep the input rows neighbouring tiles share in one L2; measured 74 -> 78 us)
    const bool piped = BNF && (OW >> 1) * 8 <= 256 && !stem_wgrad_unpiped(p);
    if (BNF && piped) {
        // BNF with at most one (2x2 block, channel octet) item per thread (output rows of <= 64 positions): the NEXT tile's y / tap map / pooled
        // gradient / input pixels are requested before this tile's MFMAs and formed into dy after them - the tile loop below waits out a
        // memory round trip per tile with only the other resident workgroup to cover it.  LDS-only barriers: __syncthreads() would wait for
        // the loads just issued.
        const int nitem = (OW >> 1) * 8;
        const int piece = t & 7, bw = t >> 3;
        uint4 pyr[4], pdr[2][2], px[XLD];
        uint2 par[2][2];
        auto prefetch = [&](int tile_) {
            const int img_ = tile_ / p.tiles_per_img, oh0_ = (tile_ - img_ * p.tiles_per_img) * p.TH;
            if (t < nitem) stem_dy_load<AT>(p, img_, oh0_ >> 1, bw, piece, pyr, pdr, par);
            const int iy0 = oh0_ * 2 - p.ph;
            const char* xsrc = (const char*)p.in + ((size_t)img_ * IH + iy0) * IW * 8;
#pragma unroll
            for (int u = 0; u < XLD; ++u) {
                px[u] = make_uint4(0u, 0u, 0u, 0u);
                if (x_row[u] >= 0 && (unsigned)(iy0 + x_row[u]) < (unsigned)IH) px[u] = *(const uint4*)(xsrc + x_goff[u]);
            }
        };
        auto lds_barrier = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        int tile = blockIdx.x;
        if (tile < p.ntiles) prefetch(tile);
        for (; tile < p.ntiles; tile += gridDim.x) {
            lds_barrier();                                         // the previous tile's reads (and the zero fill / coefficients) are done
            if (t < nitem)
                stem_dy_form<AT>(pyr, pdr, par, piece, co, [&](int k, int h, uint2 v) {
                    const int row = (k >> 1) * OW + 2 * bw + (k & 1);
                    *(uint2*)(ytile + (row >> 5) * 4096 + nat_off<128>(row & 31, piece * 16 + h * 8)) = v;
                });
#pragma unroll
            for (int u = 0; u < XLD; ++u)
                if (x_row[u] >= 0) {
                    *(uint2*)(slab + x_loff[u]) = make_uint2(px[u].x, px[u].y);
                    *(uint2*)(slab + x_loff[u] + 8) = make_uint2(px[u].z, px[u].w);
                }
            if (tile + (int)gridDim.x < p.ntiles) prefetch(tile + gridDim.x);
            lds_barrier();
#pragma unroll
            for (int g = 0; g < STEM_WG_MAXG; ++g) {
                if (g >= p.groups) break;
                v8 af[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) af[ct] = tr_frag<128, v8>(ytile + g * 4096, ct * 16, fg, fqq, fp);
#pragma unroll
                for (int j = 0; j < PPW; ++j) {
                    const int pair = wave + 4 * j;                 // wave-uniform
                    if (pair < NPAIR) {
                        const int kh = pair >> 1, nt = pair & 1;
                        const int o = kh * p.row_bytes + nt * 32;
                        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(slab + poff[g][0] + o));
                        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(slab + poff[g][1] + o));
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        const s16x8 rr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        const v8 bf = __builtin_bit_cast(v8, rr);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) acc[j][ct] = MM::mma(af[ct], bf, acc[j][ct]);
                    }
                }
            }
        }
    } else
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int img = tile / p.tiles_per_img, oh0 = (tile - img * p.tiles_per_img) * p.TH;
        __syncthreads();                                           // the previous tile's reads (and the zero fill) are done
        {   // dOut tile: position row e / 8, 16-byte piece e % 8
            const char* ysrc = (const char*)p.dout + ((size_t)img * OH + oh0) * OW * 128;
            const int live = min(p.TH, OH - oh0) * OW;             // rows past the image: zeros
            if (BNF) {                                             // one (2x2 position block, channel octet) per thread and pass; TH == 2
                const int nitem = (OW >> 1) * 8;
                for (int e = t; e < nitem; e += 256) {
                    const int piece = e & 7, bw = e >> 3;
                    stem_dy_block<AT>(p, img, oh0 >> 1, bw, piece, co, [&](int k, int h, uint2 v) {
                        const int row = (k >> 1) * OW + 2 * bw + (k & 1);
                        *(uint2*)(ytile + (row >> 5) * 4096 + nat_off<128>(row & 31, piece * 16 + h * 8)) = v;
                    });
                }
            } else {
#pragma unroll
                for (int u = 0; u < YLD; ++u) {
                    const int e = t + u * 256;
                    const int row = e >> 3, piece = e & 7;
                    if (row < npos) {
                        uint4 v = make_uint4(0u, 0u, 0u, 0u);
                        if (row < live) v = *(const uint4*)(ysrc + (size_t)row * 128 + piece * 16);
                        *(uint4*)(ytile + (row >> 5) * 4096 + nat_off<128>(row & 31, piece * 16)) = v;
                    }
                }
            }
            const int iy0 = oh0 * 2 - p.ph;
            const char* xsrc = (const char*)p.in + ((size_t)img * IH + iy0) * IW * 8;
#pragma unroll
            for (int u = 0; u < XLD; ++u)
                if (x_row[u] >= 0) {
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if ((unsigned)(iy0 + x_row[u]) < (unsigned)IH) v = *(const uint4*)(xsrc + x_goff[u]);
                    *(uint2*)(slab + x_loff[u]) = make_uint2(v.x, v.y);
                    *(uint2*)(slab + x_loff[u] + 8) = make_uint2(v.z, v.w);
                }
        }
        __syncthreads();
        if (p.h_abl & 1) continue;
#pragma unroll
        for (int g = 0; g < STEM_WG_MAXG; ++g) {
            if (g >= p.groups) break;
            v8 af[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) af[ct] = tr_frag<128, v8>(ytile + g * 4096, ct * 16, fg, fqq, fp);
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int pair = wave + 4 * j;                     // wave-uniform
                if (pair < NPAIR) {
                    const int kh = pair >> 1, nt = pair & 1;
                    const int o = kh * p.row_bytes + nt * 32;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(slab + poff[g][0] + o));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(slab + poff[g][1] + o));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 rr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const v8 bf = __builtin_bit_cast(v8, rr);
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[j][ct] = MM::mma(af[ct], bf, acc[j][ct]);
                }
            }
        }
    }
    // one slab per workgroup: [64][KH * 32], column kh * 32 + nt * 16 + fr = (kh, kw = 4 nt + fr / 4, ci = fr % 4)
    float* out = p.slab + (size_t)blockIdx.x * 64 * (KH * 32);
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pair = wave + 4 * j;
        if (pair < NPAIR)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(ct * 16 + fg * 4 + r) * (KH * 32) + pair * 16 + fr] = acc[j][ct][r];
    }
}

// ================================================================================================ 64 -> 64 channel 3x3 weight gradient
// layer1 of the ResNet trunk (four 3x3 / 1 / pad 1 convolutions 64 -> 64, mv_cnn.py:44) has the most positions per weight of the
// trunk and the smallest dW (64 x 576): through conv_wgrad_dma_kernel<64,128> its K = 576 is cut into 4.5 column tiles, each
// re-reading the dOut tile and re-gathering its taps (39 us a layer at the bench shape, 525 us at 12 x 224^2).  Here, as in the
// stem kernel above, a tile of TR whole image rows is staged ONCE - the input rows with a one-pixel halo in an LDS slab, the dOut
// rows next to it - and ALL nine taps are formed from the slab: tap (ky, kx), channel quarter cq is the B fragment read (transposed,
// ds_read_b64_tr_b16) at pixel (r + ky, x + kx), bytes 32 cq .. of its 128-byte channel row.  The 36 (tap, quarter) column tiles are
// dealt to the four waves (9 each x 4 output-channel tiles = 144 accumulator registers); per 32 positions a wave issues 26
// transposed reads for 36 MFMAs.  A persistent workgroup keeps its partial dW over all its tiles and writes ONE [64][576] fp32 slab
// (standard k order: tri_wgrad_reduce_grouped sums them).
// Slab: pixel (slab row s, column c) at ((s * (W + 2) + c) * 128) bytes, its four 32-byte channel quarters XOR-swizzled by
// (pixel & 3) so that the four position rows of a transposed read hit different banks.
struct C64WgradArgs {
    const void* in; const void* dout; float* slab;
    int B, H, W, TR, groups, tiles_per_img, ntiles, h_abl;
};
#define C64_MAXG 8                                                 // 32-position groups per tile (<= 256 positions)
template <typename AT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_c64_kernel(const C64WgradArgs p) {
    typedef Mma<typename OpOf<AT>::E> MM;
    typedef typename MM::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fg = lane >> 4, fqq = fr >> 2, fp = fr & 3;
    const int W = p.W, H = p.H, P = W + 2, TR = p.TR;
    const int npos = TR * W;
    char* const ytile = smem;                                      // [npos][64] 16-bit, nat_off<128> per 32-row group
    char* const slab = smem + (size_t)npos * 128;
    const int slab_px = (TR + 2) * P;

    // transposed-read rows of this lane in group 0: (row, column) of positions 8 fg + fqq and + 4; advanced by 32 positions per group
    int r0[2], x0[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pos = 8 * fg + 4 * h + fqq;
        r0[h] = pos / W; x0[h] = pos - r0[h] * W;
    }
    for (int i = t * 16; i < slab_px * 128; i += 256 * 16) *(uint4*)(slab + i) = make_uint4(0u, 0u, 0u, 0u);   // left / right halo stays zero

    f32x4 acc[9][4];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[j][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int img = tile / p.tiles_per_img, h0 = (tile - img * p.tiles_per_img) * TR;
        __syncthreads();                                           // the previous tile's reads (and the zero fill) are done
        {   // dOut tile: position row e / 8, 16-byte piece e % 8.  Loads go out in batches of FB per thread and are stored afterwards: written
            // as one load + one store per iteration, the ~18 iterations of a tile were 18 dependent round trips - ~18 us of latency in
            // front of 2 us of MFMAs (the kernel ran no faster than the grouped conv_wgrad_dma_kernel launches at the bench shape)
            constexpr int FB = 8;
            const char* ysrc = (const char*)p.dout + ((size_t)img * H + h0) * W * 128;
            const int ny = npos * 8;
            for (int e0 = t; e0 < ny; e0 += 256 * FB) {
                uint4 v[FB];
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const int e = e0 + u * 256;
                    v[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (e < ny) v[u] = *(const uint4*)(ysrc + (size_t)e * 16);
                }
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const int e = e0 + u * 256;
                    const int row = e >> 3, piece = e & 7;
                    if (e < ny) *(uint4*)(ytile + (row >> 5) * 4096 + nat_off<128>(row & 31, piece * 16)) = v[u];
                }
            }
            // slab rows h0 - 1 .. h0 + TR (zeros outside the image), columns 1 .. W
            const char* xsrc = (const char*)p.in + ((size_t)img * H + h0 - 1) * W * 128;     // (may point before the image: only valid rows are read)
            const int nchunk = (TR + 2) * W * 8;
            for (int e0 = t; e0 < nchunk; e0 += 256 * FB) {
                uint4 v[FB];
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const int e = e0 + u * 256;
                    const int iy = h0 - 1 + (e >> 3) / W;
                    v[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (e < nchunk && (unsigned)iy < (unsigned)H) v[u] = *(const uint4*)(xsrc + (ptrdiff_t)e * 16);
                }
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const int e = e0 + u * 256;
                    const int piece = e & 7, px = e >> 3;
                    const int srow = px / W, x = px - srow * W;
                    const int pix = srow * P + x + 1;
                    if (e < nchunk) *(uint4*)(slab + pix * 128 + ((((piece >> 1) ^ (pix & 3)) << 5) | ((piece & 1) << 4))) = v[u];
                }
            }
        }
        __syncthreads();
        if (p.h_abl & 1) continue;
        int rr0 = r0[0], xx0 = x0[0], rr1 = r0[1], xx1 = x0[1];
#pragma unroll 1
        for (int g = 0; g < p.groups; ++g) {
            v8 af[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) af[ct] = tr_frag<128, v8>(ytile + g * 4096, ct * 16, fg, fqq, fp);
            const int pc0 = rr0 * P + xx0, pc1 = rr1 * P + xx1;    // slab pixel of tap (0, 0) for the two position rows
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int nt = wave + 4 * j;                       // column tile 0..35 = (tap, channel quarter), wave-uniform
                const int tap = nt >> 2, cq = nt & 3;
                const int ky = tap / 3, kx = tap - ky * 3;
                const int toff = ky * P + kx;
                const int p0 = pc0 + toff, p1 = pc1 + toff;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(slab + p0 * 128 + ((cq ^ (p0 & 3)) << 5) + fp * 8));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(slab + p1 * 128 + ((cq ^ (p1 & 3)) << 5) + fp * 8));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 rr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const v8 bf = __builtin_bit_cast(v8, rr);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[j][ct] = MM::mma(af[ct], bf, acc[j][ct]);
            }
            xx0 += 32; while (xx0 >= W) { xx0 -= W; ++rr0; }
            xx1 += 32; while (xx1 >= W) { xx1 -= W; ++rr1; }
        }
    }
    // one slab per workgroup: [64][576], column k = tap * 64 + cq * 16 + fr
    float* out = p.slab + (size_t)blockIdx.x * 64 * 576;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int nt = wave + 4 * j;
        const int k0 = (nt >> 2) * 64 + (nt & 3) * 16;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(ct * 16 + fg * 4 + r) * 576 + k0 + fr] = acc[j][ct][r];
    }
}

// ================================================================================================ kernel-row slab weight gradient
// 3x3 / stride 1 / pad 1 layers with 16-bit storage, Cin % 64 == 0, Cout % 64 == 0, image width 4 .. 64 (every resolution-keeping
// 3x3 conv of the ResNet trunk, mv_cnn.py:44; widths that do not divide 64 - 56 / 28 / 14 / 7 at 224^2 - run steps of floor(64 / W) rows
// with the tail of the 64-position tile zero: dOut rows out of range, their B rows parked on a zero border pixel).  conv_wgrad_dma_kernel treats the layer as an im2col GEMM: every 128-column tile
// of K = (tap, ci) re-gathers its input rows (each input pixel is fetched once per TAP) and re-reads its dOut tile (once per column tile) -
// PMC had the family move 3.6x its algorithmic bytes, and its waves spend as long issuing LDS-DMA pieces as multiplying (NOTES_wgrad.md).
// Here one workgroup owns ONE KERNEL ROW: dW[co0 .. co0 + CO_T)[ky][kx = 0..2][ci0 .. ci0 + 64).  A step is 64 output positions =
// 64 / W whole image rows; its operands are staged ONCE: the dOut tile [64][CO_T] (natural layout, nat_off swizzle) and the 64 / W input
// rows the kernel row ky needs (input row y + ky - 1 of each output row y, zeros outside the image) as a slab of (W + 2)-pixel rows with
// zero borders, 128 B (64 channels) per pixel.  The three taps kx are the same slab read shifted by one pixel, so per step the
// workgroup moves CO_T * 128 + ~10 KB for 3 x 64 x CO_T x 64 MACs: 2.9x (CO_T = 128) / 2.2x (64) fewer LDS-DMA bytes - and pieces to
// issue - per FLOP than the 128 x 128 im2col tile.  Both operands are read transposed (ds_read_b64_tr_b16): A from the dOut tile, B from
// the slab, whose four 32-byte channel quarters are XOR-swizzled by pixel bits so that the 8 position rows of a half-wave read hit
// 8 different bank groups for every tap shift (sw() below).  Waves 2 x 2: (CO_T / 2 channels) x (6 of the 12 (kx, quarter) column tiles).
// Output: fp32 slabs [split][Cout][9 Cin] in the standard k order (tri_wgrad_reduce_grouped sums them), jobs of several layers per launch.
struct KrowArgs {
    const void* in; const void* dout; float* slab;
    unsigned in_bytes, dout_bytes;
    int NH, H, W, Cin, Cout, Kpad;   // NH = images x H; H, W = OUTPUT grid of an image
    int Hin, Win;                    // input grid (= H, W for stride 1; 2 H, 2 W for the stride-2 form)
    int rps, P, npiece;          // image rows per step, slab row pitch in pixels, 1 KiB DMA pieces per slab (8 pixels each)
    int nsteps, steps_per_split, nsplits, ntiles, ci_chunks;
    FastDiv dH, dP, dW;
};
struct KrowJobs {
    KrowArgs d[WGRAD_JOBS_MAX];
    int first_block[WGRAD_JOBS_MAX + 1];
    int n;
};
static_assert(sizeof(KrowJobs) <= 4096, "the job table travels in the kernel arguments (4 KB)");
#define KROW_SLAB_PX 96                                            // stride 1: rps * (W + 2) <= 96 (W = 4: 16 rows of 6 pixels)
#define KROW_SLAB_PX_S2 160                                        // stride 2: rps * (2 W + 2) <= 160 (W = 4: 16 rows of 10 pixels)
// Stride-2 form (the first conv of layer2 / 3 / 4, mv_cnn.py:44): output position (y, x) reads input row 2 y + ky - 1, column 2 x + kx - 1.
// A slab row holds the WHOLE input row (2 W + 2 columns incl. the borders) DE-INTERLEAVED - even columns first, then the odd ones (the
// DMA's per-lane source address does it for free) - so that tap kx is again the same plane read shifted: column 2 x + kx sits at plane
// (kx & 1), index x + (kx >> 1), and the eight positions of a half-wave read eight CONSECUTIVE pixels exactly as at stride 1.
// quarter swizzle of slab pixel (row r, column xs): bit 0 separates pixels two apart, bit 1 the two position octets of a half-wave
// (8 pixels apart in one row for W >= 16, the next row for W = 8, two rows on for W = 4)
// (widths that do not divide 64 - 56 / 28 / 14 / 7 of the 224^2 configuration - keep the rule of the next power of two: a read whose
//  second octet wraps into the next image row then has some 2-way conflicts; any swizzle is CORRECT as long as fill and read agree)
__device__ __forceinline__ int krow_sw(int r, int xs, int W) {
    const int hb = W > 8 ? (xs >> 3) & 1 : (W > 4 ? r & 1 : (r >> 1) & 1);
    return ((xs >> 1) & 1) | (hb << 1);
}
template <int CO_T, typename E, int S = 1>
__global__ __launch_bounds__(256, 2) void conv_wgrad_krow_kernel(const KrowJobs jobs) {
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    constexpr int KB = 64;
    constexpr int XROW = CO_T * 2, X_BYTES = KB * XROW;
    // S = 1 / 2: the stride of every job of the launch; S = 0: per job (p.Hin != p.H), the big slab for all - layer2 / 3 / 4's
    // stride-2 first convs then ride in the launch of the stride-1 layers (a launch of their own cut them into 4-57 short splits:
    // 53 MB of slabs for 21.7 GFLOP)
    constexpr int SLAB_BYTES = (S == 1 ? KROW_SLAB_PX : KROW_SLAB_PX_S2) * 128, STAGE = X_BYTES + SLAB_BYTES;
    constexpr int NSP = SLAB_BYTES / 4096;                         // slab pieces per wave (3 / 5)
    constexpr int XRPI = 1024 / XROW, XNI = KB / (4 * XRPI);       // dOut rows per wave-instruction, instructions per wave and step
    constexpr int TM = CO_T / 32;                                  // 16-channel row tiles per wave
    static_assert((4 * XRPI) % 16 == 0 && XNI >= 1, "lane -> chunk map must not depend on the instruction");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[ji + 1]) ++ji;
    const KrowArgs& p = jobs.d[ji];
    const int bid = (int)blockIdx.x - jobs.first_block[ji];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ntiles = p.ntiles;
    int split, tile;
    if (p.nsplits >= 16) {                                         // XCD pinning of the position splits, see conv_wgrad_kernel
        const int grp = bid / (8 * ntiles), rem = bid - grp * 8 * ntiles;
        split = grp * 8 + (rem & 7);
        tile = rem >> 3;
    } else {
        // few splits (layer3 / layer4: one to three): every XCD takes a CONTIGUOUS run of the (split, tile) list - the job's block range
        // starts at a multiple of 8, so bid % 8 is the XCD - i.e. the tiles of one dOut channel tile and neighbouring input-channel
        // chunks.  Dealt round-robin (the first version) every XCD streamed the whole dOut and input tensors of the layer through its
        // 4 MB L2: PMC had the launch move 975 MB of fabric traffic per step for ~130 MB of operands (profiles/r4/README.md).
        const int nb = p.nsplits * ntiles;
        if (bid >= nb) return;
        const int lid = xcd_remap(bid, nb);
        split = lid / ntiles;
        tile = lid - split * ntiles;
    }
    if (split >= p.nsplits) return;
    // tile = ((co tile * ci_chunks) + ci chunk) * 3 + ky: the three kernel rows of one (co, ci) block are neighbours (same dOut tile,
    // input rows one apart) and, with the mappings above, run on one XCD
    const int ky = tile % 3, rest = tile / 3;
    const int cic = rest % p.ci_chunks, cot = rest / p.ci_chunks;
    const int co0 = cot * CO_T, ci0 = cic * 64;
    const int ks_begin = split * p.steps_per_split, ks_end = min(p.nsteps, ks_begin + p.steps_per_split);
    const int W = p.W, P = p.P, H = p.H, Cin = p.Cin, Cout = p.Cout;
    const int st = S ? S : (p.Hin != p.H ? 2 : 1);                 // stride of this job
    const int wi = wave >> 1, wj = wave & 1;
    const int fr = lane & 15, fg = lane >> 4, fqq = fr >> 2, fp = fr & 3;

    // ---- DMA lane constants.  dOut tile: as conv_wgrad_dma_kernel's X operand
    const int xrow0 = wave * XRPI + lane / (XROW / 16), xsl = lane % (XROW / 16);
    const int xchunk = ((((xsl >> 1) ^ nat_sw<XROW>(xrow0)) & (XROW / 32 - 1)) << 1) | (xsl & 1);
    const unsigned xoff = (unsigned)((xrow0 * Cout + co0 + xchunk * 8) * 2);
    const unsigned xstep = (unsigned)(4 * XRPI * Cout * 2);
    const unsigned step_bytes = (unsigned)(p.rps * W) * (unsigned)(Cout * 2);   // dOut bytes between two steps
    // slab: piece pi = wave + 4 i covers slab pixels 8 pi .. 8 pi + 7; this lane moves the 16-byte chunk (lane & 7) of pixel 8 pi + lane / 8
    int s_r[NSP], s_col[NSP];                                      // slab row of the lane's pixel, byte offset inside an input row (< 0: zeros)
#pragma unroll
    for (int i = 0; i < NSP; ++i) {
        const int pi = wave + 4 * i;
        const int q = 8 * pi + (lane >> 3), c = lane & 7;
        const int r = (int)fdiv((uint32_t)q, p.dP), xs = q - r * P;  // LDS pixel xs of slab row r
        const int cq = ((c >> 1) ^ krow_sw(r, xs, W)) & 3;
        // slab COLUMN of that pixel: itself at stride 1; de-interleaved at stride 2 (even columns first); input x = column - 1
        const int col = st == 1 ? xs : (xs >= P / 2 ? 2 * (xs - P / 2) + 1 : 2 * xs);
        s_r[i] = r;
        s_col[i] = (col >= 1 && col <= p.Win && r < p.rps) ? ((col - 1) * Cin + ci0 + cq * 16 + (c & 1) * 8) * 2 : -1;
    }
    // ---- B fragment addresses (per lane, the same for every step): position 32 h + 8 fg + 4 half + fqq of the step, column tile b
    int boff[2][2][6];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int pos = 32 * h + 8 * fg + 4 * hf + fqq;
            const bool live = pos < p.rps * W;                       // dead tail of the tile (W does not divide 64): its dOut rows are zeros;
            const int r = live ? (int)fdiv((uint32_t)pos, p.dW) : 0; // park its B rows on pixel 0, a zero border (0 x garbage could be a NaN)
            const int x = live ? pos - r * W : 0;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int tt = 6 * wj + b, kx = live ? tt >> 2 : 0, cq = tt & 3;
                const int xs = st == 1 ? x + kx : (kx & 1) * (P / 2) + x + (kx >> 1);    // LDS pixel of slab column x + kx / 2 x + kx
                boff[h][hf][b] = (r * P + xs) * 128 + (((cq ^ krow_sw(r, xs, W)) & 3) << 5) + fp * 8;
            }
        }

    f32x4 acc[TM][6];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const v4i rsrc = make_rsrc_words(p.in, p.in_bytes);
    const v4i xrsrc = make_rsrc_words(p.dout, p.dout_bytes);         // rows past the tensor arrive as zeros
    const unsigned lds0 = lds_addr(smem) + wave * 1024;
    const int row_bytes = p.Win * Cin * 2;                           // one INPUT image row

    const int npos = p.rps * W;                                      // live positions of a step (64 when W divides 64)
    auto issue_x = [&](int ks, int buf) {
        const unsigned xb = lds0 + buf * STAGE;
        const unsigned xbase = xoff + (unsigned)ks * step_bytes;
#pragma unroll
        for (int i = 0; i < XNI; ++i)                                // tile rows past the step's positions: zeros (out-of-range fetch)
            dma16_async(xrsrc, xb + i * 4096, xrow0 + 4 * XRPI * i < npos ? (int)(xbase + i * xstep) : (int)0x80000000);
    };
    auto issue_slab = [&](int ks, int buf) {
        const unsigned sb = lds0 + buf * STAGE + X_BYTES;
        int vo[NSP];
#pragma unroll
        for (int i = 0; i < NSP; ++i) {
            const int R = ks * p.rps + s_r[i];                       // global image row of this slab row's OUTPUT row
            const int n = (int)fdiv((uint32_t)R, p.dH), yy = st * (R - n * H) + ky - 1;       // input row inside image n
            const bool ok = s_col[i] >= 0 && R < p.NH && (unsigned)yy < (unsigned)p.Hin;
            vo[i] = ok ? (n * p.Hin + yy) * row_bytes + s_col[i] : (int)0x80000000;
        }
#pragma unroll
        for (int i = 0; i < NSP; ++i)
            if (wave + 4 * i < p.npiece) dma16_async(rsrc, sb + i * 4096, vo[i]);
    };
    auto compute = [&](int buf, int h) {
        const char* xb = smem + buf * STAGE + h * 32 * XROW;
        const char* sb = smem + buf * STAGE + X_BYTES;
        v8 af[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[a] = tr_frag<XROW, v8>(xb, wi * (CO_T / 2) + a * 16, fg, fqq, fp);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sb + boff[h][0][b]));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sb + boff[h][1][b]));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 rr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const v8 bf = __builtin_bit_cast(v8, rr);
#pragma unroll
            for (int a = 0; a < TM; ++a) acc[a][b] = MM::mma(af[a], bf, acc[a][b]);
        }
    };

    if (ks_begin < ks_end) {
        issue_x(ks_begin, 0);
        issue_slab(ks_begin, 0);
        int buf = 0;
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of stage `buf` have landed
            __builtin_amdgcn_s_barrier();                                // ... everyone's have, and all reads of buf ^ 1 are done
            asm volatile("" ::: "memory");
            const bool more = ks + 1 < ks_end;
            if (more) issue_x(ks + 1, buf ^ 1);                          // the next step's pieces go out between the two halves' MFMAs
#pragma unroll
            for (int h = 0; h < 2; ++h) {                                // (unrolled: boff[h] is a compile-time index)
                if (h == 1 && more) issue_slab(ks + 1, buf ^ 1);
                compute(buf, h);
            }
            buf ^= 1;
        }
    }
    float* slab = p.slab + (size_t)split * Cout * p.Kpad;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wi * (CO_T / 2) + a * 16 + fg * 4 + r;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int tt = 6 * wj + b, kx = tt >> 2, cq = tt & 3;
                slab[(size_t)co * p.Kpad + (ky * 3 + kx) * Cin + ci0 + cq * 16 + fr] = acc[a][b][r];
            }
        }
}

// Overflow note of the reduces (round 6): the reduce holds every FINAL gradient element in a register right before its only store, so the
// optimizer's non-finite scan of these tensors (adam_guard_seg_kernel: one more read of 64 of the step's 74 MB of gradients, on the serial
// tail of the step) is free here.  `note` = the optimizer's device record (misc.hip: [0] applied steps, [2] attempt flagged, [3] steps
// skipped whole); a workgroup that stored an inf / NaN flags the current attempt exactly as the guard kernel does.
__device__ __forceinline__ unsigned wgrad_nonfinite(float x) { return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u ? 1u : 0u; }
__device__ __forceinline__ void wgrad_note_bad(unsigned bad, int* note) {
    if (note && __any(bad) && (threadIdx.x & 63) == 0) atomicMax(note + 2, note[0] + note[3] + 1);
}
// dw[co*s_co + tap*s_tap + ci*s_ci] = sum_split slab[split][co][tap*cin_stored + ci]   (ci < cin_real)
// Each thread owns 4 consecutive k (one 16-byte load per split) of one co; a block is (256 / zlanes) such quads x zlanes
// split lanes: lane z sums splits z, z + zlanes, ... and lane 0 adds the partials in a fixed order (bitwise
// reproducible for a given layer shape).  zlanes grows with splits / outputs so that the layers with hundreds of splits
// and a tiny dW (stem, voxel level 0) still put a few hundred thousand loads in flight.
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ slab, int splits, int Cout, int Kpad, int ntaps, int cin_stored,
                                                   int cin_real, float* __restrict__ dw, long s_co, long s_tap, long s_ci, int zlanes,
                                                   float out_scale, unsigned block, float4* part, unsigned& bad, int kw_real = 0,
                                                   int kw_shift = 3) {
    const int kq = 256 / zlanes;                                 // quads per block
    const int ql = threadIdx.x % kq, zl = threadIdx.x / kq;
    const int K4 = (ntaps * cin_stored) >> 2;
    const long quad = (long)block * kq + ql;
    const bool live = quad < (long)Cout * K4;
    const int co = live ? (int)(quad / K4) : 0, k = live ? (int)(quad - (long)co * K4) * 4 : 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const float* src = slab + (size_t)co * Kpad + k;
        const size_t zs = (size_t)Cout * Kpad;
#pragma unroll 4
        for (int z = zl; z < splits; z += zlanes) {
            float4 v = *(const float4*)(src + z * zs);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (zl == 0 && live) {
        for (int z = 1; z < zlanes; ++z) {
            float4 v = part[z * kq + ql];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        s.x *= out_scale; s.y *= out_scale; s.z *= out_scale; s.w *= out_scale;
        int tap = k / cin_stored;
        const int ci = k - tap * cin_stored;                         // cin_stored % 4 == 0: the quad stays inside one tap
        if (kw_real) {                                               // kernel rows padded to 8 (stem) / 4 (voxel level 0) taps in the slabs
            const int kh = tap >> kw_shift, kw = tap & ((1 << kw_shift) - 1);
            if (kw >= kw_real) return;
            tap = kh * kw_real + kw;
        }
        float* d = dw + co * s_co + tap * s_tap + ci * s_ci;
        if (s_ci == 1 && cin_real == cin_stored && (((uintptr_t)d) & 15) == 0) {
            *(float4*)d = s;
            bad |= wgrad_nonfinite(s.x) | wgrad_nonfinite(s.y) | wgrad_nonfinite(s.z) | wgrad_nonfinite(s.w);
        } else {
            if (ci + 0 < cin_real) { d[0] = s.x; bad |= wgrad_nonfinite(s.x); }
            if (ci + 1 < cin_real) { d[s_ci] = s.y; bad |= wgrad_nonfinite(s.y); }
            if (ci + 2 < cin_real) { d[2 * s_ci] = s.z; bad |= wgrad_nonfinite(s.z); }
            if (ci + 3 < cin_real) { d[3 * s_ci] = s.w; bad |= wgrad_nonfinite(s.w); }
        }
    }
}
// Row form (descriptor zlanes == 0) for the torchvision parameter layout [co][ci][tap] (s_tap == 1, s_ci == ntaps) with several
// taps: one output channel per block.  The slab row is in k = (tap, ci) order, the parameter row in (ci, tap) order: written straight
// from the quads above, every store is four scattered floats, `ntaps` apart - PMC counted 230 MB of writes per step for 54 MB of
// gradients (partial cache lines, written back more than once).  Here the summed row goes through LDS (tap-major, rows of cin + 1
// floats: bank = (tap + ci) mod 32) and leaves as ONE contiguous run of K floats.  Splits are summed in order 0, 1, ...
#define WGRAD_ROW_MAX 4640                                         // floats: 9 taps x (512 + 1)
__device__ __forceinline__ void wgrad_reduce_row(const float* __restrict__ slab, int splits, int Cout, int Kpad, int ntaps, int cin,
                                                 float* __restrict__ dw, long s_co, float out_scale, unsigned co, float* rowbuf, unsigned& bad) {
    const int K = ntaps * cin, K4 = K >> 2;
    const size_t zs = (size_t)Cout * Kpad;
    const float* src = slab + (size_t)co * Kpad;
    for (int q = threadIdx.x; q < K4; q += 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int z = 0; z < splits; ++z) {
            const float4 v = *(const float4*)(src + z * zs + q * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int k = q * 4, tap = k / cin, ci = k - tap * cin;     // cin % 4 == 0: the quad stays inside one tap
        float* d = rowbuf + tap * (cin + 1) + ci;
        s.x *= out_scale; s.y *= out_scale; s.z *= out_scale; s.w *= out_scale;
        d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w;
        bad |= wgrad_nonfinite(s.x) | wgrad_nonfinite(s.y) | wgrad_nonfinite(s.z) | wgrad_nonfinite(s.w);
    }
    __syncthreads();
    float* out = dw + (size_t)co * s_co;
    for (int j = threadIdx.x; j < K; j += 256) {
        const int ci = j / ntaps, tap = j - ci * ntaps;
        out[j] = rowbuf[tap * (cin + 1) + ci];
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, int Cout, int Kpad, int ntaps,
                                                           int cin_stored, int cin_real, float* __restrict__ dw, long s_co, long s_tap,
                                                           long s_ci, int zlanes, float out_scale) {
    __shared__ float4 part[WGRAD_ROW_MAX / 4];
    unsigned bad = 0;                                              // (this single-layer form carries no note: the optimizer scans its tensor)
    if (zlanes == 0) { wgrad_reduce_row(slab, splits, Cout, Kpad, ntaps, cin_stored, dw, s_co, out_scale, blockIdx.x, (float*)part, bad); return; }
    wgrad_reduce_block(slab, splits, Cout, Kpad, ntaps, cin_stored, cin_real, dw, s_co, s_tap, s_ci, zlanes, out_scale, blockIdx.x, part, bad);
}
// Grouped form: the reduces of up to TRI_WGRAD_GROUP_MAX layers in ONE launch (their partial kernels ran earlier into per-layer
// slabs - tri_conv_wgrad_partial).  A tower's backward then pays one reduce launch instead of one per layer (28 launches of
// 5-23 us each per step, every one of them on the critical path between a weight-gradient kernel and the next data gradient).
struct WgradGroup {
    TriWgradReduce d[TRI_WGRAD_GROUP_MAX];
    int first_block[TRI_WGRAD_GROUP_MAX + 1];
    int* note;                                                     // optional: the optimizer's overflow record (wgrad_note_bad)
};
__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(const WgradGroup g, int n) {
    __shared__ float4 part[WGRAD_ROW_MAX / 4];
    int i = 0;
    while (i + 1 < n && (int)blockIdx.x >= g.first_block[i + 1]) ++i;
    const TriWgradReduce& r = g.d[i];
    unsigned bad = 0;
    if (r.zlanes == 0)
        wgrad_reduce_row(r.slab, r.splits, r.Cout, r.Kpad, r.ntaps, r.cin_stored, r.dw, r.s_co, r.out_scale, blockIdx.x - g.first_block[i], (float*)part, bad);
    else
        wgrad_reduce_block(r.slab, r.splits, r.Cout, r.Kpad, r.ntaps, r.cin_stored, r.cin_real, r.dw, r.s_co, r.s_tap, r.s_ci, r.zlanes,
                           r.out_scale, blockIdx.x - g.first_block[i], part, bad, r.kw_real, r.kw_shift);
    wgrad_note_bad(bad, g.note);
}

static int ilog2_exact(int v) {
    for (int s = 0; s < 31; ++s) if ((1 << s) == v) return s;
    return -1;
}


// ---------------------------------------------------------------------------------------------------- gather plan
// Per output position: element offset of the origin input voxel (o*stride - pad) and per-axis tap validity bits
// (bit k of byte 0 / 1 / 2 = tap k along W / H / D reads inside the grid).  Depends only on the layer geometry, so it is
// built once per layer shape and reused by every step's wgrad (and can serve the forward gather as well).
__global__ void conv_plan_kernel(int M, int Mpad, int ID, int IH, int IW, int Cin, int OD, int OH, int OW, int KD, int KH, int KW,
                                 int stride, int pd, int ph, int pw, int* __restrict__ plan_off, unsigned* __restrict__ plan_mask) {
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Mpad) return;
    if (m >= M) { plan_off[m] = 0; plan_mask[m] = 0u; return; }
    int ow = m % OW; int r = m / OW;
    int oh = r % OH; r /= OH;
    int od = r % OD; int b = r / OD;
    int z0 = od * stride - pd, y0 = oh * stride - ph, x0 = ow * stride - pw;
    unsigned mk = 0;
    for (int k = 0; k < KW; ++k) if ((unsigned)(x0 + k) < (unsigned)IW) mk |= 1u << k;
    for (int k = 0; k < KH; ++k) if ((unsigned)(y0 + k) < (unsigned)IH) mk |= 1u << (8 + k);
    for (int k = 0; k < KD; ++k) if ((unsigned)(z0 + k) < (unsigned)ID) mk |= 1u << (16 + k);
    plan_off[m] = (((b * ID + z0) * IH + y0) * IW + x0) * Cin;
    plan_mask[m] = mk;
}

extern "C" size_t tri_conv_plan_bytes(const TriConvDesc* d) {
    long M = (long)d->B * d->OD * d->OH * d->OW;
    long Mpad = (M + 31) / 32 * 32;
    return (size_t)Mpad * 8;
}

extern "C" int tri_conv_plan_build(const TriConvDesc* d, void* plan, void* stream) {
    long M = (long)d->B * d->OD * d->OH * d->OW;
    long Mpad = (M + 31) / 32 * 32;
    if (d->KD > 8 || d->KH > 8 || d->KW > 8) { tri_set_error("conv plan: kernel extent > 8 unsupported"); return TRI_ERR_UNSUPPORTED; }
    conv_plan_kernel<<<(int)((Mpad + 255) / 256), 256, 0, (hipStream_t)stream>>>((int)M, (int)Mpad, d->ID, d->IH, d->IW, d->Cin, d->OD,
                                                                                 d->OH, d->OW, d->KD, d->KH, d->KW, d->stride, d->pad_d,
                                                                                 d->pad_h, d->pad_w, (int*)plan, (unsigned*)plan + Mpad);
    return tri_check_launch("tri_conv_plan_build");
}

#ifndef WGRAD_TARGET_BLOCKS
#define WGRAD_TARGET_BLOCKS 768
#endif
// tile shapes: 128x128 for wide layers; 64-row tiles take the whole K in one 256-column tile when it fits (the dOut
// operand is then read once, not once per j-tile: stem 7x7, voxel level 0, 1x1 down-samples), else 128 columns.
static int wgrad_target_blocks() {                             // tuning aid: TRICOLO_WGRAD_BLOCKS overrides the default
    static int v = -1;
    if (v < 0) v = WGRAD_TARGET_BLOCKS;
    return v;
}

static bool wgrad_dma_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_DMA"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

// act_fmt != 0 asks for the plan of the bf16-storage call; *dma is set when that call runs the LDS-DMA kernel
// (64-position steps), and steps_per_split is then in 64-position units.
static void wgrad_plan(const TriConvDesc* d, int act_fmt, int rowlist, int* BI, int* BJ_out, int* tiles, int* splits, int* steps_per_split,
                       int* Kpad, int* dma) {
    int ntaps = d->KD * d->KH * d->KW;
    *Kpad = (ntaps * d->Cin + 31) / 32 * 32;
    *BI = (d->Cout % 128 == 0 && *Kpad >= 128) ? 128 : 64;
    // 64-row tiles: K <= 256 is one 256-wide column tile of the register-staged kernel - unless the LDS-DMA kernel can take the
    // layer (16-bit storage, 8-channel pieces): then 128-wide tiles (layer2's 1x1/2 shortcut: K = 64, 39 -> ~8 us at the bench shape)
    const bool dma64 = act_fmt && d->Cin % 8 == 0 && d->Cout % 64 == 0 && !wgrad_dma_disabled();
    int BJ = *BI == 128 ? 128 : ((*Kpad <= 256 && !dma64) ? 256 : 128);
    *BJ_out = BJ;
    int it = (d->Cout + *BI - 1) / *BI, jt = (*Kpad + BJ - 1) / BJ;
    *tiles = it * jt;
    long M = (long)d->B * d->OD * d->OH * d->OW;
    *dma = (act_fmt && BJ == 128 && d->Cin % 8 == 0 && d->Cout % *BI == 0 && !wgrad_dma_disabled()) ? 1 : 0;
    const int unit = *dma ? 64 : 32;
    int steps = (int)((M + unit - 1) / unit);
    int max_by_steps = steps / 4 > 0 ? steps / 4 : 1;           // at least 4 k-steps per split
    int s, cap;
    if (act_fmt && *BI == 128 && wgrad_target_blocks() == WGRAD_TARGET_BLOCKS) {
        // 128x128 bf16-storage tiles (64 KiB of operand stages -> 2 workgroups per CU): at most ONE resident round of
        // workgroups, and not quite full (7/8 of the 512 slots).  More splits only add slab traffic (splits x Cout x K
        // fp32 written + re-read by the reduce) and a second, partly empty round; measured sweep in profiles/r1/README.md.
        const int fixed = (*dma ? 2 * 64 * (*BI * 2 + BJ * 2) : 2 * (32 * *BI * 2 + 32 * BJ * 2)) + 512;
        cap = (163840 / 2 - fixed) / ((*dma ? 512 : 256) * (rowlist ? 3 : 2) / 2);   // gather plan: 8 B per position of the split (+ 4 B: row list)
        if (cap > 96) cap = 96;
        s = 448 / *tiles;
    } else {
        cap = *dma ? 60 : 96;
        // register-staged kernels (fp32 storage, 4-channel layers): ~3 workgroups per CU; 64-row DMA tiles: 448 (same sweep)
        const int target = (*dma && wgrad_target_blocks() == WGRAD_TARGET_BLOCKS) ? 448 : wgrad_target_blocks();
        s = (target + *tiles - 1) / *tiles;
    }
    if (s > max_by_steps) s = max_by_steps;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    *steps_per_split = (steps + s - 1) / s;
    if (*steps_per_split > cap) *steps_per_split = cap;
    *splits = (steps + *steps_per_split - 1) / *steps_per_split;
}

int tri_internal_num_cus();                                         // conv_igemm.hip
static bool stem_wgrad_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_STEM_WGRAD"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
// geometry of conv_stem_wgrad_kernel; false when the layer does not qualify (it then runs conv_wgrad_kernel)
static bool stem_wgrad_geometry(const TriConvDesc* d, int act_fmt, StemWgradArgs* g, int* grid) {
    if (!act_fmt || stem_wgrad_disabled()) return false;
    if (d->KD != 1 || d->ID != 1 || d->pad_d != 0 || d->Cin != 4 || d->stride != 2 || d->Cout != 64) return false;
    if ((d->KH != 3 && d->KH != 5 && d->KH != 7) || d->KW > 8 || d->OW % 16 || d->IW % 2 || d->OW > 128) return false;
    if (d->pad_w < 0 || d->pad_h < 0 || d->IW + d->pad_w > 2 * d->OW + 6) return false;
    if ((long)d->B * d->IH * d->IW * 8 >= ((long)1 << 31) || (long)d->B * d->OH * d->OW * 128 >= ((long)1 << 31)) return false;
    int TH = d->OH >= 2 ? 2 : 1;
    if ((TH * d->OW) % 32) return false;
    g->B = d->B; g->IH = d->IH; g->IW = d->IW; g->OH = d->OH; g->OW = d->OW; g->KW = d->KW; g->ph = d->pad_h; g->pw = d->pad_w;
    g->TH = TH;
    g->slab_rows = (TH - 1) * 2 + d->KH;
    g->row_bytes = (d->OW + 3) * 16;
    g->groups = TH * d->OW / 32;
    g->tiles_per_img = (d->OH + TH - 1) / TH;
    g->ntiles = d->B * g->tiles_per_img;
    if (g->groups > STEM_WG_MAXG || g->slab_rows * (d->IW / 2) > 256 * 4) return false;
    const int slots = tri_internal_num_cus() * 2;
    *grid = g->ntiles < slots ? g->ntiles : slots;
    return true;
}

static bool krow_wgrad_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_KROW_WGRAD"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
static bool c64_wgrad_disabled() {
    static int v = -1;
    if (v < 0) v = 0;
    return v == 1;
}
// geometry of conv_wgrad_c64_kernel; false when the layer does not qualify
static bool c64_wgrad_geometry(const TriConvDesc* d, int act_fmt, C64WgradArgs* g, int* grid) {
    if (!act_fmt || c64_wgrad_disabled()) return false;
    if (d->KD != 1 || d->ID != 1 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_d != 0 || d->pad_h != 1 || d->pad_w != 1) return false;
    if (d->Cin != 64 || d->Cout != 64 || d->OH != d->IH || d->OW != d->IW) return false;
    if ((long)d->B * d->IH * d->IW * 128 >= ((long)1 << 31)) return false;
    const int H = d->IH, W = d->IW;
    // round 4: widths that divide 64 go to conv_wgrad_krow_kernel<64> (config 3, 32-wide layer1: 3.975 against 4.032 ms per step); this
    // kernel keeps the widths whose 64-position steps would carry a dead tail there (56-wide layer1 of 12 x 224^2: 20.39 against 20.66 ms)
    if (64 % W == 0 && W >= 4 && !krow_wgrad_disabled()) return false;
    int TR = 0;
    for (int tr = 1; tr <= H; ++tr)                                  // most positions per tile: whole rows of one image, multiple of 32, <= 256
        if (H % tr == 0 && (tr * W) % 32 == 0 && tr * W <= 32 * C64_MAXG) TR = tr;
    if (!TR) return false;
    if ((size_t)TR * W * 128 + (size_t)(TR + 2) * (W + 2) * 128 > 78 * 1024) return false;      // two workgroups per CU
    g->B = d->B; g->H = H; g->W = W; g->TR = TR; g->groups = TR * W / 32;
    g->tiles_per_img = H / TR; g->ntiles = d->B * g->tiles_per_img;
    // 256-512 per-workgroup slabs of 147 KB.  Round 2: paid from ~6 row tiles per CU on (per-GPU batch 64 of 6 x 128^2 5.03 -> 4.87 ms,
    // 12 x 224^2 25.3 -> 24.6 ms).  Round 3, against layer1's four weight gradients in ONE grouped conv_wgrad_dma_kernel launch: 6 tiles
    // per CU 4.43 (this kernel) against 4.36-4.38 ms (grouped), 42 tiles per CU 22.76 against 22.76 - so only the largest shapes kept it (12).
    // End of round 3, with the tile loads going out in batches (see the fill above): 42 tiles per CU 21.45 -> 20.9 ms (21.8 without this
    // kernel), 6 tiles per CU 4.10-4.12 against 4.14-4.15 ms (grouped) - back to 6; 3 tiles per CU (the bench shape) 2.95 against 2.81-2.84 ms:
    // 512 workgroups x 147 KB of slabs per layer are what it costs there
    static int min_per_cu = -1;
    if (min_per_cu < 0) min_per_cu = 6;
    if (g->ntiles < min_per_cu * tri_internal_num_cus()) return false;
    static int per_cu = -1;
    if (per_cu < 0) per_cu = 2;
    *grid = tri_internal_num_cus() * (per_cu < 1 ? 1 : per_cu);
    if (*grid > g->ntiles) *grid = g->ntiles;
    return true;
}

// geometry of conv_wgrad_krow_kernel (everything but pointers and the split plan); false when the layer does not qualify.
// *co_t = 128 / 64 (rows per workgroup tile), *tiles = workgroups per position split, *steps = 64-position steps
static bool krow_geometry(const TriConvDesc* d, int act_fmt, KrowArgs* g, int* co_t, int* tiles, int* steps) {
    if (!act_fmt || krow_wgrad_disabled()) return false;
    if (d->KD != 1 || d->ID != 1 || d->KH != 3 || d->KW != 3 || (d->stride != 1 && d->stride != 2) || d->pad_d != 0 || d->pad_h != 1 || d->pad_w != 1) return false;
    if (d->Cin % 64 || d->Cout % 64 || d->OH * d->stride != d->IH || d->OW * d->stride != d->IW) return false;
    const int W = d->OW, H = d->OH;
    if (W < 4 || W > 64) return false;
    const size_t in_bytes = (size_t)d->B * d->IH * d->IW * d->Cin * 2, dout_bytes = (size_t)d->B * H * W * d->Cout * 2;
    if (in_bytes >= ((size_t)1 << 31) || dout_bytes >= ((size_t)1 << 31)) return false;
    KrowArgs a{};
    a.in_bytes = (unsigned)in_bytes; a.dout_bytes = (unsigned)dout_bytes;
    a.NH = d->B * H; a.H = H; a.W = W; a.Cin = d->Cin; a.Cout = d->Cout; a.Kpad = 9 * d->Cin;
    a.Hin = d->IH; a.Win = d->IW;
    a.rps = 64 / W;
    a.P = d->stride == 1 ? ((W + 3) & ~1) : 2 * W + 2;               // even pitch: the bank half of a pixel is the parity of its index
    if (a.rps * a.P > (d->stride == 1 ? KROW_SLAB_PX : KROW_SLAB_PX_S2)) return false;
    a.npiece = (a.rps * a.P + 7) / 8;
    a.nsteps = (a.NH + a.rps - 1) / a.rps;
    a.ci_chunks = d->Cin / 64;
    *co_t = d->Cout % 128 == 0 ? 128 : 64;
    a.ntiles = (d->Cout / *co_t) * 3 * a.ci_chunks;
    a.dH = make_fastdiv(H); a.dP = make_fastdiv(a.P); a.dW = make_fastdiv(W);
    *g = a;
    *tiles = a.ntiles;
    *steps = a.nsteps;
    return true;
}
#define KROW_TARGET_BLOCKS 512                                      // two workgroups per CU (56 / 40 KB of LDS, <= 256 registers)
// split plan of a layer launched alone: one resident round of workgroups, at least 4 steps per split
static void krow_plan_alone(int tiles, int steps, int* sps, int* splits) {
    int s = KROW_TARGET_BLOCKS / tiles;
    const int max_by_steps = steps / 4 > 0 ? steps / 4 : 1;
    if (s > max_by_steps) s = max_by_steps;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    *sps = (steps + s - 1) / s;
    *splits = (steps + *sps - 1) / *sps;
}
static inline int krow_blocks(const KrowArgs& a) {               // (a multiple of 8: the next job's range starts on XCD 0 again)
    return a.nsplits >= 16 ? ((a.nsplits + 7) / 8) * 8 * a.ntiles : (a.nsplits * a.ntiles + 7) / 8 * 8;
}
// jobs of one tile height in ONE launch; block ranges are dealt longest split first so that the short jobs' workgroups fill in behind
template <int CO_T, typename E, int S>
static int launch_krow_jobs(const KrowArgs* a, int n, hipStream_t stream) {
    constexpr int STAGE = 64 * CO_T * 2 + (S == 1 ? KROW_SLAB_PX : KROW_SLAB_PX_S2) * 128;
    KrowJobs jobs{};
    int order[WGRAD_JOBS_MAX];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i)                                      // insertion sort by steps per split, descending (stable)
        for (int j = i; j > 0 && a[order[j]].steps_per_split > a[order[j - 1]].steps_per_split; --j) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        jobs.d[i] = a[order[i]];
        jobs.first_block[i] = blocks;
        blocks += krow_blocks(a[order[i]]);
    }
    jobs.first_block[n] = blocks;
    jobs.n = n;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_wgrad_krow_kernel<CO_T, E, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        attr_set = true;
    }
    conv_wgrad_krow_kernel<CO_T, E, S><<<dim3(blocks), 256, 2 * STAGE, stream>>>(jobs);
    return tri_check_launch("tri_conv_wgrad(krow)");
}
// (the slab size is a template parameter: 1 = all jobs stride 1, 2 = all stride 2, 0 = mixed - per job, big slab)
static int launch_krow(int co_t, int act_fmt, const KrowArgs* a, int n, hipStream_t s) {
    int n2 = 0;
    for (int i = 0; i < n; ++i) n2 += a[i].Hin != a[i].H;
    if (n2 && n2 != n) {
#define TRI_KR0 (act_fmt == TRI_FMT_F16 ? (co_t == 128 ? launch_krow_jobs<128, f16_t, 0>(a, n, s) : launch_krow_jobs<64, f16_t, 0>(a, n, s))   \
                                        : (co_t == 128 ? launch_krow_jobs<128, bf16_t, 0>(a, n, s) : launch_krow_jobs<64, bf16_t, 0>(a, n, s)))
        return TRI_KR0;
#undef TRI_KR0
    }
    const bool s2 = n2 > 0;
#define TRI_KR(S_)                                                                                                                      \
    (act_fmt == TRI_FMT_F16 ? (co_t == 128 ? launch_krow_jobs<128, f16_t, S_>(a, n, s) : launch_krow_jobs<64, f16_t, S_>(a, n, s))      \
                            : (co_t == 128 ? launch_krow_jobs<128, bf16_t, S_>(a, n, s) : launch_krow_jobs<64, bf16_t, S_>(a, n, s)))
    return s2 ? TRI_KR(2) : TRI_KR(1);
#undef TRI_KR
}

extern "C" size_t tri_conv_wgrad_workspace(const TriConvDesc* d) {
    size_t need = 0;
    {
        StemWgradArgs sg; int grid;
        if (stem_wgrad_geometry(d, 1, &sg, &grid)) need = (size_t)grid * 64 * d->KH * 32 * sizeof(float);
        C64WgradArgs cg;
        if (c64_wgrad_geometry(d, 1, &cg, &grid)) need = (size_t)grid * 64 * 576 * sizeof(float);
        TriVox0Geom vg;
        if (tri_internal_vox0_geometry(d->B, d->ID, d->IH, d->IW, d->Cin, d->OD, d->OH, d->OW, d->Cout, d->KD, d->KH, d->KW, d->stride, d->pad_d,
                                       d->pad_h, d->pad_w, &vg) && tri_internal_vox0_wgrad_grid(vg) > 0)
            need = (size_t)tri_internal_vox0_wgrad_grid(vg) * 32 * 144 * sizeof(float);
        KrowArgs kg; int co_t, tiles, steps;
        if (krow_geometry(d, 1, &kg, &co_t, &tiles, &steps)) {
            int sps, splits;
            krow_plan_alone(tiles, steps, &sps, &splits);
            const size_t n = (size_t)splits * d->Cout * kg.Kpad * sizeof(float);
            if (n > need) need = n;
        }
    }
    for (int mode = 0; mode < 4; ++mode) {                        // fp32 / 16-bit storage x position range / row list
        int BI, BJ, tiles, splits, sps, Kpad, dma;
        wgrad_plan(d, mode & 1, mode >> 1, &BI, &BJ, &tiles, &splits, &sps, &Kpad, &dma);
        size_t n = (size_t)splits * d->Cout * Kpad * sizeof(float);
        if (n > need) need = n;
    }
    return need;
}

// 0: conv_wgrad_kernel (register-staged), 2: conv_wgrad_dma_kernel (bf16 activation storage, LDS-DMA).  For profilers.
extern "C" int tri_conv_wgrad_kernel_family(const TriConvDesc* d, int act_fmt) {
    {
        TriVox0Geom vg;
        if (act_fmt && tri_internal_vox0_geometry(d->B, d->ID, d->IH, d->IW, d->Cin, d->OD, d->OH, d->OW, d->Cout, d->KD, d->KH, d->KW, d->stride,
                                                  d->pad_d, d->pad_h, d->pad_w, &vg) && tri_internal_vox0_wgrad_grid(vg) > 0)
            return 6;                                              // conv_vox0_wgrad_kernel when the call passes a site mask
    }
    {
        C64WgradArgs cg; KrowArgs kg; int grid, co_t, tiles, steps;
        if (!c64_wgrad_geometry(d, act_fmt, &cg, &grid) && krow_geometry(d, act_fmt, &kg, &co_t, &tiles, &steps)) return 7;   // conv_wgrad_krow_kernel
    }
    int BI, BJ, tiles, splits, sps, Kpad, dma;
    wgrad_plan(d, act_fmt, 0, &BI, &BJ, &tiles, &splits, &sps, &Kpad, &dma);
    return dma ? 2 : 0;
}

static inline int wgrad_dma_blocks(const WgradArgs& a) {              // (16+ splits: padded to whole groups of 8 for the XCD pinning)
    return a.nsplits >= 16 ? ((a.nsplits + 7) / 8) * 8 * a.ntiles : a.nsplits * a.ntiles;
}
static inline size_t wgrad_dma_plan_bytes(const WgradArgs& a) {       // gather plan (+ the row list's positions) in LDS
    const size_t entries = a.plan_ring ? (size_t)2 * WGRAD_RING_STEPS * 64 : (size_t)a.steps_per_split * 64;
    return entries * (a.row_count ? 12 : 8);
}
template <int BI, int BJ, typename E, int NWI = 2, int NWJ = 2>
static int launch_wgrad_dma_jobs(const WgradArgs* a, int n, hipStream_t stream) {
    constexpr int STAGE = 64 * (BI * 2 + BJ * 2);
    WgradJobs jobs{};
    size_t plan_bytes = 0;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        jobs.d[i] = a[i];
#ifdef WGRAD_STAMPS
        jobs.d[i].dbg = g_wgrad_dbg;
#endif
        jobs.first_block[i] = blocks;
        blocks += wgrad_dma_blocks(a[i]);
        if (wgrad_dma_plan_bytes(a[i]) > plan_bytes) plan_bytes = wgrad_dma_plan_bytes(a[i]);
    }
    jobs.first_block[n] = blocks;
    jobs.n = n;
    size_t smem = 2 * STAGE + 512 + plan_bytes;
#ifdef WGRAD_STAMPS
    smem += 2048;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<BI, BJ, E, NWI, NWJ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * STAGE + 512 + (NWI * NWJ > 4 ? 16 * 768 : 96 * 768) + 2048);
        attr_set = true;
    }
    conv_wgrad_dma_kernel<BI, BJ, E, NWI, NWJ><<<dim3(blocks), NWI * NWJ * 64, smem, stream>>>(jobs);
    return tri_check_launch("tri_conv_wgrad(dma)");
}
template <int BI, int BJ, typename E>
static int launch_wgrad_dma(const WgradArgs& a, int tiles, int splits, hipStream_t stream) {
    (void)tiles; (void)splits;
    return launch_wgrad_dma_jobs<BI, BJ, E>(&a, 1, stream);
}

#define WGRAD_MAX_STEPS 96
template <int BI, int BJ, int NSPLIT, typename AT>
static int launch_wgrad(const WgradArgs& a, int tiles, int splits, hipStream_t stream) {
    constexpr int STAGE = NSPLIT * (32 * BI * 2 + 32 * BJ * 2);
    size_t smem = 2 * STAGE + 512 + (size_t)a.steps_per_split * (a.row_count ? 384 : 256);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<BI, BJ, NSPLIT, AT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * STAGE + 512 + WGRAD_MAX_STEPS * 384);
        attr_set = true;
    }
    conv_wgrad_kernel<BI, BJ, NSPLIT, AT><<<dim3(((splits + 7) / 8) * 8 * tiles), 256, smem, stream>>>(a);
    return tri_check_launch("tri_conv_wgrad");
}

// the kernel arguments of one layer (conv_wgrad_kernel / conv_wgrad_dma_kernel) from its descriptor and split plan
static int wgrad_fill_args(const TriConvDesc* d, const void* in, const void* dout, const uint8_t* row_mask, const void* plan, float* slab,
                           const int* row_pos, const int* row_count, int split3, int act_fmt, int Kpad, int sps, int tiles, int splits,
                           WgradArgs* out) {
    WgradArgs a{};
    a.in = in; a.dout = dout; a.row_mask = row_mask; a.slab = slab; a.row_pos = row_pos; a.row_count = row_count;
    a.B = d->B; a.ID = d->ID; a.IH = d->IH; a.IW = d->IW; a.Cin = d->Cin;
    a.OD = d->OD; a.OH = d->OH; a.OW = d->OW; a.Cout = d->Cout;
    a.KD = d->KD; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pd = d->pad_d; a.ph = d->pad_h; a.pw = d->pad_w;
    a.ntaps = d->KD * d->KH * d->KW;
    if (a.ntaps > 64) { tri_set_error("wgrad: more than 64 taps unsupported"); return TRI_ERR_UNSUPPORTED; }
    a.Kpad = Kpad;
    a.M = d->B * d->OD * d->OH * d->OW;
    a.cin_shift = ilog2_exact(a.Cin);
    a.steps_per_split = sps;
    a.ntiles = tiles;
    a.nsplits = splits;
    size_t in_bytes = (size_t)a.B * a.ID * a.IH * a.IW * a.Cin * (act_fmt ? 2 : 4);
    if (act_fmt && split3) { tri_set_error("wgrad: 16-bit activation storage takes single operands (no 3-product split)"); return TRI_ERR_ARG; }
    if (!plan) { tri_set_error("wgrad: a gather plan from tri_conv_plan_build is required"); return TRI_ERR_ARG; }
    if (in_bytes >= ((size_t)1 << 31)) { tri_set_error("wgrad: input tensor >= 2 GiB (32-bit buffer offsets)"); return TRI_ERR_UNSUPPORTED; }
    {
        long Mpad = ((long)a.M + 31) / 32 * 32;
        a.plan_off = (const int*)plan;
        a.plan_mask = (const unsigned*)plan + Mpad;
        a.in_bytes = (unsigned)in_bytes;
    }
    a.dOW = make_fastdiv(a.OW); a.dOH = make_fastdiv(a.OH); a.dOD = make_fastdiv(a.OD); a.dCin = make_fastdiv(a.Cin);
    *out = a;
    return 0;
}
// reduce descriptor of a layer whose partial kernel wrote `splits` slabs [Cout][Kpad]
static void wgrad_fill_pending(const TriConvDesc* d, const float* slab, int splits, int Kpad, float* dw, long s_co, long s_tap, long s_ci,
                               int cin_real, float out_scale, TriWgradReduce* pending) {
    const int ntaps = d->KD * d->KH * d->KW;
    const long quads = (long)d->Cout * ((ntaps * d->Cin) >> 2);
    int zlanes = 1;
    while (zlanes < 64 && zlanes * 2 <= splits && quads * zlanes < 262144) zlanes *= 2;
    const int kq = 256 / zlanes;
    pending->slab = slab; pending->dw = dw; pending->s_co = s_co; pending->s_tap = s_tap; pending->s_ci = s_ci;
    pending->splits = splits; pending->Cout = d->Cout; pending->Kpad = Kpad; pending->ntaps = ntaps; pending->cin_stored = d->Cin;
    pending->cin_real = cin_real; pending->zlanes = zlanes; pending->nblocks = (int)((quads + kq - 1) / kq); pending->out_scale = out_scale;
    pending->kw_real = 0; pending->kw_shift = 0;
    // torchvision layout with several taps and not too many splits: the row form (wgrad_reduce_row), one output channel per block
    static int rows = -1;
    if (rows < 0) rows = 1;
    if (rows && ntaps > 1 && s_tap == 1 && s_ci == ntaps && s_co == (long)ntaps * d->Cin && cin_real == d->Cin && d->Cin % 4 == 0 &&
        ntaps * (d->Cin + 1) <= WGRAD_ROW_MAX && splits <= 48) {
        pending->zlanes = 0;
        pending->nblocks = d->Cout;
    }
}

// launch of conv_stem_wgrad_kernel (sg.y != NULL: the BNF instantiation) + the reduce descriptor of its per-workgroup slabs
static int stem_wgrad_launch(const TriConvDesc* d, StemWgradArgs& sg, int grid, void* workspace, size_t workspace_bytes, float* dw, long s_co,
                             long s_tap, long s_ci, int cin_real, int act_fmt, float out_scale, TriWgradReduce* pending, void* stream) {
    const size_t need = (size_t)grid * 64 * d->KH * 32 * sizeof(float);
    if (workspace_bytes < need) { tri_set_error("wgrad(stem): workspace too small"); return TRI_ERR_ARG; }
    sg.slab = (float*)workspace;
    sg.h_abl = tri_probe_ablation();
    { static int pipe = -1; if (pipe < 0) pipe = 1; sg.unpiped = !pipe; }
    const bool bnf = sg.y != nullptr;
    const size_t smem = (size_t)sg.TH * sg.OW * 128 + (size_t)sg.slab_rows * sg.row_bytes + (bnf ? 5 * 64 * sizeof(float) : 0);
    hipStream_t st = (hipStream_t)stream;
#define TRI_SWG1(KH_, T_, F_)                                                                                                  \
    do {                                                                                                                       \
        static bool attr = false;                                                                                              \
        if (!attr) {                                                                                                           \
            hipFuncSetAttribute((const void*)conv_stem_wgrad_kernel<KH_, T_, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
            attr = true;                                                                                                       \
        }                                                                                                                      \
        conv_stem_wgrad_kernel<KH_, T_, F_><<<grid, 256, smem, st>>>(sg);                                                      \
    } while (0)
#define TRI_SWG(KH_)                                                                                                           \
    case KH_:                                                                                                                  \
        if (act_fmt == TRI_FMT_F16) { if (bnf) TRI_SWG1(KH_, f16_t, true); else TRI_SWG1(KH_, f16_t, false); }                 \
        else { if (bnf) TRI_SWG1(KH_, bf16_t, true); else TRI_SWG1(KH_, bf16_t, false); }                                      \
        break;
    switch (d->KH) { TRI_SWG(7) TRI_SWG(5) TRI_SWG(3) default: break; }
#undef TRI_SWG
#undef TRI_SWG1
    int rc = tri_check_launch("tri_conv_wgrad(stem)");
    if (rc) return rc;
    const int ntaps_p = d->KH * 8;
    const long quads = (long)64 * ntaps_p;
    int zlanes = 1;
    while (zlanes < 64 && zlanes * 2 <= grid && quads * zlanes < 262144) zlanes *= 2;
    const int kq = 256 / zlanes;
    pending->slab = (const float*)workspace; pending->dw = dw; pending->s_co = s_co; pending->s_tap = s_tap; pending->s_ci = s_ci;
    pending->splits = grid; pending->Cout = 64; pending->Kpad = d->KH * 32; pending->ntaps = ntaps_p; pending->cin_stored = 4;
    pending->cin_real = cin_real; pending->zlanes = zlanes; pending->nblocks = (int)((quads + kq - 1) / kq);
    pending->out_scale = out_scale; pending->kw_real = d->KW; pending->kw_shift = 3;
    return 0;
}
// Weight gradient of the stem conv straight from the BatchNorm-backward inputs (conv -> BN -> ReLU -> MaxPool2d(3, 2, 1), mv_cnn.py:44):
// the gradient w.r.t. the conv output that tri_maxpool_bn_bwd_apply would store (and this kernel re-read) is formed while the dOut tile
// is staged.  The stem has no data gradient, so that tensor (100 MB at the bench shape) is never materialised.  y [N,OH,OW,64] conv
// output, arg / dpool [N,OH/2,OW/2,64] winning-tap map and pooled gradient, c1..c3 from tri_bn_bwd_finalize, relu_scale / relu_shift
// the forward's BN coefficients.  TRI_ERR_UNSUPPORTED when the layer does not take conv_stem_wgrad_kernel (caller: apply + wgrad).
extern "C" int tri_conv_stem_wgrad_bn(const TriConvDesc* d, const void* in, const void* y, const uint8_t* arg, const void* dpool, const float* c1,
                                      const float* c2, const float* c3, const float* relu_scale, const float* relu_shift, void* workspace,
                                      size_t workspace_bytes, float* dw, long s_co, long s_tap, long s_ci, int cin_real, int act_fmt,
                                      float out_scale, TriWgradReduce* pending, void* stream) {
    if (!pending) { tri_set_error("wgrad: pending descriptor is NULL"); return TRI_ERR_ARG; }
    StemWgradArgs sg; int grid;
    if (!stem_wgrad_geometry(d, act_fmt, &sg, &grid) || d->OH % 2 || d->OW % 2 || sg.TH != 2 || (sg.slab_rows * sg.row_bytes) % 16) {
        tri_set_error("tri_conv_stem_wgrad_bn: not a stem-kernel layer with an even output grid"); return TRI_ERR_UNSUPPORTED;
    }
    sg.in = in; sg.dout = nullptr;
    sg.y = y; sg.arg = arg; sg.dpool = dpool; sg.c1 = c1; sg.c2 = c2; sg.c3 = c3; sg.rs = relu_scale; sg.rb = relu_shift;
    return stem_wgrad_launch(d, sg, grid, workspace, workspace_bytes, dw, s_co, s_tap, s_ci, cin_real, act_fmt, out_scale, pending, stream);
}

// dw (addressed by element strides s_co / s_tap / s_ci, i.e. directly in the reference's parameter layout)
//   = sum over positions of dout x im2col(in).  row_mask (optional, per output position, buffer padded to a
// multiple of 32 bytes) marks live positions; split3 != 0 selects the 3-product bf16 split mode.
extern "C" int tri_conv_wgrad_partial(const TriConvDesc* d, const void* in, const void* dout, const uint8_t* row_mask, const void* plan,
                                      void* workspace, size_t workspace_bytes, float* dw, long s_co, long s_tap, long s_ci, int cin_real,
                                      int split3, int act_fmt, float out_scale, const int* row_pos, const int* row_count,
                                      TriWgradReduce* pending, void* stream) {
    if (!pending) { tri_set_error("wgrad: pending descriptor is NULL"); return TRI_ERR_ARG; }
    if ((row_pos == nullptr) != (row_count == nullptr)) { tri_set_error("wgrad: row_pos and row_count go together"); return TRI_ERR_ARG; }
    pending->kw_real = 0;
    {
        C64WgradArgs cg; int grid;
        if (!row_mask && !row_count && !split3 && c64_wgrad_geometry(d, act_fmt, &cg, &grid)) {
            const size_t need = (size_t)grid * 64 * 576 * sizeof(float);
            if (workspace_bytes < need) { tri_set_error("wgrad(c64): workspace too small"); return TRI_ERR_ARG; }
            cg.in = in; cg.dout = dout; cg.slab = (float*)workspace;
            cg.h_abl = tri_probe_ablation();
            const size_t smem = (size_t)cg.TR * cg.W * 128 + (size_t)(cg.TR + 2) * (cg.W + 2) * 128;
            static bool attr = false;
            if (!attr) {
                hipFuncSetAttribute((const void*)conv_wgrad_c64_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                hipFuncSetAttribute((const void*)conv_wgrad_c64_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                attr = true;
            }
            if (act_fmt == TRI_FMT_F16) conv_wgrad_c64_kernel<f16_t><<<grid, 256, smem, (hipStream_t)stream>>>(cg);
            else conv_wgrad_c64_kernel<bf16_t><<<grid, 256, smem, (hipStream_t)stream>>>(cg);
            int rc = tri_check_launch("tri_conv_wgrad(c64)");
            if (rc) return rc;
            const long quads = (long)64 * 144;
            int zlanes = 1;
            while (zlanes < 64 && zlanes * 2 <= grid && quads * zlanes < 262144) zlanes *= 2;
            const int kq = 256 / zlanes;
            pending->slab = (const float*)workspace; pending->dw = dw; pending->s_co = s_co; pending->s_tap = s_tap; pending->s_ci = s_ci;
            pending->splits = grid; pending->Cout = 64; pending->Kpad = 576; pending->ntaps = 9; pending->cin_stored = 64;
            pending->cin_real = cin_real; pending->zlanes = zlanes; pending->nblocks = (int)((quads + kq - 1) / kq);
            pending->out_scale = out_scale; pending->kw_shift = 0;
            return 0;
        }
    }
    {   // voxel level 0 over a site mask: conv_vox0_wgrad_kernel (conv_vox.hip)
        TriVox0Geom vg; int grid;
        if (row_mask && !row_count && !split3 && act_fmt &&
            tri_internal_vox0_geometry(d->B, d->ID, d->IH, d->IW, d->Cin, d->OD, d->OH, d->OW, d->Cout, d->KD, d->KH, d->KW, d->stride, d->pad_d,
                                       d->pad_h, d->pad_w, &vg) && (grid = tri_internal_vox0_wgrad_grid(vg)) > 0) {
            const size_t need = (size_t)grid * 32 * 144 * sizeof(float);
            if (workspace_bytes < need) { tri_set_error("wgrad(vox0): workspace too small"); return TRI_ERR_ARG; }
            int rc = tri_internal_vox0_wgrad_launch(vg, grid, d->B, in, dout, row_mask, (float*)workspace, act_fmt, (hipStream_t)stream);
            if (rc) return rc;
            const long quads = (long)32 * 36;
            int zlanes = 1;
            while (zlanes < 64 && zlanes * 2 <= grid && quads * zlanes < 262144) zlanes *= 2;
            const int kq = 256 / zlanes;
            pending->slab = (const float*)workspace; pending->dw = dw; pending->s_co = s_co; pending->s_tap = s_tap; pending->s_ci = s_ci;
            pending->splits = grid; pending->Cout = 32; pending->Kpad = 144; pending->ntaps = 36; pending->cin_stored = 4;
            pending->cin_real = cin_real; pending->zlanes = zlanes; pending->nblocks = (int)((quads + kq - 1) / kq);
            pending->out_scale = out_scale; pending->kw_real = 3; pending->kw_shift = 2;
            return 0;
        }
    }
    {
        StemWgradArgs sg; int grid;
        if (!row_mask && !row_count && !split3 && stem_wgrad_geometry(d, act_fmt, &sg, &grid)) {
            sg.in = in; sg.dout = dout;
            sg.y = nullptr;
            return stem_wgrad_launch(d, sg, grid, workspace, workspace_bytes, dw, s_co, s_tap, s_ci, cin_real, act_fmt, out_scale, pending, stream);
        }
    }
    if (d->Cin % 4 != 0 || d->Cout % 4 != 0) { tri_set_error("wgrad: channels must be multiples of 4"); return TRI_ERR_ARG; }
    {   // resolution-keeping 3x3 layers, 16-bit storage: conv_wgrad_krow_kernel
        KrowArgs kg; int co_t, tiles, steps;
        if (!row_mask && !row_count && !split3 && krow_geometry(d, act_fmt, &kg, &co_t, &tiles, &steps)) {
            int sps, splits;
            krow_plan_alone(tiles, steps, &sps, &splits);
            if (workspace_bytes < (size_t)splits * d->Cout * kg.Kpad * sizeof(float)) { tri_set_error("wgrad(krow): workspace too small"); return TRI_ERR_ARG; }
            kg.in = in; kg.dout = dout; kg.slab = (float*)workspace; kg.steps_per_split = sps; kg.nsplits = splits;
            int rc = launch_krow(co_t, act_fmt, &kg, 1, (hipStream_t)stream);
            if (rc) return rc;
            wgrad_fill_pending(d, (const float*)workspace, splits, kg.Kpad, dw, s_co, s_tap, s_ci, cin_real, out_scale, pending);
            return 0;
        }
    }
    int BI, BJ, tiles, splits, sps, Kpad, dma;
    wgrad_plan(d, act_fmt, row_count != nullptr, &BI, &BJ, &tiles, &splits, &sps, &Kpad, &dma);
    if (workspace_bytes < (size_t)splits * d->Cout * Kpad * sizeof(float)) { tri_set_error("wgrad: workspace too small"); return TRI_ERR_ARG; }
    WgradArgs a{};
    {
        int rc = wgrad_fill_args(d, in, dout, row_mask, plan, (float*)workspace, row_pos, row_count, split3, act_fmt, Kpad, sps, tiles, splits, &a);
        if (rc) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define TRI_WG(BI_, BJ_)                                                                                   \
    (act_fmt == TRI_FMT_F16 ? launch_wgrad<BI_, BJ_, 1, f16_t>(a, tiles, splits, s)                      \
     : act_fmt == TRI_FMT_BF16 ? launch_wgrad<BI_, BJ_, 1, bf16_t>(a, tiles, splits, s)                  \
              : (split3 ? launch_wgrad<BI_, BJ_, 2, float>(a, tiles, splits, s) : launch_wgrad<BI_, BJ_, 1, float>(a, tiles, splits, s)))
    if (dma) {
        if ((size_t)a.M * a.Cout * 2 >= ((size_t)1 << 31)) { tri_set_error("wgrad: dOut tensor >= 2 GiB (32-bit buffer offsets)"); return TRI_ERR_UNSUPPORTED; }
        if (act_fmt == TRI_FMT_F16) rc = BI == 128 ? launch_wgrad_dma<128, 128, f16_t>(a, tiles, splits, s) : launch_wgrad_dma<64, 128, f16_t>(a, tiles, splits, s);
        else rc = BI == 128 ? launch_wgrad_dma<128, 128, bf16_t>(a, tiles, splits, s) : launch_wgrad_dma<64, 128, bf16_t>(a, tiles, splits, s);
    } else if (BI == 128) rc = TRI_WG(128, 128);
    else if (BJ == 256) rc = TRI_WG(64, 256);
    else rc = TRI_WG(64, 128);
#undef TRI_WG
    if (rc) return rc;
    wgrad_fill_pending(d, (const float*)workspace, splits, Kpad, dw, s_co, s_tap, s_ci, cin_real, out_scale, pending);
    return 0;
}

// TRICOLO_WGRAD_WIDE=1: grouped jobs with Cout % 256 == 0 take 256 x 128 tiles (512-thread workgroups, one per CU).  Measured at the
// bench shape: the launches themselves 0.286 -> 0.266 ms per step, the STEP 3.08 -> 3.12 ms (a 104 KB workgroup per CU leaves the
// other towers' kernels no room beside it) - so it stays an experiment switch (profiles/r3/NOTES_wgrad.md).
static bool wgrad_wide_tiles() {
    static int v = -1;
    if (v < 0) v = 0;
    return v == 1;
}
static int wgrad_group_target(int family) {                     // resident workgroups a grouped launch is planned for
    // 128x128 tiles (64 KB of stages + the plan ring: two workgroups per CU): all 512 slots; 64x128 tiles: 448 as for single launches
    // (measured on the bench shape, profiles/r3/NOTES_wgrad.md).  TRICOLO_WGRAD_GROUP_BLOCKS="a,b" overrides (a: 128-row, b: 64-row tiles)
    // 256x128 tiles: one 512-thread workgroup per CU.
    static int v[7] = {-1, -1, -1, -1, -1, -1, -1};
    if (v[0] < 0) {
        // 4 / 6: conv_wgrad_krow_kernel<128> (stride 1 / 2), two workgroups per CU; 5: <64> stride 1, three per CU (768 slots: 40 KB of LDS,
        // 126 registers - layer1's four layers in one launch 592 -> 725 TF; the step does not move); 7: <64> stride 2 (one layer of the
        // trunk: it rides in the stride-1 launch, so its own target is the two-per-CU default)
        v[0] = 512; v[1] = 448; v[2] = 256; v[3] = KROW_TARGET_BLOCKS; v[4] = 768; v[5] = KROW_TARGET_BLOCKS; v[6] = KROW_TARGET_BLOCKS;
    }
    return v[family >= 1 && family <= 7 ? family - 1 : 1];
}
// ---- several layers in one launch
// family of a layer for grouping: 0 = not groupable (tri_conv_wgrad_partial), 1 = conv_wgrad_dma_kernel<128,128>, 2 = <64,128>, 3 = <256,128>
// (TRICOLO_WGRAD_WIDE), 4 = conv_wgrad_krow_kernel<128> (stride 1 and 2 share launches), 5 = conv_wgrad_krow_kernel<64>, 7 = its stride-2 form;
// tiles = output tiles (workgroups per split), steps = 64-position steps of the contraction
extern "C" int tri_conv_wgrad_group_info(const TriConvDesc* d, int act_fmt, int* family, int* tiles, int* steps) {
    *family = 0; *tiles = 0; *steps = 0;
    if (!act_fmt || d->Cin % 4 != 0 || d->Cout % 4 != 0) return 0;
    {
        C64WgradArgs cg; StemWgradArgs sg; int grid;
        if (c64_wgrad_geometry(d, act_fmt, &cg, &grid) || stem_wgrad_geometry(d, act_fmt, &sg, &grid)) return 0;
        KrowArgs kg; int co_t;
        if (krow_geometry(d, act_fmt, &kg, &co_t, tiles, steps)) { *family = co_t == 128 ? 4 : (d->stride == 2 ? 7 : 5); return 0; }
    }
    int BI, BJ, t, splits, sps, Kpad, dma;
    wgrad_plan(d, act_fmt, 0, &BI, &BJ, &t, &splits, &sps, &Kpad, &dma);
    if (!dma) return 0;
    *family = BI == 128 ? 1 : 2;
    if (BI == 128 && d->Cout % 256 == 0 && wgrad_wide_tiles()) {      // 256 x 128 tiles, 512-thread workgroups
        *family = 3;
        t = (d->Cout / 256) * ((Kpad + 127) / 128);
    }
    *tiles = t;
    *steps = (int)(((long)d->B * d->OD * d->OH * d->OW + 63) / 64);
    return 0;
}
// n <= TRI_WGRAD_JOBS_MAX layers of ONE family (tri_conv_wgrad_group_info), each over its dense position range or over a compact row list
// (row_pos / row_count; no row MASK), 16-bit activation storage.  Every job gets the splits that make all workgroups of the launch about
// equally long and never more than it would get alone (so tri_conv_wgrad_workspace still bounds its slab); pending[i] receives job i's
// reduce descriptor.  Row-list jobs are PLANNED for the dense position count (the list length lives on the device): a list that is
// 15 % full leaves its splits correspondingly short - the kernel re-derives the steps per split from *row_count, so the result does not
// depend on the plan, only the balance of the launch does (the contract is "correct for any occupancy, sized for a full list").
extern "C" int tri_conv_wgrad_partial_group(const TriWgradJob* jobs, int n, int act_fmt, TriWgradReduce* pending, void* stream) {
    if (n < 1 || n > WGRAD_JOBS_MAX || !jobs || !pending) { tri_set_error("wgrad group: 1..TRI_WGRAD_JOBS_MAX jobs"); return TRI_ERR_ARG; }
    static_assert(WGRAD_JOBS_MAX == TRI_WGRAD_JOBS_MAX, "header and kernel disagree");
    {
        int fam, t0, st0;
        tri_conv_wgrad_group_info(jobs[0].d, act_fmt, &fam, &t0, &st0);
        if (fam >= 4 && fam <= 7) {
            // kernel-row jobs: every job is cut into splits of about the launch's mean workgroup length (total tile-steps / resident
            // slots), never more splits than it would get alone; the launch may hold more workgroups than slots - they are dealt longest
            // first and the short ones fill in behind (one resident round would leave the few-step layers' slots idle for most of it)
            KrowArgs ka[WGRAD_JOBS_MAX];
            int co_t0 = 0, tl[WGRAD_JOBS_MAX], st[WGRAD_JOBS_MAX];
            long total = 0;
            for (int i = 0; i < n; ++i) {
                int co_t;
                if (jobs[i].row_pos || jobs[i].row_count || !krow_geometry(jobs[i].d, act_fmt, &ka[i], &co_t, &tl[i], &st[i]) || (i && co_t != co_t0)) {
                    tri_set_error("wgrad group: jobs must share one groupable kernel family"); return TRI_ERR_ARG;
                }
                co_t0 = co_t;
                total += (long)tl[i] * st[i];
            }
            int common = (int)((total + wgrad_group_target(fam) - 1) / wgrad_group_target(fam));
            if (common < 8) common = 8;
            for (int i = 0; i < n; ++i) {
                int sps_alone, splits_alone;
                krow_plan_alone(tl[i], st[i], &sps_alone, &splits_alone);
                int sps = n == 1 ? sps_alone : (common > sps_alone ? common : sps_alone);
                if (sps > st[i]) sps = st[i];
                const int splits = (st[i] + sps - 1) / sps;
                sps = (st[i] + splits - 1) / splits;                 // evened out
                const TriConvDesc* d = jobs[i].d;
                if (jobs[i].workspace_bytes < (size_t)splits * d->Cout * ka[i].Kpad * sizeof(float)) { tri_set_error("wgrad group(krow): workspace too small"); return TRI_ERR_ARG; }
                ka[i].in = jobs[i].in; ka[i].dout = jobs[i].dout; ka[i].slab = (float*)jobs[i].workspace;
                ka[i].steps_per_split = sps; ka[i].nsplits = splits;
                wgrad_fill_pending(d, (const float*)jobs[i].workspace, splits, ka[i].Kpad, jobs[i].dw, jobs[i].s_co, jobs[i].s_tap, jobs[i].s_ci,
                                   jobs[i].cin_real, jobs[i].out_scale, &pending[i]);
            }
            return launch_krow(co_t0, act_fmt, ka, n, (hipStream_t)stream);
        }
    }
    WgradArgs a[WGRAD_JOBS_MAX];
    int fam0 = 0, ind_sps[WGRAD_JOBS_MAX], tiles[WGRAD_JOBS_MAX], steps[WGRAD_JOBS_MAX], Kpads[WGRAD_JOBS_MAX];
    long total = 0;
    for (int i = 0; i < n; ++i) {
        int fam;
        tri_conv_wgrad_group_info(jobs[i].d, act_fmt, &fam, &tiles[i], &steps[i]);
        if (!fam || (i && fam != fam0)) { tri_set_error("wgrad group: jobs must share one groupable kernel family"); return TRI_ERR_ARG; }
        fam0 = fam;
        int BI, BJ, t, splits, dma;
        wgrad_plan(jobs[i].d, act_fmt, 0, &BI, &BJ, &t, &splits, &ind_sps[i], &Kpads[i], &dma);
        total += (long)tiles[i] * steps[i];
    }
    // the smallest common split length whose workgroups (tiles x splits over the jobs) all fit the launch's resident slots at once;
    // a job's own split length is then evened out (ceil(steps / splits)) and never shorter than the one it would get alone
    const int target = wgrad_group_target(fam0);
    int common = (int)((total + target - 1) / target);
    if (common < 1) common = 1;
    for (;; ++common) {
        long wgs = 0;
        for (int i = 0; i < n; ++i) {
            const int sps = common > ind_sps[i] ? common : ind_sps[i];
            wgs += (long)tiles[i] * ((steps[i] + sps - 1) / sps);
        }
        if (wgs <= target || common >= 4096) break;
    }
    for (int i = 0; i < n; ++i) {
        int sps = (n == 1 && fam0 != 3) ? ind_sps[i] : (common > ind_sps[i] ? common : ind_sps[i]);
        if (sps > steps[i]) sps = steps[i];
        const int splits = (steps[i] + sps - 1) / sps;
        if (n > 1 || fam0 == 3) sps = (steps[i] + splits - 1) / splits;
        const TriConvDesc* d = jobs[i].d;
        if (jobs[i].workspace_bytes < (size_t)splits * d->Cout * Kpads[i] * sizeof(float)) { tri_set_error("wgrad group: workspace too small"); return TRI_ERR_ARG; }
        if ((size_t)d->B * d->OD * d->OH * d->OW * d->Cout * 2 >= ((size_t)1 << 31)) { tri_set_error("wgrad: dOut tensor >= 2 GiB (32-bit buffer offsets)"); return TRI_ERR_UNSUPPORTED; }
        if ((jobs[i].row_pos == nullptr) != (jobs[i].row_count == nullptr)) { tri_set_error("wgrad group: row_pos and row_count go together"); return TRI_ERR_ARG; }
        int rc = wgrad_fill_args(d, jobs[i].in, jobs[i].dout, nullptr, jobs[i].plan, (float*)jobs[i].workspace, jobs[i].row_pos, jobs[i].row_count,
                                 0, act_fmt, Kpads[i], sps, tiles[i], splits, &a[i]);
        if (rc) return rc;
        a[i].plan_ring = sps > 2 * WGRAD_RING_STEPS ? 1 : 0;
        wgrad_fill_pending(d, (const float*)jobs[i].workspace, splits, Kpads[i], jobs[i].dw, jobs[i].s_co, jobs[i].s_tap, jobs[i].s_ci,
                           jobs[i].cin_real, jobs[i].out_scale, &pending[i]);
    }
    hipStream_t s = (hipStream_t)stream;
    if (fam0 == 3) return act_fmt == TRI_FMT_F16 ? launch_wgrad_dma_jobs<256, 128, f16_t, 4, 2>(a, n, s) : launch_wgrad_dma_jobs<256, 128, bf16_t, 4, 2>(a, n, s);
    if (act_fmt == TRI_FMT_F16) return fam0 == 1 ? launch_wgrad_dma_jobs<128, 128, f16_t>(a, n, s) : launch_wgrad_dma_jobs<64, 128, f16_t>(a, n, s);
    return fam0 == 1 ? launch_wgrad_dma_jobs<128, 128, bf16_t>(a, n, s) : launch_wgrad_dma_jobs<64, 128, bf16_t>(a, n, s);
}

extern "C" int tri_wgrad_reduce_grouped(const TriWgradReduce* pending, int n, void* stream) {
    return tri_wgrad_reduce_grouped_noted(pending, n, nullptr, stream);
}
extern "C" int tri_wgrad_reduce_grouped_noted(const TriWgradReduce* pending, int n, int* note, void* stream) {
    if (n < 0 || (n > 0 && !pending)) { tri_set_error("wgrad reduce: bad descriptor list"); return TRI_ERR_ARG; }
    for (int base = 0; base < n; base += TRI_WGRAD_GROUP_MAX) {
        const int m = n - base < TRI_WGRAD_GROUP_MAX ? n - base : TRI_WGRAD_GROUP_MAX;
        WgradGroup g{};
        int blocks = 0;
        for (int i = 0; i < m; ++i) {
            g.d[i] = pending[base + i];
            if (g.d[i].nblocks <= 0 || g.d[i].zlanes < 0 || (g.d[i].zlanes > 0 && 256 % g.d[i].zlanes)) { tri_set_error("wgrad reduce: descriptor not filled by tri_conv_wgrad_partial"); return TRI_ERR_ARG; }
            g.first_block[i] = blocks;
            blocks += g.d[i].nblocks;
        }
        g.first_block[m] = blocks;
        g.note = note;
        wgrad_reduce_grouped_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(g, m);
        int rc = tri_check_launch("tri_wgrad_reduce_grouped");
        if (rc) return rc;
    }
    return 0;
}

extern "C" int tri_conv_wgrad(const TriConvDesc* d, const void* in, const void* dout, const uint8_t* row_mask, const void* plan,
                              void* workspace, size_t workspace_bytes, float* dw, long s_co, long s_tap, long s_ci, int cin_real,
                              int split3, int act_fmt, float out_scale, const int* row_pos, const int* row_count, void* stream) {
    TriWgradReduce r;
    int rc = tri_conv_wgrad_partial(d, in, dout, row_mask, plan, workspace, workspace_bytes, dw, s_co, s_tap, s_ci, cin_real, split3, act_fmt,
                                    out_scale, row_pos, row_count, &r, stream);
    if (rc) return rc;
    if (r.kw_real) return tri_wgrad_reduce_grouped(&r, 1, stream);        // (the padded-row mapping lives in the grouped kernel's argument table)
    wgrad_reduce_kernel<<<(unsigned)r.nblocks, 256, 0, (hipStream_t)stream>>>(r.slab, r.splits, r.Cout, r.Kpad, r.ntaps, r.cin_stored, r.cin_real,
                                                                             r.dw, r.s_co, r.s_tap, r.s_ci, r.zlanes, r.out_scale);
    return tri_check_launch("tri_wgrad_reduce");
}
