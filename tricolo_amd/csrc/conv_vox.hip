// Brick kernels of the voxel tower's SubMConv3d layers (sparse_cnn.py:12-32) for the 16-bit storage modes on gfx950.
//
// conv_igemm.hip gathers im2col rows tap by tap: every input site crosses the L2 -> LDS path 27 times, every k-step pays a
// dependent gather latency and every 128-row tile re-streams the filter bank.  The 3D analogue of conv_stem_kernel /
// conv_halo2d_kernel does not: a workgroup owns a BRICK of the dense grid, stages the brick's sites plus a one-site halo ONCE in
// an LDS slab and forms all 27 taps by shifted slab reads; the submanifold rule is applied per RUN of 16 x-consecutive sites:
// runs without an active site are skipped (wave-uniform), rows of inactive sites inside an active run are neither stored nor
// counted in the BatchNorm sums.
//
// conv_vox0_kernel: level 0 (4 stored input channels = 8 B per site, 32 output channels).
//   * brick = 2 z-planes x TY rows x all V columns = 64 or 128 runs; one brick per workgroup, workgroups of empty bricks leave
//     after reading <= 2 KB of site mask; the dispatcher balances the rest (4-5 workgroups per CU overlap each other's slab fill,
//     MFMAs and stores);
//   * with four stored channels one MFMA k-step (32) is 8 taps.  K-step (kd, p) holds the taps (kd, kh, kw) with kh = the lane's
//     k-group fq (fq = 3: zero weights) and kw = 2 p, 2 p + 1 (kw = 3: zero weights): six k-steps, and the slab address of a
//     lane's operand is  (run base: one VALU add per run) + (lane part fq * row pitch + site * 8) + an IMMEDIATE kd * plane pitch
//     + kw * 8 - the fragment reads cost no address arithmetic (the first version ordered the taps by a table to fit four k-steps
//     and spent ~100 instructions per run, 650 issue cycles against 128 of MFMA: in-kernel stamps, profiles/r3/NOTES_vox.md);
//   * row pitch = 128 mod 256 bytes, so the two 16-lane groups that one LDS cycle of a ds_read_b64 serves (k-groups fq, fq + 1 =
//     adjacent slab rows, 16 consecutive sites = 128 contiguous bytes each) hit disjoint banks: conflict-free without a swizzle;
//   * the whole filter bank (32 x 6 k-steps) is 48 registers of MFMA A fragments, loaded once per workgroup under the mask test,
//     with the output channels permuted so that a lane ends up with 8 CONSECUTIVE channels of one site: one 16-byte store per
//     lane, full 64-byte rows per 4 lanes;
//   * BatchNorm sums (of the values as stored) stay in registers over the workgroup's runs: one record per workgroup.
#include "common.h"
#include <stdlib.h>
#include "conv_vox.h"

int tri_internal_num_cus();                                                   // conv_igemm.hip

template <int N>
__device__ __forceinline__ float vox_row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}

struct Vox0Args {
    const void* in;            // [B, V, V, V, 4] 16-bit, zeros at inactive sites
    const void* w;             // packed operand rows [32][Kpad] (k = tap * 4 + channel)
    void* out;                 // [B, V, V, V, 32]; rows of inactive sites are not written
    const uint8_t* mask;       // [B * V^3] site mask, or NULL (every site active)
    float* stats;              // [grid][2][32] or NULL
    int B, Kpad;
    unsigned in_bytes;
#ifdef VOX_PROBE
    long long* dbg;            // [grid][8] stamps of wave 0
    int abl;                   // timing probes (wrong results): 1 no MFMA, 2 no stores, 4 no slab loads, 8 no fragment reads, 16 prologue only
#endif
};
#ifdef VOX_PROBE
#define VOX_ABL(bit) (p.abl & (bit))
#define VOX_STAMP(i) do { if (p.dbg && t == 0) p.dbg[(size_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define VOX_STAMPV(i, v) do { if (p.dbg && t == 0) p.dbg[(size_t)blockIdx.x * 8 + (i)] = (long long)(v); } while (0)
#else
#define VOX_ABL(bit) 0
#define VOX_STAMP(i)
#define VOX_STAMPV(i, v)
#endif

template <int V, int TY>
struct Vox0Cfg {
    static constexpr int PITCH = V * 8 + 128;                                  // [16 B pad | V sites x 8 B | 112 B pad]: 128 mod 256
    static constexpr int PLANE0 = (TY + 2) * PITCH;
    static constexpr int PLANE = PLANE0 + ((((PLANE0 >> 7) & 1) == 0) ? 128 : 0);   // 128 mod 256 as well
    static constexpr int SLAB = 4 * PLANE;
    static constexpr int RPR = V / 16;                                         // runs per grid row
    static constexpr int HALF = TY * RPR;                                      // runs per z-plane of the brick
    static constexpr int RUNS = 2 * HALF;                                      // 64 or 128
    static constexpr int NYB = V / TY;
    static constexpr int CPR = V / 2;                                          // 16-byte chunks per row
    static constexpr int ITEMS = (TY + 2) * CPR;                               // chunks per slab plane
    static constexpr int MAXC = (4 * ITEMS + 255) / 256;
    static constexpr int WROW = 264;                                           // filter-bank row pitch in LDS: 256 B + 8 (rows 2 banks apart)
    static constexpr size_t SMEM = (size_t)SLAB + RUNS * 16 + 4 * 32 * 2 * sizeof(float) + 32 * WROW;
    static_assert(RUNS == 64 || RUNS == 128, "a brick is one or two 64-run mask words");
    static_assert(2 * PLANE + 24 < 65536, "fragment-read immediates");
};

template <typename AT, int V, int TY>
__global__ __launch_bounds__(256, 4) void conv_vox0_kernel(const Vox0Args p) {
    typedef Vox0Cfg<V, TY> C;
    typedef typename OpOf<AT>::E E;
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
    // brick of this workgroup.  Workgroups i, i + 256, ... land on the same CU when the whole grid is resident at once (32^3 at the
    // bench batch: 1,024 workgroups, four per CU); in sample-major order those would be the SAME spatial brick of four samples - the
    // dense centre bricks on some CUs, the empty corner bricks on others (stamps: workgroup lifetimes 6.5 us median, 14 us for the
    // densest CU).  Rotating the spatial index by the sample number deals every CU a mix.
    constexpr int PS = (V / 2) * C::NYB;                                       // bricks per sample
    const int b = blockIdx.x / PS;
    const int sp = (blockIdx.x % PS + b * (PS / 2 + 1)) % PS;
    const int yb = sp % C::NYB, zp = sp / C::NYB;
    const int z0 = zp * 2, y0 = yb * TY;
    char* const slab = smem;
    uint8_t* const lmask = (uint8_t*)(smem + C::SLAB);                         // [RUNS][16 sites]
    float* const red = (float*)(lmask + C::RUNS * 16);                         // [4 waves][32][2]
    char* const wbuf = (char*)(red + 256);                                     // [32 rows][WROW] packed filter bank
    VOX_STAMP(0);
    VOX_STAMPV(5, __builtin_amdgcn_s_memrealtime());

    // ---- site mask of the brick: run r = (plane r / HALF, row (r % HALF) / RPR, x-run r % RPR) is 16 contiguous bytes
    uint4 mv = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
    if (t < C::RUNS && p.mask) mv = *(const uint4*)(p.mask + ((size_t)(b * V + z0 + t / C::HALF) * V + y0) * V + (t % C::HALF) * 16);

    // ---- filter bank: 8 KB, two coalesced 16-byte loads per thread into LDS, in flight under the mask test (the first versions read the
    // MFMA fragments straight from global memory: 24 eight-byte loads per wave scattered over 16 rows - ~400 L1 requests per wave, and
    // with every workgroup of the launch doing it at once the texture path was busy for microseconds: 3.9 us until the mask test)
    uint4 wld[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) wld[u] = *(const uint4*)((const char*)p.w + (size_t)(t + u * 256) * 16);
    if (t < C::RUNS) *(uint4*)(lmask + t * 16) = mv;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = t + u * 256;                                             // 16-byte chunk c of the [32][128] bank: row c / 16
        char* d = wbuf + (c >> 4) * C::WROW + (c & 15) * 16;
        *(uint2*)d = make_uint2(wld[u].x, wld[u].y);
        *(uint2*)(d + 8) = make_uint2(wld[u].z, wld[u].w);
    }
    __syncthreads();
    unsigned long long m0, m1 = 0ull;
    {
        const uint4 ma = *(const uint4*)(lmask + lane * 16);
        m0 = __ballot((ma.x | ma.y | ma.z | ma.w) != 0u);
        if (C::RUNS == 128) {
            const uint4 mb = *(const uint4*)(lmask + 1024 + lane * 16);
            m1 = __ballot((mb.x | mb.y | mb.z | mb.w) != 0u);
        }
    }
    VOX_STAMP(1);
    VOX_STAMPV(4, __popcll(m0) + __popcll(m1));
    if ((m0 | m1) == 0ull) {                                                   // empty brick (the same answer in every wave)
        VOX_STAMPV(6, __builtin_amdgcn_s_memrealtime());
        if (p.stats && t < 64) p.stats[(size_t)blockIdx.x * 64 + t] = 0.f;
        return;
    }

    // ---- slab: 4 planes x (TY + 2) rows x [16 B pad | V sites x 8 B | pad]; rows / planes outside the grid are out-of-range buffer
    // offsets (zeros), so all loads of a thread are issued back to back
    {
        const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
        uint4 pre[C::MAXC];
        int dst[C::MAXC];
#pragma unroll
        for (int u = 0; u < C::MAXC; ++u) {
            const int c = t + u * 256;
            const int zz = (c >= C::ITEMS) + (c >= 2 * C::ITEMS) + (c >= 3 * C::ITEMS);
            const int i = c - zz * C::ITEMS;
            const int yy = i / C::CPR, ch = i % C::CPR;
            const int gz = z0 - 1 + zz, gy = y0 - 1 + yy;
            const bool inside = c < 4 * C::ITEMS;
            dst[u] = inside ? zz * C::PLANE + yy * C::PITCH + 16 + ch * 16 : -1;
            const bool ok = inside && (unsigned)gz < (unsigned)V && (unsigned)gy < (unsigned)V;
            const unsigned voff = ok ? (unsigned)(((((b * V + gz) * V + gy) * V) + 2 * ch) * 8) : 0x80000000u;
            pre[u] = make_uint4(0u, 0u, 0u, 0u);
            if (!VOX_ABL(4)) pre[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(irs, voff, 0, 0));
        }
        // the one-site x halo left and right of every row
        if (t < 4 * (TY + 2)) {
            const int zz = t / (TY + 2);
            char* r = slab + zz * C::PLANE + (t - zz * (TY + 2)) * C::PITCH;
            *(uint2*)(r + 8) = make_uint2(0u, 0u);
            *(uint2*)(r + 16 + V * 8) = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < C::MAXC; ++u)
            if (dst[u] >= 0) *(uint4*)(slab + dst[u]) = pre[u];
    }
    // A fragment (kd, p, ct): row i = fr is output channel 8 (i >> 2) + 4 ct + (i & 3), so that the accumulator registers r = 0..3 of
    // lane (fr, fq) are channels 8 fq + 4 ct + r of site fr; elements 0-3 / 4-7 = the four channels of tap (kd, kh = fq, kw = 2 p) /
    // (kd, fq, 2 p + 1); k-group 3 and kw = 3 are zeros
    v8 wf[3][2][2];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const char* row = wbuf + (8 * (fr >> 2) + 4 * ct + (fr & 3)) * C::WROW + ((kd * 3 + (fq == 3 ? 0 : fq)) * 3 + 2 * pp) * 8;
                uint2 lo = *(const uint2*)row, hi = make_uint2(0u, 0u);
                if (pp == 0) hi = *(const uint2*)(row + 8);
                if (fq == 3) { lo = make_uint2(0u, 0u); hi = lo; }
                wf[kd][pp][ct] = __builtin_bit_cast(v8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }

    // ---- this wave's runs: the active runs whose ordinal (in run order) is the wave's number modulo 4
    unsigned long long mine0, mine1 = 0ull;
    {
        const unsigned long long below = (1ull << lane) - 1ull;
        const int o0 = __popcll(m0 & below);
        mine0 = __ballot(((m0 >> lane) & 1ull) && (o0 & 3) == wave);
        if (C::RUNS == 128) {
            const int o1 = __popcll(m0) + __popcll(m1 & below);
            mine1 = __ballot(((m1 >> lane) & 1ull) && (o1 & 3) == wave);
        }
    }
    // lane part of every fragment address: slab row kh = fq (the zero-weight k-group 3 re-reads row 1: finite data, and an odd
    // number of row pitches away from k-group 2 it shares an LDS cycle with), site fr
    const int lofs = (fq == 3 ? 1 : fq) * C::PITCH + fr * 8;
    __syncthreads();

    VOX_STAMP(2);
    if (VOX_ABL(16)) return;
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = cs0, cq0 = cs0, cq1 = cs0;
#pragma unroll 1
    for (int g = 0; g < (C::RUNS == 128 ? 2 : 1); ++g) {
        unsigned long long m = g ? mine1 : mine0;
#pragma unroll 1
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            const int r = g * 64 + bit;
            const int zl = r / C::HALF, idx = r % C::HALF, yl = idx / C::RPR, xr = idx % C::RPR;
            // operand of tap (kd, kh, kw) for site x0 + fr: slab plane zl + kd, row yl + kh, byte 16 + (x0 + fr + kw - 1) * 8
            const char* sb = slab + (zl * C::PLANE + yl * C::PITCH + 8 + xr * 128) + lofs;
            v8 bf[3][2];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                uint2 t0 = make_uint2(bit, 0u), t1 = t0, t2 = t0;
                if (!VOX_ABL(8)) {
                    t0 = *(const uint2*)(sb + kd * C::PLANE);
                    t1 = *(const uint2*)(sb + kd * C::PLANE + 8);
                    t2 = *(const uint2*)(sb + kd * C::PLANE + 16);
                }
                bf[kd][0] = __builtin_bit_cast(v8, make_uint4(t0.x, t0.y, t1.x, t1.y));
                bf[kd][1] = __builtin_bit_cast(v8, make_uint4(t2.x, t2.y, t2.x, t2.y));       // elements 4-7 meet zero weights (kw = 3)
            }
            const int live = lmask[r * 16 + fr];
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    if (VOX_ABL(1)) { a0[0] += __builtin_bit_cast(float, ((uint4)__builtin_bit_cast(uint4, bf[kd][pp])).x); continue; }
                    a0 = MM::mma(wf[kd][pp][0], bf[kd][pp], a0);
                    a1 = MM::mma(wf[kd][pp][1], bf[kd][pp], a1);
                }
            if (live) {
                typedef E e4 __attribute__((ext_vector_type(4)));
                const e4 h0 = __builtin_convertvector(a0, e4), h1 = __builtin_convertvector(a1, e4);     // packed conversions
                const v8 o8 = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                const size_t site = ((size_t)(b * V + z0 + zl) * V + y0 + yl) * V + xr * 16 + fr;
                if (!VOX_ABL(2)) *(v8*)((AT*)p.out + site * 32 + fq * 8) = o8;
                f32x4 r0 = {(float)o8[0], (float)o8[1], (float)o8[2], (float)o8[3]};
                f32x4 r1 = {(float)o8[4], (float)o8[5], (float)o8[6], (float)o8[7]};
                cs0 += r0; cq0 += r0 * r0;
                cs1 += r1; cq1 += r1 * r1;
            }
        }
    }

    VOX_STAMP(3);
    VOX_STAMPV(6, __builtin_amdgcn_s_memrealtime());
    if (p.stats) {                                                             // one record per workgroup
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s_ = ct ? cs1[r] : cs0[r], q_ = ct ? cq1[r] : cq0[r];
                s_ += vox_row_ror<8>(s_); q_ += vox_row_ror<8>(q_);
                s_ += vox_row_ror<4>(s_); q_ += vox_row_ror<4>(q_);
                s_ += vox_row_ror<2>(s_); q_ += vox_row_ror<2>(q_);
                s_ += vox_row_ror<1>(s_); q_ += vox_row_ror<1>(q_);
                if (fr == 0) {
                    const int ch = 8 * fq + 4 * ct + r;
                    red[(wave * 32 + ch) * 2 + 0] = s_;
                    red[(wave * 32 + ch) * 2 + 1] = q_;
                }
            }
        __syncthreads();
        if (t < 32) {
            float s_ = 0.f, q_ = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s_ += red[(w * 32 + t) * 2]; q_ += red[(w * 32 + t) * 2 + 1]; }
            p.stats[(size_t)blockIdx.x * 64 + t] = s_;
            p.stats[(size_t)blockIdx.x * 64 + 32 + t] = q_;
        }
    }
}


// ================================================================================================ level 0: weight gradient
// conv_vox0_wgrad_kernel.  dW[co][kd, kh, kw, ci] = sum over active sites of dOut[site][co] * In[site + tap][ci] for level 0 (3 -> 32
// channels, sparse_cnn.py:12-14).  Through conv_wgrad_kernel<64,256,1> this is an 8-byte gather per (site, tap) into tiles of which
// three quarters are padding: 37-65 us at the bench shapes, the slowest MFMA kernel of the voxel tower's backward.  Here - the 3D
// twin of conv_stem_wgrad_kernel - the brick's input sites are staged once in the forward kernel's LDS slab and BOTH operands are
// read transposed (ds_read_b64_tr_b16: the contraction index is the SITE):
//   * contraction groups of 32 x-consecutive sites (two runs); groups without an active site are skipped, dOut rows of inactive sites
//     are zeroed while the group's [32 sites][32 channels] tile is staged in the wave's own 2 KB of LDS (they may never have been written);
//   * the A fragments (dOut^T, two 16-channel tiles) come from that tile, the B fragment of kernel row (kd, kh) straight from the slab:
//     row = site (8 B apart), 16 columns = the 4 channels of sites x - 1 .. x + 2, i.e. taps kw = 0, 1, 2 and one padding tap: 9 kernel
//     rows x 2 channel tiles = 18 MFMAs per group against 22 transposed reads;
//   * a persistent workgroup (bricks w, w + G, ...; same rotation as the forward kernel) keeps its 32 x 144 partial sums in registers
//     (72 per lane) over all its bricks and groups; the four waves' sums are added in wave order through LDS and leave as ONE fp32 slab
//     [32][9 rows x 4 taps x 4 channels] per workgroup, which tri_wgrad_reduce_grouped sums (kernel rows padded to 4 taps:
//     TriWgradReduce.kw_real = 3, kw_shift = 2).
struct Vox0WgradArgs {
    const void* in;            // [B, V, V, V, 4] 16-bit, zeros at inactive sites
    const void* dout;          // [B, V, V, V, 32]; rows of inactive sites are not read
    const uint8_t* mask;       // [B * V^3] site mask, or NULL (every site active)
    float* slab;               // [grid][32][144]
    int B, nbricks;
    unsigned in_bytes;
};
typedef short vox_s16x4 __attribute__((ext_vector_type(4)));
typedef short vox_s16x8 __attribute__((ext_vector_type(8)));
template <typename V8>
__device__ __forceinline__ V8 vox_tr_frag(const char* lo_addr, const char* hi_addr) {
    const vox_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vox_s16x4 __attribute__((address_space(3)))*)lo_addr);
    const vox_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((vox_s16x4 __attribute__((address_space(3)))*)hi_addr);
    const vox_s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(V8, r);
}
template <typename AT, int V, int TY>
__global__ __launch_bounds__(256, 3) void conv_vox0_wgrad_kernel(const Vox0WgradArgs p) {
    typedef Vox0Cfg<V, TY> C;
    typedef typename OpOf<AT>::E E;
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    constexpr int NG = C::RUNS / 2;                                            // 32-site groups per brick (32 or 64)
    constexpr int PS = (V / 2) * C::NYB;                                       // bricks per sample
    static_assert(C::SLAB >= 4 * 6 * 64 * 16, "the slab doubles as the cross-wave reduction buffer");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fg = lane >> 4, fqq = fr >> 2, fp = fr & 3;
    char* const slab = smem;
    uint8_t* const lmask = (uint8_t*)(smem + C::SLAB);                         // [RUNS][16 sites]
    char* const ytile = (char*)(lmask + C::RUNS * 16) + wave * 2048;           // this wave's [32 sites][32 channels] dOut tile
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);

    f32x4 acc[9][2];
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) { acc[kk][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[kk][1] = acc[kk][0]; }
    // lane constants: dOut tile pieces this lane stages (site e >> 2, 16-byte part e & 3 for e = lane, lane + 64; the two 32-byte halves
    // of a row swap places for rows 8-15 and 24-31, so the two 16-lane groups one LDS cycle serves hit disjoint banks) and fragment rows
    int ysite[2], ydst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = lane + 64 * u;
        ysite[u] = e >> 2;
        ydst[u] = (e >> 2) * 64 + (((((e >> 1) & 1) ^ ((e >> 5) & 1)) << 5) | ((e & 1) << 4));
    }
    const int arow_lo = 8 * fg + fqq, arow_hi = arow_lo + 4;
    int aoff[2][2];                                                            // [channel tile][lo / hi row]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        aoff[ct][0] = arow_lo * 64 + ((ct ^ ((arow_lo >> 3) & 1)) << 5) + fp * 8;
        aoff[ct][1] = arow_hi * 64 + ((ct ^ ((arow_hi >> 3) & 1)) << 5) + fp * 8;
    }
    const int boff_lo = (arow_lo + fp) * 8, boff_hi = (arow_hi + fp) * 8;      // slab: site row + column quad

#pragma unroll 1
    for (int bi = blockIdx.x; bi < p.nbricks; bi += gridDim.x) {
        const int b = bi / PS;
        const int sp = (bi % PS + b * (PS / 2 + 1)) % PS;
        const int yb = sp % C::NYB, zp = sp / C::NYB;
        const int z0 = zp * 2, y0 = yb * TY;
        __syncthreads();                                                       // the previous brick's reads of lmask / slab are done
        {
            uint4 mv = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
            if (t < C::RUNS && p.mask) mv = *(const uint4*)(p.mask + ((size_t)(b * V + z0 + t / C::HALF) * V + y0) * V + (t % C::HALF) * 16);
            if (t < C::RUNS) *(uint4*)(lmask + t * 16) = mv;
        }
        __syncthreads();
        unsigned long long mg;                                                 // active groups of the brick
        {
            bool any = false;
            if (lane < NG) {
                const uint4 ma = *(const uint4*)(lmask + lane * 32), mb = *(const uint4*)(lmask + lane * 32 + 16);
                any = (ma.x | ma.y | ma.z | ma.w | mb.x | mb.y | mb.z | mb.w) != 0u;
            }
            mg = __ballot(any);
        }
        if (mg == 0ull) continue;                                              // empty brick (the same answer in every wave)
        {   // slab fill, as conv_vox0_kernel
            uint4 pre[C::MAXC];
            int dst[C::MAXC];
#pragma unroll
            for (int u = 0; u < C::MAXC; ++u) {
                const int c = t + u * 256;
                const int zz = (c >= C::ITEMS) + (c >= 2 * C::ITEMS) + (c >= 3 * C::ITEMS);
                const int i = c - zz * C::ITEMS;
                const int yy = i / C::CPR, ch = i % C::CPR;
                const int gz = z0 - 1 + zz, gy = y0 - 1 + yy;
                const bool inside = c < 4 * C::ITEMS;
                dst[u] = inside ? zz * C::PLANE + yy * C::PITCH + 16 + ch * 16 : -1;
                const bool ok = inside && (unsigned)gz < (unsigned)V && (unsigned)gy < (unsigned)V;
                const unsigned voff = ok ? (unsigned)(((((b * V + gz) * V + gy) * V) + 2 * ch) * 8) : 0x80000000u;
                pre[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(irs, voff, 0, 0));
            }
            if (t < 4 * (TY + 2)) {
                const int zz = t / (TY + 2);
                char* r = slab + zz * C::PLANE + (t - zz * (TY + 2)) * C::PITCH;
                *(uint2*)(r + 8) = make_uint2(0u, 0u);
                *(uint2*)(r + 16 + V * 8) = make_uint2(0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < C::MAXC; ++u)
                if (dst[u] >= 0) *(uint4*)(slab + dst[u]) = pre[u];
        }
        // this wave's groups: the active ones whose ordinal is the wave's number modulo 4
        unsigned long long mine;
        {
            const unsigned long long below = (1ull << lane) - 1ull;
            mine = __ballot(((mg >> lane) & 1ull) && (__popcll(mg & below) & 3) == wave);
        }
        __syncthreads();                                                       // slab complete

        auto ybase = [&](int gi) -> const char* {                              // first dOut row of group gi
            const int r0 = 2 * gi, zl = r0 / C::HALF, idx = r0 % C::HALF, yl = idx / C::RPR, xr = idx % C::RPR;
            return (const char*)p.dout + ((((size_t)(b * V + z0 + zl) * V + y0 + yl) * V) + xr * 16) * 64;
        };
        uint4 ld[2];
        int cur = mine ? __builtin_ctzll(mine) : -1;
        if (cur >= 0) {
            mine &= mine - 1;
            const char* src = ybase(cur);
#pragma unroll
            for (int u = 0; u < 2; ++u) ld[u] = lmask[cur * 32 + ysite[u]] ? *(const uint4*)(src + (lane + 64 * u) * 16) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll 1
        while (cur >= 0) {
            const int nxt = mine ? __builtin_ctzll(mine) : -1;
            if (nxt >= 0) mine &= mine - 1;
            __builtin_amdgcn_wave_barrier();                                   // the previous group's fragment reads are issued (in order per wave)
#pragma unroll
            for (int u = 0; u < 2; ++u) *(uint4*)(ytile + ydst[u]) = ld[u];
            if (nxt >= 0) {                                                    // the next group's rows fly under this group's MFMAs
                const char* src = ybase(nxt);
#pragma unroll
                for (int u = 0; u < 2; ++u) ld[u] = lmask[nxt * 32 + ysite[u]] ? *(const uint4*)(src + (lane + 64 * u) * 16) : make_uint4(0u, 0u, 0u, 0u);
            }
            __builtin_amdgcn_wave_barrier();
            const int r0 = 2 * cur, zl = r0 / C::HALF, idx = r0 % C::HALF, yl = idx / C::RPR, xr = idx % C::RPR;
            const char* sb = slab + zl * C::PLANE + yl * C::PITCH + 8 + xr * 128;   // site x0 - 1 of slab row (zl, yl)
            v8 af[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) af[ct] = vox_tr_frag<v8>(ytile + aoff[ct][0], ytile + aoff[ct][1]);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const char* rb = sb + kd * C::PLANE + kh * C::PITCH;
                    const v8 bf = vox_tr_frag<v8>(rb + boff_lo, rb + boff_hi);
                    acc[kd * 3 + kh][0] = MM::mma(af[0], bf, acc[kd * 3 + kh][0]);
                    acc[kd * 3 + kh][1] = MM::mma(af[1], bf, acc[kd * 3 + kh][1]);
                }
            cur = nxt;
        }
    }

    // ---- the four waves' sums, added in wave order: three passes of 6 accumulator tiles through the (now idle) slab
    f32x4* const red = (f32x4*)slab;                                           // [4 waves][6 tiles][64 lanes]
    float* const out = p.slab + (size_t)blockIdx.x * 32 * 144;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) red[(wave * 6 + j * 2 + ct) * 64 + lane] = acc[pass * 3 + j][ct];
        __syncthreads();
        for (int e = t; e < 6 * 64; e += 256) {
            const int tile = e >> 6, ln = e & 63;
            f32x4 sum = red[(0 * 6 + tile) * 64 + ln];
#pragma unroll
            for (int w = 1; w < 4; ++w) sum += red[(w * 6 + tile) * 64 + ln];
            const int kk = pass * 3 + (tile >> 1), ct = tile & 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(ct * 16 + (ln >> 4) * 4 + r) * 144 + kk * 16 + (ln & 15)] = sum[r];
        }
    }
}

static bool vox_disabled(const char* name) {
    const char* e = getenv(name);
    return e && e[0] == '1';
}

// brick rows per workgroup: 32^3 -> 16 (64 runs), 64^3 -> 8 (64 runs; TRICOLO_VOX0_TY=16: 128 runs), 128^3 -> 8 (128 runs)
static int vox0_ty(int V) {
    static int env = -1;
    if (env < 0) env = 0;
    if (V == 64 && env == 16) return 16;
    return V == 32 ? 16 : 8;
}

bool tri_internal_vox0_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                                int pd, int ph, int pw, TriVox0Geom* g) {
    static int off = -1;
    if (off < 0) off = vox_disabled("TRICOLO_NO_VOX0") ? 1 : 0;               // A/B switch: level 0 stays on conv_igemm_kernel
    if (off) return false;
    const int V = ID;
    if (IH != V || IW != V || OD != V || OH != V || OW != V || (V != 32 && V != 64 && V != 128)) return false;
    if (cin != 4 || cout != 32 || KD != 3 || KH != 3 || KW != 3 || stride != 1 || pd != 1 || ph != 1 || pw != 1) return false;
    if ((long)B * V * V * V * 8 >= (1L << 31)) return false;                  // 32-bit buffer offsets
    g->V = V;
    g->TY = vox0_ty(V);
    g->grid = B * (V / 2) * (V / g->TY);
    return true;
}

template <typename AT, int V, int TY>
static int vox0_launch_t(const Vox0Args& a, int grid, hipStream_t stream) {
    typedef Vox0Cfg<V, TY> C;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv_vox0_kernel<AT, V, TY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
        attr = true;
    }
    conv_vox0_kernel<AT, V, TY><<<grid, 256, C::SMEM, stream>>>(a);
    return tri_check_launch("tri_conv(vox0)");
}

int tri_internal_vox0_launch(const TriVox0Geom& g, int B, const void* in, const void* w, int kpad, void* out, const uint8_t* mask, float* stats,
                             int act_fmt, hipStream_t stream) {
    Vox0Args a{};
    a.in = in; a.w = w; a.out = out; a.mask = mask; a.stats = stats;
    a.B = B; a.Kpad = kpad;
    a.in_bytes = (unsigned)((size_t)B * g.V * g.V * g.V * 8);
#ifdef VOX_PROBE
    { const char* e = getenv("TRICOLO_VOX_ABL"); a.abl = e ? atoi(e) : 0; }
    { const char* e = getenv("TRICOLO_VOX_DBG"); a.dbg = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
#define TRI_VOX0(V_, TY_)                                                                                          \
    if (g.V == V_ && g.TY == TY_)                                                                                  \
        return act_fmt == TRI_FMT_F16 ? vox0_launch_t<f16_t, V_, TY_>(a, g.grid, stream) : vox0_launch_t<bf16_t, V_, TY_>(a, g.grid, stream);
    TRI_VOX0(32, 16)
    TRI_VOX0(64, 8)
    TRI_VOX0(64, 16)
    TRI_VOX0(128, 8)
#undef TRI_VOX0
    tri_set_error("conv(vox0): brick shape not instantiated");
    return TRI_ERR_UNSUPPORTED;
}

template <typename AT, int V, int TY>
static int vox0_wgrad_launch_t(const Vox0WgradArgs& a, int grid, hipStream_t stream) {
    typedef Vox0Cfg<V, TY> C;
    constexpr size_t SMEM = (size_t)C::SLAB + C::RUNS * 16 + 4 * 2048;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv_vox0_wgrad_kernel<AT, V, TY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM);
        attr = true;
    }
    conv_vox0_wgrad_kernel<AT, V, TY><<<grid, 256, SMEM, stream>>>(a);
    return tri_check_launch("tri_conv_wgrad(vox0)");
}
// workgroups (= fp32 slabs [32][144]) of the level-0 weight-gradient kernel: persistent, three per CU
int tri_internal_vox0_wgrad_grid(const TriVox0Geom& g) {
    static int off = -1;
    if (off < 0) off = vox_disabled("TRICOLO_NO_VOX0_WGRAD") ? 1 : 0;         // A/B switch: level 0 stays on conv_wgrad_kernel
    if (off) return 0;
    const int slots = 3 * tri_internal_num_cus();
    return g.grid < slots ? g.grid : slots;
}
int tri_internal_vox0_wgrad_launch(const TriVox0Geom& g, int grid, int B, const void* in, const void* dout, const uint8_t* mask, float* slab,
                                   int act_fmt, hipStream_t stream) {
    Vox0WgradArgs a{};
    a.in = in; a.dout = dout; a.mask = mask; a.slab = slab;
    a.B = B; a.nbricks = g.grid;
    a.in_bytes = (unsigned)((size_t)B * g.V * g.V * g.V * 8);
#define TRI_VOX0W(V_, TY_)                                                                                         \
    if (g.V == V_ && g.TY == TY_)                                                                                  \
        return act_fmt == TRI_FMT_F16 ? vox0_wgrad_launch_t<f16_t, V_, TY_>(a, grid, stream) : vox0_wgrad_launch_t<bf16_t, V_, TY_>(a, grid, stream);
    TRI_VOX0W(32, 16)
    TRI_VOX0W(64, 8)
    TRI_VOX0W(64, 16)
    TRI_VOX0W(128, 8)
#undef TRI_VOX0W
    tri_set_error("conv_wgrad(vox0): brick shape not instantiated");
    return TRI_ERR_UNSUPPORTED;
}
