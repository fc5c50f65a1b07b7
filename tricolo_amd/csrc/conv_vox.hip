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
//   * brick = 2 z-planes x TY rows x all V columns = 128 runs (TY = 1024 / V); one brick per workgroup, workgroups of empty
//     bricks leave after reading 2 KB of site mask; the dispatcher balances the rest (2-3 workgroups per CU overlap each other's
//     slab fill, MFMAs and stores);
//   * with four stored channels one MFMA k-step (32) is 8 taps; the 27 taps (+ 5 zero-weight slots) are ORDERED so that the two
//     16-lane groups served by one LDS cycle of a ds_read_b64 read rows that are an odd number of row / plane pitches apart and
//     both pitches are 128 mod 256 bytes: conflict-free fragment reads without a swizzle (derivation at VOX0_WTAP);
//   * the whole filter bank (32 x 128 k) is 32 registers of MFMA A fragments, loaded once per workgroup, with the output channels
//     permuted so that a lane ends up with 8 CONSECUTIVE channels of one site: one 16-byte store per lane, full 64-byte rows per
//     4 lanes;
//   * BatchNorm sums (of the values as stored) stay in registers over the workgroup's runs: one record per workgroup.
#include "common.h"
#include <stdlib.h>
#include "conv_vox.h"

template <int N>
__device__ __forceinline__ float vox_row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}

// slot (k-step s, k-group fq, half e) = s * 8 + fq * 2 + e  ->  tap whose 4 channels fill elements 4 e .. 4 e + 3 of that lane's
// fragment.  ds_read_b64 serves lanes 0-31 (fq 0, 1) and 32-63 (fq 2, 3) in one LDS cycle each; a 16-lane group reads 16
// consecutive sites = 128 contiguous bytes = 32 banks, so the two groups of a cycle are conflict-free iff their addresses differ
// by 128 mod 256 bytes.  Row pitch and plane pitch are both 128 mod 256, so that holds iff the two taps share kw and differ in
// (kd + kh) parity.  Per kw there are 5 even and 4 odd (kd, kh): 4 even-odd pairs, the centre row left over - it and the five
// unused slots are paired with zero-weight slots that read a valid (finite) site of the opposite parity.
__device__ const signed char VOX0_WTAP[32] = {0, 6, 3, 9, 18, 24, 15, 21, 1, 7, 4, 10, 19, 25, 16, 22,
                                              2, 8, 5, 11, 20, 26, 17, 23, 12, 13, -1, -1, 14, -1, -1, -1};     // weight tap (-1: zeros)
__device__ const signed char VOX0_ATAP[32] = {0, 6, 3, 9, 18, 24, 15, 21, 1, 7, 4, 10, 19, 25, 16, 22,
                                              2, 8, 5, 11, 20, 26, 17, 23, 12, 13, 3, 4, 14, 13, 5, 4};         // tap whose site is read

struct Vox0Args {
    const void* in;            // [B, V, V, V, 4] 16-bit, zeros at inactive sites
    const void* w;             // packed operand rows [32][Kpad] (k = tap * 4 + channel)
    void* out;                 // [B, V, V, V, 32]; rows of inactive sites are not written
    const uint8_t* mask;       // [B * V^3] site mask, or NULL (every site active)
    float* stats;              // [grid][2][32] or NULL
    int B, V, TY, nyb, Kpad;
    int pitch, plane, slab_bytes, vshift;
};

template <typename AT>
__global__ __launch_bounds__(256, 2) void conv_vox0_kernel(const Vox0Args p) {
    typedef typename OpOf<AT>::E E;
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
    const int V = p.V, TY = p.TY;
    int bid = blockIdx.x;
    const int yb = bid % p.nyb;
    bid /= p.nyb;
    const int hz = V >> 1;
    const int zp = bid % hz, b = bid / hz;
    const int z0 = zp * 2, y0 = yb * TY;
    char* const slab = smem;
    uint8_t* const lmask = (uint8_t*)(smem + p.slab_bytes);                    // [128 runs][16 sites]
    float* const red = (float*)(lmask + 2048);                                 // [4 waves][32][2]

    // ---- site mask of the brick: run r = (plane r >> 6, row (r & 63) >> vshift, x-run r & (V / 16 - 1)) is 16 contiguous bytes
    if (t < 128) {
        uint4 mv = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
        if (p.mask) mv = *(const uint4*)(p.mask + ((size_t)(b * V + z0 + (t >> 6)) * V + y0) * V + (t & 63) * 16);
        *(uint4*)(lmask + t * 16) = mv;
    }
    __syncthreads();
    const uint4 ma = *(const uint4*)(lmask + lane * 16), mb = *(const uint4*)(lmask + 1024 + lane * 16);
    const unsigned long long m0 = __ballot((ma.x | ma.y | ma.z | ma.w) != 0u), m1 = __ballot((mb.x | mb.y | mb.z | mb.w) != 0u);
    if ((m0 | m1) == 0ull) {                                                   // empty brick (the same answer in every wave)
        if (p.stats && t < 64) p.stats[(size_t)blockIdx.x * 64 + t] = 0.f;
        return;
    }

    // ---- filter bank -> registers.  A fragment (s, ct): row i = fr is output channel 8 (i >> 2) + 4 ct + (i & 3), so that the
    // accumulator registers r = 0..3 of lane (fr, fq) are channels 8 fq + 4 ct + r of site fr
    v8 wf[4][2];
    {
        const uint16_t* w = (const uint16_t*)p.w;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ta = VOX0_WTAP[s * 8 + fq * 2], tb = VOX0_WTAP[s * 8 + fq * 2 + 1];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const uint16_t* row = w + (size_t)(8 * (fr >> 2) + 4 * ct + (fr & 3)) * p.Kpad;
                uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
                if (ta >= 0) lo = *(const uint2*)(row + ta * 4);
                if (tb >= 0) hi = *(const uint2*)(row + tb * 4);
                wf[s][ct] = __builtin_bit_cast(v8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
        }
    }
    // per-lane slab offsets of the two taps of every k-step (relative to the run's first site)
    int coff[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int tp = VOX0_ATAP[s * 8 + fq * 2 + e];
            const int kw = tp % 3, kh = (tp / 3) % 3, kd = tp / 9;
            coff[s][e] = (kd - 1) * p.plane + (kh - 1) * p.pitch + (kw - 1) * 8 + fr * 8;
        }

    // ---- slab: 4 planes x (TY + 2) rows x [16 B pad | V sites x 8 B | pad]; rows / planes outside the grid are zeros
    {
        const int cpr = V >> 1, cshift = p.vshift + 3;                         // 16-byte chunks per row (2 sites each)
        const int items = (TY + 2) * cpr;                                      // per plane
        constexpr int MAXC = 10;
        uint4 pre[MAXC];
        int dst[MAXC];
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            const int c = t + u * 256;
            const int zz = (c >= items) + (c >= 2 * items) + (c >= 3 * items);
            const int i = c - zz * items;
            const int yy = i >> cshift, ch = i & (cpr - 1);
            const int gz = z0 - 1 + zz, gy = y0 - 1 + yy;
            const bool inside = c < 4 * items;
            dst[u] = inside ? zz * p.plane + yy * p.pitch + 16 + ch * 16 : -1;
            pre[u] = make_uint4(0u, 0u, 0u, 0u);
            if (inside && (unsigned)gz < (unsigned)V && (unsigned)gy < (unsigned)V)
                pre[u] = *(const uint4*)((const char*)p.in + ((((size_t)(b * V + gz) * V + gy) * V) + 2 * ch) * 8);
        }
        // the one-site x halo left and right of every row
        const int nrows = 4 * (TY + 2);
        if (t < nrows) {
            const int zz = (t >= TY + 2) + (t >= 2 * (TY + 2)) + (t >= 3 * (TY + 2));
            char* r = slab + zz * p.plane + (t - zz * (TY + 2)) * p.pitch;
            *(uint2*)(r + 8) = make_uint2(0u, 0u);
            *(uint2*)(r + 16 + V * 8) = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < MAXC; ++u)
            if (dst[u] >= 0) *(uint4*)(slab + dst[u]) = pre[u];
    }
    __syncthreads();

    // ---- runs: every wave walks the brick's active runs and takes those whose ordinal is its own modulo 4
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = cs0, cq0 = cs0, cq1 = cs0;
    const int xmask = (1 << p.vshift) - 1;
    int ord = 0;
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
        unsigned long long m = g ? m1 : m0;
#pragma unroll 1
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            if ((ord++ & 3) != wave) continue;
            const int yl = bit >> p.vshift, xr = bit & xmask;
            const char* sb = slab + (g + 1) * p.plane + (yl + 1) * p.pitch + 16 + xr * 128;
            v8 bf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const uint2 lo = *(const uint2*)(sb + coff[s][0]), hi = *(const uint2*)(sb + coff[s][1]);
                bf[s] = __builtin_bit_cast(v8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
            const int live = lmask[(g * 64 + bit) * 16 + fr];
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0 = MM::mma(wf[s][0], bf[s], a0);
                a1 = MM::mma(wf[s][1], bf[s], a1);
            }
            if (live) {
                v8 o8;
                o8[0] = (E)a0[0]; o8[1] = (E)a0[1]; o8[2] = (E)a0[2]; o8[3] = (E)a0[3];
                o8[4] = (E)a1[0]; o8[5] = (E)a1[1]; o8[6] = (E)a1[2]; o8[7] = (E)a1[3];
                const size_t site = ((size_t)(b * V + z0 + g) * V + y0 + yl) * V + xr * 16 + fr;
                *(v8*)((AT*)p.out + site * 32 + fq * 8) = o8;
                f32x4 r0 = {(float)o8[0], (float)o8[1], (float)o8[2], (float)o8[3]};
                f32x4 r1 = {(float)o8[4], (float)o8[5], (float)o8[6], (float)o8[7]};
                cs0 += r0; cq0 += r0 * r0;
                cs1 += r1; cq1 += r1 * r1;
            }
        }
    }

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

static bool vox_disabled(const char* name) {
    const char* e = getenv(name);
    return e && e[0] == '1';
}

bool tri_internal_vox0_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                                int pd, int ph, int pw, TriVox0Geom* g) {
    static int off = -1;
    if (off < 0) off = vox_disabled("TRICOLO_NO_VOX0") ? 1 : 0;               // A/B switch: level 0 stays on conv_igemm_kernel
    if (off) return false;
    const int V = ID;
    if (IH != V || IW != V || OD != V || OH != V || OW != V || (V != 32 && V != 64 && V != 128)) return false;
    if (cin != 4 || cout != 32 || KD != 3 || KH != 3 || KW != 3 || stride != 1 || pd != 1 || ph != 1 || pw != 1) return false;
    if ((long)B * (V / 2) * 16 >= (1L << 31)) return false;
    g->V = V;
    g->TY = 1024 / V;
    g->nyb = V / g->TY;
    g->pitch = V * 8 + 128;                                                   // 128 mod 256 for V = 32, 64, 128
    g->plane = (g->TY + 2) * g->pitch;
    if (((g->plane >> 7) & 1) == 0) g->plane += 128;                          // plane pitch 128 mod 256 too
    g->slab_bytes = 4 * g->plane;
    g->vshift = V == 32 ? 1 : (V == 64 ? 2 : 3);
    g->grid = B * (V / 2) * g->nyb;
    return true;
}

int tri_internal_vox0_launch(const TriVox0Geom& g, int B, const void* in, const void* w, int kpad, void* out, const uint8_t* mask, float* stats,
                             int act_fmt, hipStream_t stream) {
    Vox0Args a{};
    a.in = in; a.w = w; a.out = out; a.mask = mask; a.stats = stats;
    a.B = B; a.V = g.V; a.TY = g.TY; a.nyb = g.nyb; a.Kpad = kpad;
    a.pitch = g.pitch; a.plane = g.plane; a.slab_bytes = g.slab_bytes; a.vshift = g.vshift;
    const size_t smem = (size_t)g.slab_bytes + 2048 + 4 * 32 * 2 * sizeof(float);
    static size_t attr_f16 = 0, attr_bf16 = 0;
    if (act_fmt == TRI_FMT_F16) {
        if (smem > attr_f16) { hipFuncSetAttribute((const void*)conv_vox0_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_f16 = smem; }
        conv_vox0_kernel<f16_t><<<g.grid, 256, smem, stream>>>(a);
    } else {
        if (smem > attr_bf16) { hipFuncSetAttribute((const void*)conv_vox0_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_bf16 = smem; }
        conv_vox0_kernel<bf16_t><<<g.grid, 256, smem, stream>>>(a);
    }
    return tri_check_launch("tri_conv(vox0)");
}
