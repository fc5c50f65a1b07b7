// conv_s2g_kernel: the 3x3 / stride 2 / pad 1 convolutions that open layer3 and layer4 of the ResNet trunk (128 -> 256 channels on 16x16
// maps, 256 -> 512 on 8x8 maps at the bench shape; torchvision resnet18 layer3[0].conv1 / layer4[0].conv1 behind mv_cnn.py:18-20), forward,
// 16-bit storage, gfx950.
//
// Through conv_dma_kernel these layers are im2col GEMMs with 128 x 64 tiles whose A tile gathers stride-2 taps (every input pixel crosses
// the L2 -> LDS path 2.25 x, every tile re-streams its weight panel through LDS): 21 / 28 us at the bench shape, 350 / 260 TFLOP/s.  This is
// conv_voxg_kernel's scheme (conv_voxg.hip) for a dense 2D stride-2 layer:
//   * a workgroup owns a UNIT of whole images (3 images of 8x8 outputs = 192 rows, 6 images of 4x4 outputs = 96 rows) and 64 output
//     channels; the unit's input images, 32 channels at a time, are stationary in LDS in a SPACE-TO-DEPTH slab: four parity planes
//     (py, px) of (OH + 1) x (OW + 1) sites per image, plane (py, px) holding the zero-padded pixel (2 j + py, 2 i + px).  Tap (ky, kx) of
//     output (oy, ox) reads padded pixel (2 oy + ky, 2 ox + kx) = site (oy + (ky >> 1), ox + (kx >> 1)) of plane (ky & 1, kx & 1): a row's
//     slab address is ITS base + a per-tap constant, and the rows of a fragment (consecutive ox) read consecutive sites of one plane -
//     the stride-2 gather costs nothing in the MFMA loop and no LDS bank conflicts beyond the row wrap;
//   * the slab is filled as in conv_voxg_kernel: every thread requests its 16-byte pieces of the NEXT chunk (one per 16 output rows of the
//     unit: four pieces per input pixel, coalesced 64-byte runs) before the current chunk's MFMAs and writes them to the other buffer after
//     them; the padding sites are zeroed once per workgroup.  (An LDS-DMA fill through a fifth, producer wave was built first: with five
//     waves a wave has 256 registers instead of 512 and the 12-fragment variant spilled whatever was cut - ring depth, B-fragment window.)
//   * weights as in conv_voxg_kernel: FRAGMENT-MAJOR packed operand, straight from L2 into MFMA registers through a ring of ten
//     fragments x two channel tiles per wave; waves split (channel pair, tap parity): 2 x 2, tap j of a chunk belongs to wave j mod 2
//     (five k-steps per chunk and wave, the last one of the odd wave a zero-weight dummy); partial sums exchanged through LDS at the end;
//   * BatchNorm sums of the stored values, one [2][64] slice of record `unit` per workgroup.
#include "common.h"
#include <stdlib.h>
#include "conv_vox.h"

int tri_internal_num_cus();                                                   // conv_igemm.hip

struct S2gArgs {
    const void* in;            // [N, H, W, Cin] 16-bit
    const void* w;             // packed operand, FRAGMENT-MAJOR (tri_weight_prep frag = 1): [Cout / 16][Kpad / 32][64 lanes][8], k = (ky * 3 + kx) * Cin + channel
    void* out;                 // [N, OH, OW, Cout]
    float* stats;              // [nunits][2][Cout] or NULL
    int N, H, W, OH, OW, Cin, Cout, Kpad;
    int ipu, nunits;           // images per unit, units
    int PW, PS, NSP, HB;       // plane pitch OW + 1, plane sites per image (OH + 1) * PW, sites per half plane of a unit (multiple of 32), its bytes
    unsigned in_bytes, w_bytes;
};

// B fragments are read by hand (asm) with counted waits: left to the compiler, the 12-fragment variant kept TWO fragment registers and waited
// lgkmcnt(1) in front of every MFMA pair - each pair sat out its own LDS read (25 us for layer3's opening layer, conv_dma_kernel: 21).
__device__ __forceinline__ v4i s2g_lds_read16(unsigned addr) {
    v4i r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
// wait until at most n LDS operations are outstanding (n is a constant after unrolling); f becomes "defined here", so the MFMA that consumes
// it cannot be scheduled above the wait
__device__ __forceinline__ void s2g_wait(v4i& f, const int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f)); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f)); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f)); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f)); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(f)); break;
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f)); break;
        case 7: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(f)); break;
        case 8: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f)); break;
        case 9: asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(f)); break;
        case 10: asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(f)); break;
        default: asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(f)); break;
    }
}

template <int N>
__device__ __forceinline__ float s2g_row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}


template <typename AT, int NRT>
__global__ __launch_bounds__(256, 1) void conv_s2g_kernel(const S2gArgs p) {
    typedef typename OpOf<AT>::E E;
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    constexpr int TN = 2, WC = 2, WK = 2, CT = WC * TN * 16, KPC = 5, U = 2, RING = U * KPC, MAXL = NRT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
    const int wc = wave / WK, wk = wave - wc * WK;
    char* const slab = smem;
    const int BUF = 8 * p.HB;

    // ---- workgroup -> (unit, channel tile): workgroups of one channel tile share an XCD (blockIdx % 8), so its weights stay in that L2
    const int nct = p.Cout / CT;
    int unit, ctile;
    {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        if (nct >= 8) { const int q = nct >> 3; ctile = xcd + 8 * (j % q); unit = j / q; }
        else { const int r = 8 / nct; ctile = xcd % nct; unit = xcd / nct + r * j; }
    }
    if (unit >= p.nunits) return;
    const int b0 = unit * p.ipu;
    const int ns = min(p.ipu, p.N - b0);                                      // images of this unit
    const int opi = p.OH * p.OW;                                              // output positions per image
    const int nrows = ns * opi;
    const int n0 = ctile * CT;
    const int nchunks = p.Cin >> 5;

    // ---- weight ring of the MFMA waves (see conv_voxg_kernel): slot (c mod U) * KPC + i holds k-step (chunk c, tap wk + i * WK)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    unsigned wrow[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) wrow[tn] = (unsigned)(((n0 >> 4) + wc * TN + tn) * (p.Kpad >> 5) * 1024 + lane * 16);
    v8 wf[RING][TN];
    auto load_slot = [&](const int slot, const int chunk, const int i) {
        const int tap = wk + i * WK;
        const bool ok = tap < 9 && chunk < nchunks;
        const unsigned koff = (unsigned)((tap * nchunks + chunk) * 1024);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            wf[slot][tn] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(wrs, ok ? wrow[tn] + koff : 0x80000000u, 0, 0));
    };
    auto load_ring = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < KPC; ++i) load_slot(u * KPC + i, u, i);
    };

    // ---- slab pieces of this thread: piece e = t + 256 u is the 16-byte quarter (e & 3) of input pixel (e >> 2) of the unit (raster order
    // over its images: coalesced).  Input pixel (iy, ix) = padded pixel (iy + 1, ix + 1) = site ((iy + 1) >> 1, (ix + 1) >> 1) of plane
    // ((iy + 1) & 1, (ix + 1) & 1); quarter q goes to half q >> 1, part q & 1.  Padding sites are never written: both buffers are zeroed once.
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const int ppi = p.H * p.W, npix = ns * ppi;
    // Piece u of a thread is pixel (t >> 2) + 64 u, quarter t & 3: the SAME quarter, 64 pixels further on - the same (iy, ix) of a later image
    // (maps of <= 64 pixels) or an even number of image rows further down (W divides 32: the parity plane does not change), so source and
    // destination of piece u are piece 0's plus wave-uniform steps: two registers instead of two arrays, no per-piece branch (the
    // geometry check admits only such maps).  Pieces past the unit's pixels (a short last unit) read out of range - zeros - and land on
    // site 0 of plane (0, 0), which is padding.
    const int px0 = t >> 2, q = t & 3;
    const int img0 = px0 / ppi, rem0 = px0 - img0 * ppi, iy0 = rem0 / p.W, ix0 = rem0 - iy0 * p.W;
    const int dst0 = ((((iy0 + 1) & 1) * 2 + ((ix0 + 1) & 1)) * 2 + (q >> 1)) * p.HB + (img0 * p.PS + ((iy0 + 1) >> 1) * p.PW + ((ix0 + 1) >> 1)) * 32 + (q & 1) * 16;
    const unsigned src0 = (unsigned)((((size_t)b0 * ppi + px0) * p.Cin + q * 8) * 2);
    const int img_step = ppi <= 64 ? 64 / ppi : 0;                            // images per 64 pixels (small maps)
    const int rows64 = ppi <= 64 ? 0 : 64 / p.W;                              // image rows per 64 pixels (an even number), maps of > 64 pixels
    const int rpi = ppi <= 64 ? 1 : p.H / rows64;                             // steps per image
    auto piece_dst = [&](const int u) -> int {                                // (u is an unrolled index: wave-uniform arithmetic)
        const int img = ppi <= 64 ? u * img_step : u / rpi, rr = ppi <= 64 ? 0 : (u - img * rpi) * (rows64 >> 1);
        return img * p.PS * 32 + rr * p.PW * 32;
    };
    uint4 pre[MAXL];
    auto slab_request = [&](int chunk) {
#pragma unroll
        for (int u = 0; u < MAXL; ++u) {
            const bool ok = px0 + 64 * u < npix;
            pre[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                irs, ok ? src0 + (unsigned)(u * 64 * p.Cin * 2 + chunk * 64) : 0x80000000u, 0, 0));
        }
    };
    auto slab_commit = [&](int buf) {
        char* const base = slab + buf * BUF;
#pragma unroll
        for (int u = 0; u < MAXL; ++u) {
            const bool ok = px0 + 64 * u < npix;
            *(uint4*)(base + (ok ? dst0 + piece_dst(u) : (q & 1) * 16)) = pre[u];
        }
    };
    const unsigned lds0 = lds_addr(slab);
    slab_request(0);
    load_ring();                                                              // (behind the first chunk's pixels: the first MFMA needs those, and ring slot 0)
    for (int i = t * 16; i < 2 * BUF; i += 256 * 16) *(uint4*)(slab + i) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    slab_commit(0);
    __syncthreads();

    // ---- rows: output position r of the unit (image r / opi, raster order inside it) -> slab byte offset of its tap (0, 0) site
    const int lane_part = (fq >> 1) * p.HB + (fq & 1) * 16;                   // half (fq >> 1), 16-byte part (fq & 1) of a site's 32 channels
    int lbase[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        int r = rt * 16 + fr;
        if (r >= nrows) r = 0;                                                // rows past the end multiply the first row again: discarded
        const int img = r / opi, rem = r - img * opi, oy = rem / p.OW, ox = rem - oy * p.OW;
        lbase[rt] = (img * p.PS + oy * p.PW + ox) * 32 + lane_part;
    }
    // slab offset of tap wk + i * WK: plane (ky & 1, kx & 1) = two half planes further per plane, site (ky >> 1) * PW + (kx >> 1) further on
    auto tap_off = [&](const int i) -> int {
        const int tap = wk + i * WK, ky = tap / 3, kx = tap - ky * 3;
        return tap < 9 ? ((ky & 1) * 2 + (kx & 1)) * 2 * p.HB + ((ky >> 1) * p.PW + (kx >> 1)) * 32 : 0;
    };

    f32x4 acc[NRT][TN];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[rt][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int c0 = 0; c0 < nchunks; c0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u;                                             // (nchunks is even: Cin % 64 == 0)
            const unsigned sb = lds0 + (c & 1) * BUF;
#ifndef S2G_ABL_NOSLAB
            if (c + 1 < nchunks) slab_request(c + 1);
#endif
            v4i bf[NRT];
            {
                const int toff = tap_off(0);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) bf[rt] = s2g_lds_read16(sb + lbase[rt] + toff);
            }
#pragma unroll
            for (int i = 0; i < KPC; ++i) {
                const int toff = i + 1 < KPC ? tap_off(i + 1) : 0;
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    s2g_wait(bf[rt], i + 1 < KPC ? NRT - 1 : NRT - 1 - rt);   // the reads issued after this fragment's may still be in flight
                    const v8 b = __builtin_bit_cast(v8, bf[rt]);
#ifndef S2G_ABL_NOMMA
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[rt][tn] = MM::mma(wf[u * KPC + i][tn], b, acc[rt][tn]);
#else
                    acc[rt][0][0] += (float)b[0];
#endif
#ifndef S2G_ABL_NOREAD
                    if (i + 1 < KPC) bf[rt] = s2g_lds_read16(sb + lbase[rt] + toff);
#endif
                }
#ifndef S2G_ABL_NORING
                load_slot(u * KPC + i, c + U, i);                             // the slot's next tenant: the same tap of chunk c + U
#endif
            }
            if (c + 1 < nchunks) {
#ifndef S2G_ABL_NOSLAB
                slab_commit((c + 1) & 1);
#endif                                     // (that buffer was last read in chunk c - 1: every wave is past it)
                __syncthreads();
            }
        }
    }

    // ---- partial sums of the two tap shares: tile (rt, tn) of channel pair wc is finished by the wave whose wk = tile index mod 2; the other
    // wave of the pair hands it its partial sums through LDS (the slab is idle now)
    __syncthreads();
    f32x4* const red = (f32x4*)slab;                                          // [wave][rt][tn][64 lanes]
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            if (((rt * TN + tn) % WK) != wk) red[((wave * NRT + rt) * TN + tn) * 64 + lane] = acc[rt][tn];
    __syncthreads();
    float cs[TN][4], cq[TN][4];                                               // BatchNorm sums of this wave's epilogue share
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) { cs[tn][r] = 0.f; cq[tn][r] = 0.f; }
    {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if (((rt * TN + tn) % WK) != wk) continue;
                const f32x4 other = red[(((wc * WK + (wk ^ 1)) * NRT + rt) * TN + tn) * 64 + lane];
                const f32x4 v = wk == 0 ? acc[rt][tn] + other : other + acc[rt][tn];       // (wave order: the sum does not depend on who adds)
                const int r = rt * 16 + fr;
                if (r < nrows) {
                    typedef E e4 __attribute__((ext_vector_type(4)));
                    const e4 h = __builtin_convertvector(v, e4);
                    *(e4*)((AT*)p.out + ((size_t)b0 * opi + r) * p.Cout + n0 + (wc * TN + tn) * 16 + fq * 4) = h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float f = (float)h[q]; cs[tn][q] += f; cq[tn][q] += f * f; }
                }
            }
    }
    // ---- BatchNorm sums: over the 16 rows of a fragment (DPP), then over the two waves that finished tiles of the same channels
    if (p.stats) {
        __syncthreads();
        float* const sred = (float*)slab;                                     // [wave][TN][16 channels][2]
        {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float s_ = cs[tn][q], q_ = cq[tn][q];
                    s_ += s2g_row_ror<8>(s_); q_ += s2g_row_ror<8>(q_);
                    s_ += s2g_row_ror<4>(s_); q_ += s2g_row_ror<4>(q_);
                    s_ += s2g_row_ror<2>(s_); q_ += s2g_row_ror<2>(q_);
                    s_ += s2g_row_ror<1>(s_); q_ += s2g_row_ror<1>(q_);
                    if (fr == 0) {
                        sred[((wave * TN + tn) * 16 + fq * 4 + q) * 2 + 0] = s_;
                        sred[((wave * TN + tn) * 16 + fq * 4 + q) * 2 + 1] = q_;
                    }
                }
        }
        __syncthreads();
        if (t < CT) {
            const int cw = t / (TN * 16), rem = t - cw * TN * 16;             // channel pair (wc), (tn, channel) inside it
            float s_ = 0.f, q_ = 0.f;
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                s_ += sred[(((cw * WK + w) * TN) * 16 + rem) * 2 + 0];
                q_ += sred[(((cw * WK + w) * TN) * 16 + rem) * 2 + 1];
            }
            p.stats[((size_t)unit * 2 + 0) * p.Cout + n0 + t] = s_;
            p.stats[((size_t)unit * 2 + 1) * p.Cout + n0 + t] = q_;
        }
    }
}

// ------------------------------------------------------------------------------------------------ plan + launch
static bool s2g_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_S2G"); v = (e && e[0] == '1') ? 1 : 0; }       // A/B switch: these layers stay on conv_dma_kernel
    return v == 1;
}

bool tri_internal_s2g_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                               int pd, int ph, int pw, TriS2gGeom* g) {
    if (s2g_disabled()) return false;
    if (ID != 1 || OD != 1 || KD != 1 || KH != 3 || KW != 3 || stride != 2 || pd != 0 || ph != 1 || pw != 1) return false;
    if (IH % 2 || IW % 2 || OH != IH / 2 || OW != IW / 2) return false;
    if (cin % 64 != 0 || cin < 128 || cout % 64 != 0) return false;           // (64-channel inputs: layer2's opening layer has kernels of its own)
    const int opi = OH * OW, ppi = IH * IW;
    if (opi > 192 || opi < 4) return false;
    if (ppi <= 64 ? 64 % ppi != 0 : (32 % IW != 0 || ppi % 64 != 0)) return false;      // the slab pieces of a thread step uniformly (see the kernel)
    if ((long)B * IH * IW * cin * 2 >= (1L << 31) || (long)cout * 9 * cin * 2 >= (1L << 31)) return false;      // 32-bit buffer offsets
    const int cus = tri_internal_num_cus(), nct = cout / 64;
    if (nct >= 8 ? nct % 8 : 8 % nct) return false;
    // Measured at the bench shape (profiles/r5/NOTES_voxel.md): layer4's opening layer (8x8 maps, units of 6 images = 96 rows) 27 -> 20 us against
    // conv_dma_kernel; layer3's (16x16 maps, units of 3 images = 192 rows, a 12-fragment variant of this kernel) 25 against 21 us - a chunk's
    // MFMAs (0.8 us) are shorter than the latency of the next chunk's slab pieces, which one chunk of look-ahead cannot hide.  The plan therefore
    // takes maps of at most 16 output positions and units of at most 96 rows; the 12-fragment instantiation was dropped in round 6.
    if (opi > 16) return false;
    // images per unit: up to 96 rows, halved while the launch would leave CUs without a workgroup
    int ipu = 96 / opi;
    if (ipu < 1) return false;
    if (ipu > B) ipu = B;
    while (ipu > 1 && (long)((B + ipu - 1) / ipu) * nct < cus) ipu = (ipu + 1) / 2;
    const int PW = OW + 1, PS = (OH + 1) * PW;
    const int NSP = (ipu * PS + 31) / 32 * 32;
    g->ipu = ipu; g->nunits = (B + ipu - 1) / ipu; g->nrt = 6;
    g->PW = PW; g->PS = PS; g->NSP = NSP; g->HB = NSP * 32 + 128;          // (+128: the two halves of a site - lanes fq 0,1 / 2,3 of a fragment read - land on different banks)
    g->grid = nct >= 8 ? g->nunits * nct : 8 * ((g->nunits + 8 / nct - 1) / (8 / nct));
    size_t smem = (size_t)2 * 8 * g->HB;
    const size_t need = (size_t)4 * g->nrt * 2 * 64 * sizeof(f32x4);          // epilogue scratch (aliases the slab)
    if (need > smem) smem = need;
    g->smem = (int)smem;
    return smem <= 160 * 1024;
}

template <typename AT, int NRT>
static int s2g_launch_t(const S2gArgs& a, const TriS2gGeom& g, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv_s2g_kernel<AT, NRT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    conv_s2g_kernel<AT, NRT><<<g.grid, 256, g.smem, stream>>>(a);
    return tri_check_launch("tri_conv(s2g)");
}

int tri_internal_s2g_launch(const TriS2gGeom& g, int B, int IH, int IW, int cin, int cout, int kpad, const void* in, const void* w, void* out,
                            float* stats, int act_fmt, hipStream_t stream) {
    S2gArgs a{};
    a.in = in; a.w = w; a.out = out; a.stats = stats;
    a.N = B; a.H = IH; a.W = IW; a.OH = IH / 2; a.OW = IW / 2; a.Cin = cin; a.Cout = cout; a.Kpad = kpad;
    a.ipu = g.ipu; a.nunits = g.nunits; a.PW = g.PW; a.PS = g.PS; a.NSP = g.NSP; a.HB = g.HB;
    a.in_bytes = (unsigned)((size_t)B * IH * IW * cin * 2);
    a.w_bytes = (unsigned)((size_t)cout * kpad * 2);
    return act_fmt == TRI_FMT_F16 ? s2g_launch_t<f16_t, 6>(a, g, stream) : s2g_launch_t<bf16_t, 6>(a, g, stream);
}
