// conv_voxg_kernel: the SubMConv3d layers of the COARSE voxel grids (2^3, 4^3, 8^3 sites per sample: levels 2-4 of the 32^3 tower,
// levels 3-4 of the 64^3 tower; sparse_cnn.py:22-32) and their data gradients, 16-bit storage, gfx950.
//
// Through conv_dma_kernel these layers are im2col GEMMs over a compact row list: every site row crosses the L2 -> LDS path once per
// tap (27 x), every 128 x 64 tile re-streams its weight panel through LDS, and the small ones (32^3 inputs: 245 .. 4,054 rows) run
// split-K with a finish launch: 10 + 5 us per level at the bench shape, 47 / 50 us (324 / 270 TFLOP/s) for levels 3 / 4 of 64^3 x 64
// - both bound by the bytes a CU can pull from L2 (~29 B/clk), at 43 FLOP per byte.  Here the ACTIVATIONS are stationary in LDS and
// the WEIGHTS go straight from L2 into MFMA registers:
//   * a workgroup owns a UNIT - `spu` whole samples of the dense grid (one 8^3 sample, 2-4 4^3 samples, 4-32 2^3 samples) - and CT
//     output channels.  It reads the unit's site mask, ranks the active sites (ballot / popcount: no row list from outside) and
//     stages the unit's sites, 32 input channels at a time, in a zero-PADDED slab: padded coordinate (z, y, x) of a (D+2) x (D+2) x
//     (D+1) box per sample (row pitch D + 1: the right halo of a row is the left halo of the next), so the neighbour of a site under
//     tap (kd, kh, kw) is the site ((kd * (D+2)) + kh) * (D+1) + kw further on - a CONSTANT, whatever the site: no validity masks,
//     no coordinate arithmetic in the MFMA loop;
//   * slab layout: the 64 bytes of a site's 32-channel chunk are split into two 32-byte halves kept in two planes
//     ([half][site][32 B]).  Lane (fr, fq) of a 16-row B fragment reads 16 bytes at  half(fq >> 1) + site(fr) * 32 + (fq & 1) * 16:
//     for 16 consecutive sites the 16 lanes of every ds_read_b128 service group land on 16 distinct 16-byte bank slots
//     ((2 s + (fq & 1)) mod 16) - conflict-free WITHOUT a swizzle, so a tap is an address offset and nothing else;
//   * the MFMA rows are the ACTIVE sites only (16 per tile, in raster order: mostly x-consecutive runs) - no padded or inactive row is
//     multiplied (the round-3 slab attempt for these levels multiplied every padded site of a run and lost: NOTES_vox.md);
//   * the four waves split the workgroup's (output-channel tile, k-step) space: WC waves along the channels (TN tiles of 16 each),
//     WK waves along the taps (tap j of a chunk belongs to wave j mod WK; partial sums are added through LDS at the end).  Every
//     weight fragment is needed by exactly ONE wave, so it is loaded from L2 straight into that wave's registers as an MFMA A
//     fragment (the packed operand is stored FRAGMENT-MAJOR, so a wave instruction reads 1 KiB contiguously: with row-major rows a
//     fragment was 16 rows x 64 B and every load took the texture addresser ~150 cycles) through a RING of 14-28 register fragments per wave that is refilled slot by slot: no
//     LDS traffic, no barrier per k-step - one barrier per 32-channel chunk (the slab is double-buffered: the next chunk's sites are
//     requested before the current chunk's MFMAs and written to the other buffer after them);
//   * units with more than 16 * NRT active sites run in passes (weights re-streamed per pass);
//   * epilogue: partial sums of the WK waves added in LDS, rounded to the storage type, stored as 8-byte pieces (rows of inactive
//     sites are never written), BatchNorm sums of the stored values: one [2][CT] slice of record `unit` per workgroup.
// Bytes a CU pulls per FLOP: weights CT x K once per unit, slab once per (unit, channel tile): 120-250 FLOP per byte.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "conv_vox.h"

int tri_internal_num_cus();                                                   // conv_igemm.hip

struct VoxgArgs {
    const void* in;            // [B, D, D, D, Cin] 16-bit (values at inactive sites are ignored)
    const void* w;             // packed operand, FRAGMENT-MAJOR (tri_weight_prep frag = 1): [Cout / 16][Kpad / 32][64 lanes][8], k = tap * Cin + channel
    void* out;                 // [B, D, D, D, Cout]; rows of inactive sites are not written
    const uint8_t* mask;       // [B * D^3] site mask, or NULL (every site active)
    float* stats;              // [nunits][2][Cout] or NULL
    int B, D, logD, Cin, Cout, Kpad;
    int spu, nunits;           // samples per unit, units
    int mirror;                // 1: data gradient (tap (kd, kh, kw) reads the site at -(kd-1, kh-1, kw-1))
    unsigned in_bytes, w_bytes;
#ifdef VOXG_PROBE
    long long* dbg;            // [grid][16] stamps of wave 0 (tools/probes/voxg_stamps.py)
#endif
};
#ifdef VOXG_PROBE
#define VOXG_STAMP(i) do { if (p.dbg && t == 0) p.dbg[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define VOXG_STAMPV(i, v) do { if (p.dbg && t == 0) p.dbg[(size_t)blockIdx.x * 16 + (i)] = (long long)(v); } while (0)
#else
#define VOXG_STAMP(i)
#define VOXG_STAMPV(i, v)
#endif

template <int N>
__device__ __forceinline__ float voxg_row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}

#define VOXG_MAXROWS 512                                                      // active sites of a unit (8^3 sample, all active)

// LDS: row tables + unit mask (VOXG_TAB bytes) | [2 slab buffers][2 halves][NS sites x 32 B]; the epilogue scratch aliases the slab
#define VOXG_TAB (2 * VOXG_MAXROWS * 4 + 16 * 4 + 512 + 192)                  // 4,864: a multiple of 256
struct VoxgGeom {
    int P, RY, PZ, sample_sites, NS, half_bytes, buf_bytes;
};
__host__ __device__ inline VoxgGeom voxg_geom(int D, int spu) {
    VoxgGeom g;
    g.P = D + 1; g.RY = D + 2; g.PZ = D + 2;
    g.sample_sites = g.PZ * g.RY * g.P;
    g.NS = spu * g.sample_sites + g.P + 2;                                    // (+ the far corner of the last sample's last site)
    g.half_bytes = (g.NS * 32 + 255) / 256 * 256 + 128;                       // 128 mod 256: the two halves of a site land on different banks when written
    g.buf_bytes = 2 * g.half_bytes;
    return g;
}

template <typename AT, int TN, int WC, int WK, int NRT, int RING>
__global__ __launch_bounds__(256, 1) void conv_voxg_kernel(const VoxgArgs p) {
    typedef typename OpOf<AT>::E E;
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    static_assert(WC * WK == 4, "four waves");
    constexpr int CT = WC * TN * 16;                                          // output channels of a workgroup
    // 16-byte slab pieces per thread and chunk: four per ACTIVE site, up to 4 per thread (256 rows) held in registers between their request
    // (at the start of the previous chunk) and their LDS write (at its end); the rows past 256 of a very dense unit are copied in a plain
    // loop at the chunk boundary.  The variants with NRT < 12 run units of at most 16 * NRT sites.
    constexpr int MAXL = NRT == 12 ? 4 : (NRT * 16 * 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
    const int wc = wave / WK, wk = wave - wc * WK;
    const int D = p.D, lD = p.logD, D3 = 1 << (3 * lD);
    const VoxgGeom G = voxg_geom(D, p.spu);
    char* const slab = smem + VOXG_TAB;
    int* const row_lds = (int*)smem;                                          // [VOXG_MAXROWS] byte offset (site * 32) of the row's (-1,-1,-1) corner
    int* const row_glob = row_lds + VOXG_MAXROWS;                             // [VOXG_MAXROWS] global site index
    int* const wcnt = row_glob + VOXG_MAXROWS;                                // [8] per-(half, wave) counts of the ranking

    // ---- workgroup -> (unit, channel tile): workgroups of one channel tile share an XCD (blockIdx % 8), so its weights stay in that L2
    const int nct = p.Cout / CT;
    int unit, ctile;
    {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        if (nct >= 8) { const int q = nct >> 3; ctile = xcd + 8 * (j % q); unit = j / q; }
        else { const int r = 8 / nct; ctile = xcd % nct; unit = xcd / nct + r * j; }
    }
    if (unit >= p.nunits) return;
    const int b0 = unit * p.spu;
    const int ns = min(p.spu, p.B - b0);                                      // samples of this unit
    const int nsites = ns << (3 * lD);                                        // dense sites of this unit (<= 512)
    const int n0 = ctile * CT;
    VOXG_STAMP(0);

    // ---- site mask of the unit: two bytes per thread, requested first (everything below waits for it)
    uint8_t m0 = 0, m1 = 0;
    if (t < nsites) m0 = p.mask ? p.mask[((size_t)b0 << (3 * lD)) + t] : 1;
    if (t + 256 < nsites) m1 = p.mask ? p.mask[((size_t)b0 << (3 * lD)) + t + 256] : 1;

    // ---- weight fragments: wave (wc, wk) owns output channels n0 + (wc * TN + tn) * 16 + fr and the taps wk, wk + WK, ...
    // A wave's k-steps run chunk-major: (chunk c, i) = tap wk + i * WK of input channels 32 c .. 32 c + 31, i < KPC (the last one of a chunk
    // is a dummy for the waves that own one tap fewer).  Their A fragments live in a RING of registers: slot (c mod U) * KPC + i, reloaded
    // with k-step (c + U, i) right behind the MFMAs that consumed it - RING fragments (x TN) stay in flight per wave, requested from the
    // first instruction of the kernel on (the loop is unrolled over U chunks so that every slot index is a compile-time constant).
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    unsigned wrow[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)                                            // fragment (channel tile, k-step ks) = 1 KiB at (tile * Kpad / 32 + ks) * 1024: lane * 16 inside
        wrow[tn] = (unsigned)(((n0 >> 4) + wc * TN + tn) * (p.Kpad >> 5) * 1024 + lane * 16);
    const int nchunks = p.Cin >> 5;
    constexpr int KPC = (27 + WK - 1) / WK;                                   // k-steps per chunk and wave: 7 (WK = 4) or 14 (WK = 2)
    constexpr int U = RING / KPC;                                             // chunks per unrolled loop body
    static_assert(U * KPC == RING && U >= 1, "ring = whole chunks");
    v8 wf[RING][TN];
    auto load_slot = [&](const int slot, const int chunk, const int i) {
        const int tap = wk + i * WK;
        const bool ok = tap < 27 && chunk < nchunks;
        const unsigned koff = (unsigned)((tap * nchunks + chunk) * 1024);   // k-step index = (tap * Cin + 32 chunk) / 32
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
    // ---- zero fill of the slab buffers: only ACTIVE sites are ever written afterwards, so the padding and the inactive sites stay zero for
    // every chunk (and whatever the tensor holds at inactive sites is never read).  Buffer 0 is cleared while the mask is on its way,
    // buffer 1 while the first chunk's sites are (start-up order - stamps: with the weight ring requested first, its 112 KB occupied
    // the texture path for 2 k cycles in front of the mask-dependent chain, and the first MFMA issued 10 k cycles into the kernel)
    auto zero_fill = [&](int first, int count) {
        for (int i = first * G.buf_bytes + t * 16; i < (first + count) * G.buf_bytes; i += 256 * 16) *(uint4*)(slab + i) = make_uint4(0u, 0u, 0u, 0u);
    };
    VOXG_STAMP(8);
    zero_fill(0, 1);
    VOXG_STAMP(10);

    // ---- rank of every active site in raster order (first half: sites 0..255, second half: 256..511) -> row tables
    {
        const unsigned long long b0m = __ballot(m0 != 0), b1m = __ballot(m1 != 0);
        VOXG_STAMP(11);
        if (lane == 0) { wcnt[wave] = __popcll(b0m); wcnt[4 + wave] = __popcll(b1m); }
        __syncthreads();
        VOXG_STAMP(12);
        int base0 = 0, base1 = 0, tot0 = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) { base0 += wcnt[w]; base1 += wcnt[4 + w]; }
            tot0 += wcnt[w];
        }
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = t + 256 * h;
            if (h ? m1 : m0) {
                const int rk = h ? tot0 + base1 + __popcll(b1m & below) : base0 + __popcll(b0m & below);
                const int bl = s >> (3 * lD), r = s & (D3 - 1);
                const int z = r >> (2 * lD), y = (r >> lD) & (D - 1), x = r & (D - 1);
                row_lds[rk] = (((bl * G.PZ + z) * G.RY + y) * G.P + x) * 32;  // the (-1, -1, -1) corner of the site's neighbourhood
                row_glob[rk] = (b0 << (3 * lD)) + s;
            }
        }
    }
    __syncthreads();                                                          // zero fill + tables visible
    int nrows = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) nrows += wcnt[w];
    VOXG_STAMP(1);
    VOXG_STAMPV(6, nrows);
    VOXG_STAMPV(7, __builtin_amdgcn_s_memrealtime());
    if (nrows == 0) {                                                         // nothing active in this unit (the same answer in every wave)
        if (p.stats && t < 2 * CT) p.stats[((size_t)unit * 2 + t / CT) * p.Cout + n0 + t % CT] = 0.f;
        return;
    }

    // ---- slab pieces of this thread: piece e = t + 256 u is the 16-byte quarter (e & 3) of active row (e >> 2)
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const int centre = (G.RY * G.P + G.P + 1) * 32;
    unsigned src[MAXL];
    int dst[MAXL];
#pragma unroll
    for (int u = 0; u < MAXL; ++u) {
        const int e = t + 256 * u, r = e >> 2, q = e & 3;
        const bool ok = r < nrows;
        dst[u] = ok ? row_lds[r] + centre + (q >> 1) * G.half_bytes + (q & 1) * 16 : -1;
        src[u] = ok ? (unsigned)(((size_t)row_glob[r] * p.Cin + q * 8) * 2) : 0x80000000u;
    }
    uint4 pre[MAXL];
    auto slab_request = [&](int chunk) {
#pragma unroll
        for (int u = 0; u < MAXL; ++u)
            if (dst[u] >= 0) pre[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(irs, src[u] + chunk * 64, 0, 0));
    };
    auto slab_commit = [&](int buf, int chunk) {
        char* const base = slab + buf * G.buf_bytes;
#pragma unroll
        for (int u = 0; u < MAXL; ++u)
            if (dst[u] >= 0) *(uint4*)(base + dst[u]) = pre[u];
        if (NRT == 12) {                                                      // rows 256 .. of a dense unit (rare): load and write at once
#pragma unroll 1
            for (int e = t + 256 * MAXL; e < 4 * nrows; e += 256) {
                const int r = e >> 2, q = e & 3;
                const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                    irs, (unsigned)(((size_t)row_glob[r] * p.Cin + q * 8) * 2) + chunk * 64, 0, 0));
                *(uint4*)(base + row_lds[r] + centre + (q >> 1) * G.half_bytes + (q & 1) * 16) = v;
            }
        }
    };
    slab_request(0);
    VOXG_STAMP(9);
    load_ring();                                                              // (behind the first chunk's sites: the first MFMA needs those, and ring slot 0)
    zero_fill(1, 1);

    const int lane_part = (fq >> 1) * G.half_bytes + (fq & 1) * 16;
    float cs[TN][4], cq[TN][4];                                               // BatchNorm sums of this wave's epilogue share
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) { cs[tn][r] = 0.f; cq[tn][r] = 0.f; }
    // slab offset of tap wk + i * WK (the mirrored tap for the data gradient); i is an unrolled loop index, so these are scalar values
    // (a wave that owns one tap fewer runs a DUMMY last k-step of every chunk - zero weights, tap offset 0 - so that the loop is branch-free)
    auto tap_off = [&](const int i) -> int {
        const int tap = wk + i * WK;
        int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        if (p.mirror) { kd = 2 - kd; kh = 2 - kh; kw = 2 - kw; }
        return tap < 27 ? ((kd * G.RY + kh) * G.P + kw) * 32 : 0;
    };

    const int npass = (nrows + 16 * NRT - 1) / (16 * NRT);
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        const int r0 = pass * 16 * NRT;
        const int nrt = min(NRT, (nrows - r0 + 15) >> 4);
        int lbase[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const int r = r0 + rt * 16 + fr;
            lbase[rt] = (r < nrows ? row_lds[r] : 0) + lane_part;             // rows past the end read the first sites of the slab: discarded
        }
        f32x4 acc[NRT][TN];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[rt][tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (pass > 0) {
            // the previous pass's epilogue scratch lay over the slab: once every wave is done with it, everything is zeroed again
            slab_request(0);
            load_ring();
            __syncthreads();
            zero_fill(0, 2);
            __syncthreads();
        }
        slab_commit(0, 0);
        __syncthreads();

        VOXG_STAMP(2);
        // The chunk loop, for a COMPILE-TIME number NT of row tiles: a run-time guard per tile made every tile a basic block, and at every
        // join the compiler waited lgkmcnt(0) - each MFMA pair then waited out the LDS read issued just before it (stamps: 3.6 x the
        // MFMA time).  NT is the pass's tile count rounded up to 2 / 4 / 6 / 8 / 10 / 12; the surplus tiles multiply rows past the end of the
        // list (discarded), the loop body is branch-free and its fragment reads run one k-step ahead of the MFMAs.
        auto chunk_loop = [&](auto NTc) {
            constexpr int NT = decltype(NTc)::value < NRT ? decltype(NTc)::value : NRT;
#pragma unroll 1
            for (int c0 = 0; c0 < nchunks; c0 += U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = c0 + u;
                    if (c < nchunks) {                                        // (U = 4 with two chunks: the same answer in every wave)
                        const char* const sb = slab + (c & 1) * G.buf_bytes;
                        if (c + 1 < nchunks) slab_request(c + 1);
                        v8 bf[NT];
                        {
                            const int toff = tap_off(0);
#pragma unroll
                            for (int rt = 0; rt < NT; ++rt) bf[rt] = *(const v8*)(sb + lbase[rt] + toff);
                        }
#pragma unroll
                        for (int i = 0; i < KPC; ++i) {
                            const int toff = i + 1 < KPC ? tap_off(i + 1) : 0;
#pragma unroll
                            for (int rt = 0; rt < NT; ++rt) {
#pragma unroll
                                for (int tn = 0; tn < TN; ++tn) acc[rt][tn] = MM::mma(wf[u * KPC + i][tn], bf[rt], acc[rt][tn]);
                                if (i + 1 < KPC) bf[rt] = *(const v8*)(sb + lbase[rt] + toff);
                            }
                            load_slot(u * KPC + i, c + U, i);                 // the slot's next tenant: the same tap of chunk c + U
                        }
                        if (c + 1 < nchunks) {
                            slab_commit((c + 1) & 1, c + 1);                  // (that buffer was last read in chunk c - 1: every wave is past it)
                            __syncthreads();
                        }
                    }
                }
            }
        };
        if (NRT <= 2 || nrt <= 2) chunk_loop(std::integral_constant<int, 2>{});
        else if (NRT <= 4 || nrt <= 4) chunk_loop(std::integral_constant<int, 4>{});
        else if (nrt <= 6) chunk_loop(std::integral_constant<int, 6>{});
        else if (NRT <= 8 || nrt <= 8) chunk_loop(std::integral_constant<int, 8>{});
        else if (nrt <= 10) chunk_loop(std::integral_constant<int, 10>{});
        else chunk_loop(std::integral_constant<int, 12>{});
        VOXG_STAMP(3);

        // ---- partial sums of the WK tap shares: tile (rt, tn) of channel column wc is finished by the wave whose wk = tile index mod WK;
        // the other waves of the column hand it their partial sums through LDS (the slab is idle now), added in wave order
        __syncthreads();
        f32x4* const red = (f32x4*)slab;                                      // [wave][rt][tn][64 lanes]
        if (WK > 1) {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
                if (rt < nrt) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        if (((rt * TN + tn) % WK) != wk) red[((wave * NRT + rt) * TN + tn) * 64 + lane] = acc[rt][tn];
                }
            __syncthreads();
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if (rt >= nrt || (WK > 1 && ((rt * TN + tn) % WK) != wk)) continue;
                f32x4 v = acc[rt][tn];
                if (WK > 1) {
#pragma unroll
                    for (int w = 0; w < WK; ++w) {
                        const f32x4 pw = w == wk ? acc[rt][tn] : red[(((wc * WK + w) * NRT + rt) * TN + tn) * 64 + lane];
                        v = w == 0 ? pw : v + pw;
                    }
                }
                const int r = r0 + rt * 16 + fr;
                if (r < nrows) {
                    typedef E e4 __attribute__((ext_vector_type(4)));
                    const e4 h = __builtin_convertvector(v, e4);
                    *(e4*)((AT*)p.out + (size_t)row_glob[r] * p.Cout + n0 + (wc * TN + tn) * 16 + fq * 4) = h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float f = (float)h[q]; cs[tn][q] += f; cq[tn][q] += f * f; }
                }
            }
    }

    VOXG_STAMP(4);
    // ---- BatchNorm sums: over the 16 sites of a fragment row (DPP), then over the waves that finished tiles of the same channels
    if (p.stats) {
        __syncthreads();
        float* const sred = (float*)slab;                                     // [wave][TN][16 channels][2]
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s_ = cs[tn][q], q_ = cq[tn][q];
                s_ += voxg_row_ror<8>(s_); q_ += voxg_row_ror<8>(q_);
                s_ += voxg_row_ror<4>(s_); q_ += voxg_row_ror<4>(q_);
                s_ += voxg_row_ror<2>(s_); q_ += voxg_row_ror<2>(q_);
                s_ += voxg_row_ror<1>(s_); q_ += voxg_row_ror<1>(q_);
                if (fr == 0) {
                    sred[((wave * TN + tn) * 16 + fq * 4 + q) * 2 + 0] = s_;
                    sred[((wave * TN + tn) * 16 + fq * 4 + q) * 2 + 1] = q_;
                }
            }
        __syncthreads();
        if (t < CT) {
            const int cw = t / (TN * 16), rem = t - cw * TN * 16;             // channel column (wc), (tn, channel) inside it
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
    VOXG_STAMP(5);
}

// ------------------------------------------------------------------------------------------------ plan + launch
static bool voxg_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_VOXG"); v = (e && e[0] == '1') ? 1 : 0; }      // A/B switch: these levels stay on conv_dma_kernel
    return v == 1;
}

bool tri_internal_voxg_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                                int pd, int ph, int pw, TriVoxgGeom* g) {
    if (voxg_disabled()) return false;
    const int D = ID;
    if (IH != D || IW != D || OD != D || OH != D || OW != D || (D != 2 && D != 4 && D != 8)) return false;
    if (KD != 3 || KH != 3 || KW != 3 || stride != 1 || pd != 1 || ph != 1 || pw != 1) return false;
    if (cin % 64 != 0 || cout % 16 != 0) return false;                        // (two 32-channel chunks per loop iteration)
    if ((long)B * D * D * D * cin * 2 >= (1L << 31) || (long)cout * 27 * cin * 2 >= (1L << 31)) return false;   // 32-bit buffer offsets
    const int D3 = D * D * D;
    // samples per unit: about 128 active sites at the occupancies these grids have (25 % at 8^3, 45 % at 4^3, 95 % at 2^3), at most 512
    // dense sites, halved while the launch would leave CUs without a workgroup of 16 output channels
    const double occ = D == 8 ? 0.25 : (D == 4 ? 0.45 : 0.95);
    int spu = 1;
    while (2 * spu * D3 <= 512 && occ * 2 * spu * D3 <= 128.0 && 2 * spu <= B) spu *= 2;
    const int cus = tri_internal_num_cus();
    while (spu > 1 && (long)((B + spu - 1) / spu) * (cout / 16) < cus) spu /= 2;
    const int nunits = (B + spu - 1) / spu;
    // output channels per workgroup: as many as keep one workgroup per CU (fewer weight re-reads of the slab, fewer partial-sum waves)
    int ct = 16;
    while (ct < 64 && cout % (2 * ct) == 0 && (long)nunits * (cout / (2 * ct)) >= cus) ct *= 2;
    static int force_ct = -1, force_spu = -1;                                 // tuning aids
    if (force_ct < 0) { const char* e = getenv("TRICOLO_VOXG_CT"); force_ct = e ? atoi(e) : 0; }
    if (force_spu < 0) { const char* e = getenv("TRICOLO_VOXG_SPU"); force_spu = e ? atoi(e) : 0; }
    if (force_ct == 16 || force_ct == 32 || force_ct == 64) { if (cout % force_ct == 0) ct = force_ct; }
    if (force_spu > 0 && force_spu * D3 <= 512) { spu = force_spu; }
    g->D = D; g->spu = spu; g->nunits = (B + spu - 1) / spu; g->ct = ct;
    const int nct = cout / ct;
    if (nct >= 8) { if (nct % 8) return false; g->grid = g->nunits * nct; }
    else { if (8 % nct) return false; const int r = 8 / nct; g->grid = 8 * ((g->nunits + r - 1) / r); }
    const VoxgGeom vg = voxg_geom(D, spu);
    g->smem = VOXG_TAB + 2 * vg.buf_bytes;
    if (g->smem > 160 * 1024) return false;
    return true;
}

template <typename AT, int TN, int WC, int WK, int NRT, int RING>
static int voxg_launch_t(const VoxgArgs& a, const TriVoxgGeom& g, hipStream_t stream) {
    // the epilogue scratch (partial sums of every wave, statistics) aliases the slab buffers
    const size_t need = (size_t)4 * NRT * TN * 64 * sizeof(f32x4);
    const VoxgGeom vg = voxg_geom(g.D, g.spu);
    size_t smem = g.smem;
    if (need > (size_t)2 * vg.buf_bytes) smem += need - 2 * vg.buf_bytes;     // (tiny slabs: 2^3 grids)
    if (smem > 160 * 1024) { tri_set_error("conv(voxg): LDS budget exceeded"); return TRI_ERR_UNSUPPORTED; }
    static size_t attr = 0;
    if (smem > attr) {
        hipFuncSetAttribute((const void*)conv_voxg_kernel<AT, TN, WC, WK, NRT, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = 160 * 1024;
    }
    conv_voxg_kernel<AT, TN, WC, WK, NRT, RING><<<g.grid, 256, smem, stream>>>(a);
    return tri_check_launch("tri_conv(voxg)");
}

int tri_internal_voxg_launch(const TriVoxgGeom& g, int B, int cin, int cout, int kpad, const void* in, const void* w, void* out, const uint8_t* mask,
                             float* stats, int transposed, int act_fmt, hipStream_t stream) {
    VoxgArgs a{};
    a.in = in; a.w = w; a.out = out; a.mask = mask; a.stats = stats;
    a.B = B; a.D = g.D; a.logD = g.D == 8 ? 3 : (g.D == 4 ? 2 : 1); a.Cin = cin; a.Cout = cout; a.Kpad = kpad;
    a.spu = g.spu; a.nunits = g.nunits; a.mirror = transposed ? 1 : 0;
    a.in_bytes = (unsigned)((size_t)B * g.D * g.D * g.D * cin * 2);
    a.w_bytes = (unsigned)((size_t)cout * kpad * 2);
#ifdef VOXG_PROBE
    { const char* e = getenv("TRICOLO_VOXG_DBG"); a.dbg = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
    // row tiles per pass: the unit's dense sites bound its active rows (2^3 / 4^3 units of few samples need 2-8 tiles, never a second pass)
    const int nsites = g.spu * g.D * g.D * g.D;
    const int nrt = nsites <= 32 ? 2 : (nsites <= 64 ? 4 : (nsites <= 128 ? 8 : 12));
#define TRI_VOXG_N(CT_, TN_, WC_, WK_, RING_, NRT_)                                                                       \
    if (g.ct == CT_ && nrt == NRT_)                                                                                       \
        return act_fmt == TRI_FMT_F16 ? voxg_launch_t<f16_t, TN_, WC_, WK_, NRT_, RING_>(a, g, stream)                    \
                                      : voxg_launch_t<bf16_t, TN_, WC_, WK_, NRT_, RING_>(a, g, stream);
#define TRI_VOXG(CT_, TN_, WC_, WK_, RING_)                                                                               \
    TRI_VOXG_N(CT_, TN_, WC_, WK_, RING_, 2) TRI_VOXG_N(CT_, TN_, WC_, WK_, RING_, 4)                                     \
    TRI_VOXG_N(CT_, TN_, WC_, WK_, RING_, 12)
    // (ring of 28 fragments for the 16-channel workgroups - four chunks of seven taps in flight; with 8 row tiles that variant spills, so
    //  it keeps 14 like the two-tile waves)
    TRI_VOXG(16, 1, 1, 4, 28) TRI_VOXG_N(16, 1, 1, 4, 14, 8)
    TRI_VOXG(32, 2, 1, 4, 14) TRI_VOXG_N(32, 2, 1, 4, 14, 8)
    TRI_VOXG(64, 2, 2, 2, 14) TRI_VOXG_N(64, 2, 2, 2, 14, 8)
#undef TRI_VOXG_N
#undef TRI_VOXG
    tri_set_error("conv(voxg): channel tile not instantiated");
    return TRI_ERR_UNSUPPORTED;
}

// ================================================================================================ level 1: 32 -> 64 channels on 16^3 / 32^3 grids
// conv_voxb_kernel.  Level 1 (sparse_cnn.py:17) has ONE 32-channel chunk and a filter bank of only 110 KB, but grids too large for a
// whole-sample slab: through conv_vox1_kernel (16^3 grids: 16-site x-runs by the mask, 2.2 x the active rows multiplied, every wave
// reading every B fragment: LDS-bound) it took 14.7 us at the bench shape, through conv_igemm_kernel (32^3 grids: im2col gather over the
// row list, one tap per k-step, register-staged) 88-95 us at 64^3 x 64 - a chain of exposed gather latencies at 6 x the MFMA time.
// Here conv_voxg_kernel's scheme runs over BRICKS of the grid with the filter bank stationary:
//   * a brick = 2 z-planes x BY rows x all D columns (BY = 4: 128 sites at 16^3, 256 at 32^3); its REGION = the brick plus a one-site
//     halo in z and y (4 x (BY + 2) rows; x needs none: the row pitch D + 1 supplies the zero column).  A persistent workgroup (two
//     per CU: one ranks / loads while the other multiplies) walks bricks wg, wg + G, ...;
//   * per brick: the region's site mask is ranked twice (ballot / popcount) - the active INTERIOR sites become the MFMA rows, all
//     active REGION sites the slab's load list; only those sites are loaded (64 B each) into a zero-padded LDS slab in
//     conv_voxg_kernel's two-half layout, and cleared again after the brick (the slab is zeroed once per workgroup);
//   * wave w owns output channels 16 w .. 16 w + 15 and keeps their 27 A fragments in registers for the whole launch (fragment-major
//     operand: 27 contiguous 1 KiB loads per wave); a brick's rows run in passes of up to 8 tiles of 16 rows, branch-free per tile
//     count, B fragments read one tap ahead; no K split, so a lane's accumulators are final: 8-byte stores, BatchNorm sums in
//     registers over all bricks, one record per workgroup (each wave writes its own 16 channels);
//   * the next brick's mask bytes are requested before the current brick's MFMAs.
struct VoxbArgs {
    const void* in;            // [B, D, D, D, 32] 16-bit (values at inactive sites are never read)
    const void* w;             // packed operand, fragment-major [4 tiles][27 k-steps][64 lanes][8]
    void* out;                 // [B, D, D, D, 64]; rows of inactive sites are not written
    const uint8_t* mask;       // [B * D^3] site mask, or NULL (every site active)
    float* stats;              // [grid][2][64] or NULL
    int B, nbricks;
    unsigned in_bytes;
#ifdef VOXG_PROBE
    long long* dbg;            // [grid][16]: cycles per phase summed over the workgroup's bricks (wave 0)
#endif
};
#ifdef VOXG_PROBE
#define VOXB_T(i) do { if (t == 0) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); acc_t[i] += now_ - last_t; last_t = now_; } } while (0)
#else
#define VOXB_T(i)
#endif

template <int D>
struct VoxbCfg {
    static constexpr int BZ = 2, BY = D == 16 ? 4 : 128 / D;                  // interior sites: 128 (16^3 grids: more, better balanced bricks) / 256
    static constexpr int RZ = BZ + 2, RY = BY + 2, P = D + 1;
    static constexpr int NREG = RZ * RY * D;                                  // region sites (768 / 640)
    static constexpr int ROUNDS = (NREG + 255) / 256;                         // ranking rounds (3)
    static constexpr int NS = RZ * RY * P + P + 2;                            // padded slab sites
    static constexpr int HALF = (NS * 32 + 255) / 256 * 256 + 128;
    static constexpr int SLAB = 2 * HALF;
    static constexpr int NYB = D / BY, PS = (D / BZ) * NYB;                   // bricks per sample
    // LDS: [2 table sets: load_off[NREG], load_glob[NREG], row_lds[256], row_glob[256]] | counts | slab
    static constexpr int NINT = BZ * BY * D;                                  // interior sites
    static constexpr int TABSET = (2 * NREG + 2 * NINT) * 4;
    static constexpr int CNT = 2 * TABSET;
    static constexpr int SLAB0 = (CNT + 256 + 255) / 256 * 256;
    static constexpr size_t SMEM = (size_t)SLAB0 + SLAB;
};

template <typename AT, int D>
__global__ __launch_bounds__(256, 2) void conv_voxb_kernel(const VoxbArgs p) {
    typedef VoxbCfg<D> C;
    typedef typename OpOf<AT>::E E;
    typedef Mma<E> MM;
    typedef typename MM::v8 v8;
    constexpr int NRT = 8, MAXL = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fq = lane >> 4;
    char* const slab = smem + C::SLAB0;
    int* const wcnt = (int*)(smem + C::CNT);                                  // [ROUNDS][4 waves][2]
    const int G = gridDim.x, wg = blockIdx.x;

    // ---- filter bank of this wave's 16 output channels: 27 contiguous fragment loads, kept for the whole launch
    v8 wf[27];
    {
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 4 * 27 * 1024, 0x00020000);
#pragma unroll
        for (int k = 0; k < 27; ++k)
            wf[k] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (unsigned)((wave * 27 + k) * 1024 + lane * 16), 0, 0));
    }
    // ---- region site of this thread in each ranking round (constant over bricks): e = t + 256 j -> (zr, yr, x)
    int rzy[C::ROUNDS], rx[C::ROUNDS];                                        // zr * 64 + yr, x  (-1: past the region)
    int sp32[C::ROUNDS];                                                      // padded slab site * 32 (bytes inside a half)
    bool inner[C::ROUNDS];
#pragma unroll
    for (int j = 0; j < C::ROUNDS; ++j) {
        const int e = t + 256 * j;
        const int rr = e / D, x = e % D, zr = rr / C::RY, yr = rr % C::RY;
        const bool ok = e < C::NREG;
        rzy[j] = ok ? zr * 64 + yr : -1;
        rx[j] = x;
        sp32[j] = ((zr * C::RY + yr) * C::P + x + 1) * 32;
        inner[j] = ok && zr >= 1 && zr <= C::BZ && yr >= 1 && yr <= C::BY;
    }
    auto brick_of = [&](int j, int& b, int& z0, int& y0) {
        b = j / C::PS;
        const int sp = (j % C::PS + b * (C::PS / 2 + 1)) % C::PS;             // rotated by the sample (see conv_vox0_kernel)
        z0 = (sp / C::NYB) * C::BZ;
        y0 = (sp % C::NYB) * C::BY;
    };
    auto load_mask = [&](int j, uint8_t (&m)[C::ROUNDS], int (&gsite)[C::ROUNDS]) {
        int b, z0, y0;
        brick_of(j, b, z0, y0);
#pragma unroll
        for (int r = 0; r < C::ROUNDS; ++r) {
            const int z = z0 - 1 + (rzy[r] >> 6), y = y0 - 1 + (rzy[r] & 63);
            const bool ok = rzy[r] >= 0 && j < p.nbricks && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)D;
            gsite[r] = ((b * D + z) * D + y) * D + rx[r];
            m[r] = ok ? (p.mask ? p.mask[gsite[r]] : (uint8_t)1) : (uint8_t)0;
        }
    };
    // zero the slab once: afterwards a brick writes its active region sites and clears exactly those again
    for (int i = t * 16; i < C::SLAB; i += 256 * 16) *(uint4*)(slab + i) = make_uint4(0u, 0u, 0u, 0u);

    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const int lane_part = (fq >> 1) * C::HALF + (fq & 1) * 16;
    const int centre = (C::RY * C::P + C::P + 1) * 32;                        // (the row tables hold the (-1,-1,-1) corner: site - centre)
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cq = cs;
    uint8_t mk[C::ROUNDS];
    int gs[C::ROUNDS];
    load_mask(wg, mk, gs);
    int nload_prev = 0;
    int it = 0;
#ifdef VOXG_PROBE
    long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_t = (long long)__builtin_amdgcn_s_memtime();
    const long long t_begin = last_t;
    int n_nonempty = 0, n_rows = 0;
#endif
#pragma unroll 1
    for (int j = wg; j < p.nbricks; j += G, ++it) {
        int* const tab = (int*)(smem + (it & 1) * C::TABSET);
        int* const load_off = tab, * const load_glob = tab + C::NREG, * const row_lds = tab + 2 * C::NREG, * const row_glob = row_lds + C::NINT;
        int* const ptab = (int*)(smem + ((it & 1) ^ 1) * C::TABSET);          // the previous brick's load list
        VOXB_T(7);
        __syncthreads();                                                      // [S1] every wave is done with the previous brick's slab
        VOXB_T(0);
        // clear the previous brick's sites (its load list is still in the other table set)
        for (int e = t; e < 4 * nload_prev; e += 256) {
            const int r = e >> 2, q = e & 3;
            *(uint4*)(slab + ptab[r] + (q >> 1) * C::HALF + (q & 1) * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
        // rank: region sites -> load list, interior sites -> MFMA rows (raster order)
        unsigned long long ball[C::ROUNDS], balr[C::ROUNDS];
#pragma unroll
        for (int r = 0; r < C::ROUNDS; ++r) {
            ball[r] = __ballot(mk[r] != 0);
            balr[r] = __ballot(mk[r] != 0 && inner[r]);
            if (lane == 0) { wcnt[(r * 4 + wave) * 2] = __popcll(ball[r]); wcnt[(r * 4 + wave) * 2 + 1] = __popcll(balr[r]); }
        }
        VOXB_T(1);
        __syncthreads();
        VOXB_T(2);
        int nload = 0, nrows = 0;
        {
            int basel[C::ROUNDS], baser[C::ROUNDS];
#pragma unroll
            for (int r = 0; r < C::ROUNDS; ++r)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w == wave) { basel[r] = nload; baser[r] = nrows; }
                    nload += wcnt[(r * 4 + w) * 2];
                    nrows += wcnt[(r * 4 + w) * 2 + 1];
                }
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int r = 0; r < C::ROUNDS; ++r)
                if (mk[r]) {
                    const int kl = basel[r] + __popcll(ball[r] & below);
                    load_off[kl] = sp32[r];
                    load_glob[kl] = gs[r];
                    if (inner[r]) {
                        const int kr = baser[r] + __popcll(balr[r] & below);
                        row_lds[kr] = sp32[r] - centre;
                        row_glob[kr] = gs[r];
                    }
                }
        }
        __syncthreads();                                                      // [S2] tables visible; the clears are ordered before the slab writes below
        VOXB_T(3);
        // the next brick's mask bytes fly under this brick's loads and MFMAs
        load_mask(j + G, mk, gs);
        nload_prev = nload;
        if (nrows > 0) {
            // slab: 64 bytes of every active region site (four 16-byte pieces)
            {
                uint4 pre[MAXL];
#pragma unroll
                for (int u = 0; u < MAXL; ++u) {
                    const int e = t + 256 * u;
                    if (e < 4 * nload) pre[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(irs, (unsigned)(load_glob[e >> 2] * 64 + (e & 3) * 16), 0, 0));
                }
#pragma unroll
                for (int u = 0; u < MAXL; ++u) {
                    const int e = t + 256 * u;
                    if (e < 4 * nload) *(uint4*)(slab + load_off[e >> 2] + ((e & 3) >> 1) * C::HALF + (e & 1) * 16) = pre[u];
                }
#pragma unroll 1
                for (int e = t + 256 * MAXL; e < 4 * nload; e += 256) {
                    const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(irs, (unsigned)(load_glob[e >> 2] * 64 + (e & 3) * 16), 0, 0));
                    *(uint4*)(slab + load_off[e >> 2] + ((e & 3) >> 1) * C::HALF + (e & 1) * 16) = v;
                }
            }
            __syncthreads();                                                  // [S3] slab complete
            VOXB_T(4);
#ifdef VOXG_PROBE
            ++n_nonempty; n_rows += nrows;
#endif
#pragma unroll 1
            for (int r0 = 0; r0 < nrows; r0 += 16 * NRT) {
                const int nrt = min(NRT, (nrows - r0 + 15) >> 4);
                auto pass = [&](auto NTc) {
                    constexpr int NT = decltype(NTc)::value;
                    int lbase[NT];
                    f32x4 acc[NT];
#pragma unroll
                    for (int rt = 0; rt < NT; ++rt) {
                        const int r = r0 + rt * 16 + fr;
                        lbase[rt] = (r < nrows ? row_lds[r] : 0) + lane_part;
                        acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                    v8 bf[NT];
#pragma unroll
                    for (int rt = 0; rt < NT; ++rt) bf[rt] = *(const v8*)(slab + lbase[rt]);
#pragma unroll
                    for (int k = 0; k < 27; ++k) {
                        constexpr int dummy = 0;
                        const int kn = k + 1;
                        const int toff = ((kn / 9 * C::RY + (kn / 3) % 3) * C::P + kn % 3) * 32;
#pragma unroll
                        for (int rt = 0; rt < NT; ++rt) {
                            acc[rt] = MM::mma(wf[k], bf[rt], acc[rt]);
                            if (k + 1 < 27) bf[rt] = *(const v8*)(slab + lbase[rt] + toff);
                        }
                        (void)dummy;
                    }
#pragma unroll
                    for (int rt = 0; rt < NT; ++rt) {
                        const int r = r0 + rt * 16 + fr;
                        if (r < nrows) {
                            typedef E e4 __attribute__((ext_vector_type(4)));
                            const e4 h = __builtin_convertvector(acc[rt], e4);
                            *(e4*)((AT*)p.out + (size_t)row_glob[r] * 64 + wave * 16 + fq * 4) = h;
                            const f32x4 rv = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                            cs += rv;
                            cq += rv * rv;
                        }
                    }
                };
                if (nrt <= 1) pass(std::integral_constant<int, 1>{});
                else if (nrt <= 2) pass(std::integral_constant<int, 2>{});
                else if (nrt <= 3) pass(std::integral_constant<int, 3>{});
                else if (nrt <= 4) pass(std::integral_constant<int, 4>{});
                else if (nrt <= 6) pass(std::integral_constant<int, 6>{});
                else pass(std::integral_constant<int, 8>{});
            }
            VOXB_T(5);
        }
    }
#ifdef VOXG_PROBE
    if (p.dbg && t == 0) {
        for (int i = 0; i < 8; ++i) p.dbg[(size_t)blockIdx.x * 16 + i] = acc_t[i];
        p.dbg[(size_t)blockIdx.x * 16 + 8] = (long long)__builtin_amdgcn_s_memtime() - t_begin;
        p.dbg[(size_t)blockIdx.x * 16 + 9] = n_nonempty;
        p.dbg[(size_t)blockIdx.x * 16 + 10] = n_rows;
        p.dbg[(size_t)blockIdx.x * 16 + 11] = it;
    }
#endif
    if (p.stats) {                                                            // one record per workgroup: wave w owns channels 16 w ..
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s_ = cs[r], q_ = cq[r];
            s_ += voxg_row_ror<8>(s_); q_ += voxg_row_ror<8>(q_);
            s_ += voxg_row_ror<4>(s_); q_ += voxg_row_ror<4>(q_);
            s_ += voxg_row_ror<2>(s_); q_ += voxg_row_ror<2>(q_);
            s_ += voxg_row_ror<1>(s_); q_ += voxg_row_ror<1>(q_);
            if (fr == 0) {
                p.stats[(size_t)blockIdx.x * 128 + wave * 16 + fq * 4 + r] = s_;
                p.stats[(size_t)blockIdx.x * 128 + 64 + wave * 16 + fq * 4 + r] = q_;
            }
        }
    }
}

static bool voxb_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_VOXB"); v = (e && e[0] == '1') ? 1 : 0; }      // A/B switch: level 1 stays on conv_igemm_kernel
    return v == 1;
}

bool tri_internal_voxb_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                                int pd, int ph, int pw, TriVoxbGeom* g) {
    if (voxb_disabled()) return false;
    const int D = ID;
    // 16^3 grids (32^3 inputs) by default: 15.1 / 23.4 us against conv_vox1_kernel's 16.4 / 26.0 at batch 32 / 64.  On 32^3 grids (64^3
    // inputs, 8,192 bricks, three quarters of them empty) the per-brick chain - mask, two rankings, tables, slab gather, three barriers,
    // and a B-fragment read per MFMA with two workgroups sharing the LDS (stamps: 11 k cycles per non-empty brick against 2.6 k of
    // MFMAs) - makes it 100 us against conv_igemm_kernel's 88: TRICOLO_VOXB_32=1 plans it there anyway (tests do, in a child process)
    static int big = -1;
    if (big < 0) { const char* e = getenv("TRICOLO_VOXB_32"); big = (e && e[0] == '1') ? 1 : 0; }
    if (IH != D || IW != D || OD != D || OH != D || OW != D || (D != 16 && !(D == 32 && big))) return false;
    if (cin != 32 || cout != 64 || KD != 3 || KH != 3 || KW != 3 || stride != 1 || pd != 1 || ph != 1 || pw != 1) return false;
    if ((long)B * D * D * D * 64 >= (1L << 31)) return false;                 // 32-bit buffer offsets
    g->D = D;
    g->nbricks = B * (D / 2) * (D / (D == 16 ? 4 : 128 / D));
    int grid = 2 * tri_internal_num_cus();                                    // persistent: two workgroups per CU
    if (grid > g->nbricks) grid = g->nbricks;
    g->grid = grid;
    return true;
}

template <typename AT, int D>
static int voxb_launch_t(const VoxbArgs& a, int grid, hipStream_t stream) {
    typedef VoxbCfg<D> C;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)conv_voxb_kernel<AT, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
        attr = true;
    }
    conv_voxb_kernel<AT, D><<<grid, 256, C::SMEM, stream>>>(a);
    return tri_check_launch("tri_conv(voxb)");
}

int tri_internal_voxb_launch(const TriVoxbGeom& g, int B, const void* in, const void* w, void* out, const uint8_t* mask, float* stats, int act_fmt,
                             hipStream_t stream) {
    VoxbArgs a{};
    a.in = in; a.w = w; a.out = out; a.mask = mask; a.stats = stats;
    a.B = B; a.nbricks = g.nbricks;
    a.in_bytes = (unsigned)((size_t)B * g.D * g.D * g.D * 64);
#ifdef VOXG_PROBE
    { const char* e = getenv("TRICOLO_VOXG_DBG"); a.dbg = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
    if (g.D == 16) return act_fmt == TRI_FMT_F16 ? voxb_launch_t<f16_t, 16>(a, g.grid, stream) : voxb_launch_t<bf16_t, 16>(a, g.grid, stream);
    return act_fmt == TRI_FMT_F16 ? voxb_launch_t<f16_t, 32>(a, g.grid, stream) : voxb_launch_t<bf16_t, 32>(a, g.grid, stream);
}
