// Operand-feed ceilings of one CU (gfx950): how many bytes per clock a workgroup can pull from L2 / HBM when the weights go
//   * through LDS-DMA like the activations (what conv_halo_rows_kernel / conv_dma_kernel do today), or
//   * straight into registers as MFMA fragments (buffer_load_dwordx4, fragment-major packed weights: 1 KiB contiguous per wave load)
// while the activations keep using LDS-DMA.  No MFMAs, no LDS reads: the numbers are CEILINGS of the feed, per CU and for the chip.
// The traffic mix is layer4's (512 -> 512, 3x3, 128 x 64 tile: 590 KiB of weights + 295 KiB of slab per tile; weights shared by the
// workgroups of one output-channel tile -> L2 hits, activations private).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -o tools/probes/build/feed_probe tools/probes/feed_probe.hip
// Usage: feed_probe            (prints a table: mode x workgroups per CU)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../tricolo_amd/csrc/common.h"

constexpr int WP = 4, AP = 2;            // 1 KiB pieces per wave per step: weights, activations
constexpr int NB = 3, D = 2;             // buffers, steps in flight
constexpr int STEPS = 36;                // 36 x (16 KiB + 8 KiB) per workgroup

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u4 load16_async(v4i rsrc, int voff) {
    u4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rsrc) : "memory");
    return r;
}

// MODE bit 0: weights through registers (else LDS-DMA), bit 1: no weights at all, bit 2: no activations at all
template <int MODE>
__global__ __launch_bounds__(256) void feed_kernel(const char* w, unsigned wbytes, const char* a, unsigned abytes, int ntn, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const v4i wr = make_rsrc_words(w, wbytes), ar = make_rsrc_words(a, abytes);
    const unsigned lds0 = lds_addr(smem);
    constexpr bool WREG = MODE & 1, NOW = (MODE & 2) != 0, NOA = (MODE & 4) != 0;
    constexpr int WSTEP = 4 * WP * 1024, ASTEP = 4 * AP * 1024;
    constexpr int RING = (WREG || NOW ? 0 : WSTEP) + (NOA ? 0 : ASTEP);          // LDS bytes per step
    const int wbase = (blockIdx.x % ntn) * (STEPS * WSTEP) + wave * WP * 1024 + lane * 16;
    const int abase = (int)(((size_t)blockIdx.x * (STEPS * ASTEP)) % (abytes - STEPS * ASTEP)) / 16 * 16 + wave * AP * 1024 + lane * 16;
    u4 wb[NB][WP];
    u4 acc = {0u, 0u, 0u, 0u};
    constexpr int PER_STEP = (NOW ? 0 : WP) + (NOA ? 0 : AP);                     // vector-memory instructions per wave per step
    auto issue = [&](int s, int buf) {
        const unsigned ring = lds0 + buf * RING;
        if (!NOA)
#pragma unroll
            for (int i = 0; i < AP; ++i) dma16_async(ar, ring + (wave * AP + i) * 1024, abase + s * ASTEP + i * 1024);
        if (!NOW) {
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                if (WREG) wb[buf][i] = load16_async(wr, wbase + s * WSTEP + i * 1024);
                else dma16_async(wr, ring + (NOA ? 0 : ASTEP) + (wave * WP + i) * 1024, wbase + s * WSTEP + i * 1024);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s, s % NB);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + D < STEPS) issue(s + D, (s + D) % NB);
        const int left = (STEPS - 1 - s) < D ? (STEPS - 1 - s) : D;
        // step s has landed when at most `left` later steps are outstanding
        if (left * PER_STEP == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (left * PER_STEP == PER_STEP) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_STEP) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER_STEP) : "memory");
        if (WREG && !NOW) {
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                asm volatile("" : "+v"(wb[s % NB][i]));
                acc ^= wb[s % NB][i];
            }
        }
        if (RING) __syncthreads();                                               // (what a consumer of the LDS ring would need)
    }
    if (RING) acc.x ^= *(const unsigned*)(smem + t * 4);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
static void run(const char* name, const char* w, unsigned wbytes, const char* a, unsigned abytes, unsigned* sink) {
    constexpr bool WREG = MODE & 1, NOW = (MODE & 2) != 0, NOA = (MODE & 4) != 0;
    constexpr int RING = (WREG || NOW ? 0 : 4 * WP * 1024) + (NOA ? 0 : 4 * AP * 1024);
    const size_t smem = (size_t)NB * RING + 1024;
    hipFuncSetAttribute((const void*)feed_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-44s", name);
    for (int per_cu = 1; per_cu <= 4; ++per_cu) {
        if (per_cu * smem > 160 * 1024) { printf("  %18s", "-"); continue; }
        const int grid = 256 * per_cu;
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0, 0);
            feed_kernel<MODE><<<grid, 256, smem, 0>>>(w, wbytes, a, abytes, 8, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it && ms < best) best = ms;
        }
        const double bytes = (double)grid * STEPS * ((NOW ? 0 : 4 * WP * 1024) + (NOA ? 0 : 4 * AP * 1024));
        printf("  %6.1f us %5.1f B/clk", best * 1e3, bytes / (best * 1e-3) / 256 / 2.4e9);
    }
    printf("\n");
}

int main() {
    const unsigned wbytes = 8u * STEPS * 4 * WP * 1024, abytes = 1u << 30;
    char *w, *a; unsigned* sink;
    hipMalloc(&w, wbytes); hipMalloc(&a, abytes); hipMalloc(&sink, 4);
    hipMemset(w, 1, wbytes); hipMemset(a, 2, abytes); hipMemset(sink, 0, 4);
    printf("per CU feed at 2.4 GHz; columns: 1 / 2 / 3 / 4 workgroups of 256 threads per CU (time of the launch, bytes per clock per CU)\n");
    run<0>("weights + activations through LDS-DMA", w, wbytes, a, abytes, sink);
    run<1>("weights -> registers, activations LDS-DMA", w, wbytes, a, abytes, sink);
    run<4>("weights only, LDS-DMA", w, wbytes, a, abytes, sink);
    run<5>("weights only, -> registers", w, wbytes, a, abytes, sink);
    run<2>("activations only, LDS-DMA", w, wbytes, a, abytes, sink);
    return 0;
}
