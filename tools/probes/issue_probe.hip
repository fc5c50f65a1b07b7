// What does ISSUING an LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction) cost the issuing wave, and is that
// cost contention on the CU's vector-memory path or a fixed price per instruction?  conv_halo_rows_kernel's stamps (halo_probe) show
// ~115-140 cycles per piece on a wave that also runs the MFMAs, with all four waves of the workgroup issuing at the same moment.
// One workgroup of 256 threads per CU, L2-resident source (the workgroups share 2 MiB), s_memtime around the issue burst of wave 0:
//   mode 0: all four waves issue P pieces at the same moment (what the kernels do today)
//   mode 1: only wave 0 issues P pieces, the others wait at the barrier
//   mode 2: wave 0 issues 4 P pieces (every wave's share, other waves' quarters through the scalar offset), the others wait
//   mode 3: as mode 0, but wave w starts w * 64 cycles late (s_sleep)
//   mode 4: wave 0 issues 4 P pieces while waves 1-3 run MFMAs (the rotating-producer form)
//   mode 5: all four waves interleave P pieces with 8 MFMAs each (today's k-step shape); time of the whole burst
//   mode 6: 8 P MFMAs alone (baseline of mode 5)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -o tools/probes/build/issue_probe tools/probes/issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../../tricolo_amd/csrc/common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16_soff(v4i rsrc, unsigned lds_dst, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

template <int MODE, int P>
__global__ __launch_bounds__(256, 1) void issue_kernel(const char* src, unsigned bytes, long long* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const v4i rs = make_rsrc_words(src, bytes);
    const unsigned lds0 = lds_addr(smem);
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f); b[i] = (_Float16)(i * 0.5f); }
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    long long t_issue = 0, t_land = 0;
    for (int r = 0; r < reps; ++r) {
        const int base = ((blockIdx.x * 7 + r * 13) % 64) * (32 * 1024);       // 2 MiB window shared by all workgroups
        __syncthreads();
        if (MODE == 3) for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(1);
        const long long c0 = __builtin_readcyclecounter();
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < P; ++i) dma16_async(rs, lds0 + (i * 4 + wave) * 1024, base + (i * 4 + wave) * 1024 + lane * 16);
        } else if (MODE == 1) {
            if (wave == 0) {
#pragma unroll
                for (int i = 0; i < P; ++i) dma16_async(rs, lds0 + (i * 4) * 1024, base + (i * 4) * 1024 + lane * 16);
            }
        } else if (MODE == 2 || MODE == 4) {
            if (wave == 0) {
#pragma unroll
                for (int i = 0; i < 4 * P; ++i) dma16_soff(rs, lds0 + i * 1024, base + lane * 16, i * 1024);
            } else if (MODE == 4) {
#pragma unroll 1
                for (int k = 0; k < 4 * P; ++k)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
                dma16_async(rs, lds0 + (i * 4 + wave) * 1024, base + (i * 4 + wave) * 1024 + lane * 16);
            }
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        }
        if (MODE >= 4) { asm volatile("" : "+v"(acc[0]), "+v"(acc[7])); }
        const long long c1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long c2 = __builtin_readcyclecounter();
        if (r) { t_issue += c1 - c0; t_land += c2 - c1; }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 123.456f) out[0] = 1;
    if (lane == 0) {
        out[(blockIdx.x * 4 + wave) * 2 + 0] = t_issue / (reps - 1);
        out[(blockIdx.x * 4 + wave) * 2 + 1] = t_land / (reps - 1);
    }
}

template <int MODE, int P>
static void run(const char* name, const char* src, unsigned bytes, long long* out) {
    const size_t smem = 64 * 1024;
    hipFuncSetAttribute((const void*)issue_kernel<MODE, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int it = 0; it < 2; ++it) issue_kernel<MODE, P><<<256, 256, smem, 0>>>(src, bytes, out, 20);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 4 * 2);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    printf("%-66s P=%2d |", name, P);
    for (int w = 0; w < 4; ++w) {
        std::vector<long long> is, ld;
        for (int b = 0; b < 256; ++b) { is.push_back(h[(b * 4 + w) * 2]); ld.push_back(h[(b * 4 + w) * 2 + 1]); }
        std::sort(is.begin(), is.end()); std::sort(ld.begin(), ld.end());
        printf("  w%d issue %5lld land +%5lld", w, is[128], ld[128]);
    }
    printf("\n");
}

int main() {
    const unsigned bytes = 4u << 20;
    char* src; long long* out;
    hipMalloc(&src, bytes); hipMalloc(&out, 256 * 4 * 2 * 8);
    hipMemset(src, 1, bytes);
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("median over 256 workgroups (one per CU) of the mean over 19 bursts; cycles of the burst on each wave, then the wait for its landing\n");
    run<0, 2>("0: four waves issue P pieces at once", src, bytes, out);
    run<0, 6>("0: four waves issue P pieces at once", src, bytes, out);
    run<1, 2>("1: wave 0 alone issues P pieces", src, bytes, out);
    run<1, 6>("1: wave 0 alone issues P pieces", src, bytes, out);
    run<2, 2>("2: wave 0 issues 4P pieces (soffset), others wait", src, bytes, out);
    run<2, 6>("2: wave 0 issues 4P pieces (soffset), others wait", src, bytes, out);
    run<3, 6>("3: four waves, each 64 cycles after the other", src, bytes, out);
    run<4, 2>("4: wave 0 issues 4P pieces, waves 1-3 run 32P MFMAs", src, bytes, out);
    run<4, 6>("4: wave 0 issues 4P pieces, waves 1-3 run 32P MFMAs", src, bytes, out);
    run<5, 6>("5: four waves, P x (8 MFMAs + 1 piece)", src, bytes, out);
    run<6, 6>("6: four waves, P x 8 MFMAs", src, bytes, out);
    return 0;
}
