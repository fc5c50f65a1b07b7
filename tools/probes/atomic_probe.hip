// What do exact fixed-point BatchNorm sums through device-scope int64 atomics cost a producer kernel?  (Idea priced here: conv epilogues
// add their per-workgroup partial sums - converted to fixed point, so the total does not depend on the order - into [3][C] accumulators
// instead of writing [2][C] records, and the consumers derive the BatchNorm coefficients themselves: the 45 finalize launches of a step,
// 130-160 us on the default line, would go.)  G workgroups of 256 threads spin for `work` microseconds (a stand-in for the conv), then
//   mode 0: write a [2][C] fp32 record each (today)
//   mode 1: C * 3 atomicAdd(unsigned long long) per workgroup into ONE [3][C] accumulator (all workgroups collide on every address)
//   mode 2: as mode 1 with 8 accumulator copies, copy = blockIdx % 8 (one per XCD if workgroups are dealt round-robin)
// and a second kernel checks the totals.  Prints the kernel time (HIP events, back-to-back average) per mode.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/build/atomic_probe tools/probes/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ __launch_bounds__(256) void producer(int mode, int C, long long spin_cycles, float* rec, unsigned long long* acc) {
    const long long t0 = clock64();
    while (clock64() - t0 < spin_cycles) { }
    __syncthreads();
    const int t = threadIdx.x;
    if (mode == 0) {
        for (int c = t; c < 2 * C; c += 256) rec[(size_t)blockIdx.x * 2 * C + c] = 1.0f + (float)(c & 7);
        return;
    }
    unsigned long long* a = acc + (mode == 2 ? (size_t)(blockIdx.x & 7) * 3 * C : 0);
    for (int c = t; c < 3 * C; c += 256) atomicAdd(a + c, (unsigned long long)(1 + (c & 7)));
}

int main(int argc, char** argv) {
    const int reps = 50;
    float* rec; unsigned long long* acc;
    hipMalloc(&rec, (size_t)8192 * 2 * 512 * 4);
    hipMalloc(&acc, (size_t)8 * 3 * 512 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfgs[][2] = {{512, 64}, {384, 128}, {256, 256}, {192, 512}, {4096, 64}};
    for (auto& cf : cfgs) {
        const int G = cf[0], C = cf[1];
        for (double work_us : {0.0, 20.0}) {
            for (int mode = 0; mode < 3; ++mode) {
                hipMemset(acc, 0, (size_t)8 * 3 * 512 * 8);
                const long long spin = (long long)(work_us * 100.0);          // clock64 on gfx9 counts at 100 MHz
                producer<<<G, 256>>>(mode, C, spin, rec, acc);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < reps; ++r) producer<<<G, 256>>>(mode, C, spin, rec, acc);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> h(8 * 3 * C);
                hipMemcpy(h.data(), acc, h.size() * 8, hipMemcpyDeviceToHost);
                unsigned long long tot = 0, want = (unsigned long long)G * (reps + 1) * 1;   // channel 0 adds 1 per workgroup per launch
                for (int k = 0; k < (mode == 2 ? 8 : 1); ++k) tot += h[(size_t)k * 3 * C];
                printf("G %4d C %3d work %4.0f us mode %d: %7.2f us per launch%s\n", G, C, work_us, mode, ms * 1e3 / reps,
                       mode == 0 ? "" : (tot == want ? "  (sum exact)" : "  (SUM WRONG)"));
            }
        }
    }
    return 0;
}
