// In-kernel phase stamps of conv_wgrad_dma_kernel (wave 0 of every workgroup) on the weight gradients of the bench shape's 3x3 layers.
// Build (cross-compiles here, runs on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWGRAD_STAMPS -Iinclude -o tools/probes/build/wgrad_probe tools/probes/wgrad_probe.hip tricolo_amd/csrc/misc.hip tricolo_amd/csrc/conv_igemm.hip tricolo_amd/csrc/conv_vox.hip
// Usage: wgrad_probe [images = 192] [jobs = 1]
// Stamp ids (WSTAMP in conv_wgrad.hip): 2 tap table + gather plan staged, 3 lane constants, then per 64-position step 4 loop top,
// 5 DMA wait, 6 barrier, 7 next stage issued, 8 32 MFMAs; 9 loop done, 10 slab stored.
#include "../../tricolo_amd/csrc/conv_wgrad.hip"
#include <vector>
#include <map>
#include <algorithm>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 192;
    struct L { int hw, c; } layers[] = {{32, 64}, {16, 128}, {8, 256}, {4, 512}};
    hipMalloc(&g_wgrad_dbg, (size_t)4096 * 256 * 8);
    for (auto l : layers) {
        TriConvDesc d = {B, 1, l.hw, l.hw, l.c, 1, l.hw, l.hw, l.c, 1, 3, 3, 1, 0, 1, 1};
        const int njobs = argc > 2 ? atoi(argv[2]) : 1;             // > 1: that many copies of the layer in ONE grouped launch
        const size_t M = (size_t)B * l.hw * l.hw, K = 9 * l.c;
        void *in, *dout, *plan, *ws; float* dw;
        hipMalloc(&in, M * l.c * 2); hipMalloc(&dout, M * l.c * 2); hipMalloc(&dw, K * l.c * 4);
        hipMemset(in, 0x11, M * l.c * 2); hipMemset(dout, 0x11, M * l.c * 2);
        hipMalloc(&plan, tri_conv_plan_bytes(&d));
        tri_conv_plan_build(&d, plan, nullptr);
        const size_t wsb = tri_conv_wgrad_workspace(&d);
        hipMalloc(&ws, wsb * njobs);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            hipMemset(g_wgrad_dbg, 0, (size_t)4096 * 256 * 8);
            hipEventRecord(e0, 0);
            int rc;
            if (njobs > 1) {
                TriWgradJob jobs[TRI_WGRAD_JOBS_MAX];
                TriWgradReduce pend[TRI_WGRAD_JOBS_MAX];
                for (int j = 0; j < njobs; ++j) jobs[j] = TriWgradJob{&d, in, dout, plan, (char*)ws + j * wsb, wsb, dw, (long)K, 1, 9, l.c, 1.0f};
                rc = tri_conv_wgrad_partial_group(jobs, njobs, TRI_FMT_F16, pend, nullptr);
                if (!rc) rc = tri_wgrad_reduce_grouped(pend, njobs, nullptr);
                if (it == 0) printf("   grouped x%d: splits %d per job\n", njobs, pend[0].splits);
            } else
            rc = tri_conv_wgrad(&d, in, dout, nullptr, plan, ws, wsb, dw, (long)K, 1, 9, l.c, 0, TRI_FMT_F16, 1.0f, nullptr, nullptr, nullptr);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            if (rc) { printf("tri_conv_wgrad: %d %s\n", rc, tri_last_error()); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        std::vector<long long> h((size_t)4096 * 256);
        hipMemcpy(h.data(), g_wgrad_dbg, h.size() * 8, hipMemcpyDeviceToHost);
        printf("== %dx%d C=%d family %d: partial + reduce %.1f us (event-timed), slabs %.1f MB\n", l.hw, l.hw, l.c, tri_conv_wgrad_kernel_family(&d, TRI_FMT_F16),
               best * 1e3, wsb / 1e6);
        long long tmin = -1, tmax = 0; int nwg = 0;
        std::map<int, double> tot;
        const long long m = 0xFFFFFFFFFFFFll;
        for (int wg = 0; wg < 4096; ++wg) {
            const long long* s = &h[(size_t)wg * 256];
            const int n = (int)s[255];
            if (n < 2) continue;
            ++nwg;
            if (tmin < 0 || (s[0] & m) < tmin) tmin = s[0] & m;
            tmax = std::max(tmax, s[n - 1] & m);
            for (int i = 1; i < n; ++i) tot[(int)(s[i] >> 48)] += (double)((s[i] & m) - (s[i - 1] & m));
        }
        double all = 0;
        for (auto& kv : tot) all += kv.second;
        printf("   %d workgroups, first start -> last end %lld cycles; mean cycles per workgroup spent reaching stamp id:\n", nwg, tmax - tmin);
        for (auto& kv : tot) printf("     %2d %9.0f  (%4.1f %%)\n", kv.first, kv.second / nwg, 100.0 * kv.second / all);
        printf("     total %9.0f\n", all / nwg);
        for (int wg : {0, nwg / 2, nwg - 1}) {
            const long long* s = &h[(size_t)wg * 256];
            const int n = (int)s[255];
            printf("   wg %d: start +%lld, end +%lld, stamps %d; first 36 deltas:", wg, (s[0] & m) - tmin, (s[n - 1] & m) - tmin, n);
            for (int i = 1; i < std::min(n, 37); ++i) printf(" %d:%lld", (int)(s[i] >> 48), (s[i] & m) - (s[i - 1] & m));
            printf("\n");
        }
        hipFree(in); hipFree(dout); hipFree(dw); hipFree(plan); hipFree(ws);
    }
    return 0;
}
