// In-kernel phase stamps of the halo kernels (conv_halo2d_kernel / conv_halo_rows_kernel, whichever the plan picks - force one with
// TRICOLO_HALO_ROWS=0 / 2; wave 0 of every workgroup) on the four 3x3 layers of the bench shape.  Stamp ids (HSTAMP in conv_igemm.hip):
// conv_halo2d_kernel 2 constants done, 3 slab issued, 4 slab landed, 5 unit's weights landed, 6 barrier, 7 next unit issued, 8 MFMAs,
// 9 epilogue start, 10 stores issued, 11 tile done; conv_halo_rows_kernel 2 constants + first DMAs issued, 3 first fragments read,
// per kernel row 4 k-steps 0-3, 5 DMA wait, 6 barrier, 8 k-steps 4-5 + DMA issue, 10 epilogue, 11 statistics written.
// Build (cross-compiles here, runs on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHALO_STAMPS -Iinclude -o tools/probes/build/halo_probe tools/probes/halo_probe.hip \
//         tricolo_amd/csrc/{misc,conv_c64,conv_vox,conv_pw,conv_s2g}.hip  (usage: halo_probe [images = 192] [first layer 0..3])
// Prints, per layer: kernel time, and for a few workgroups the cycles between consecutive stamps summed by phase.
#include "../../tricolo_amd/csrc/conv_igemm.hip"
#include <vector>
#include <map>
#include <algorithm>

static const char* kPhase[] = {"", "start", "2", "3", "4", "5", "6", "7", "8", "9", "10", "11"};

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 192;                       // images = per-GPU batch 32 x 6 views
    const int first = argc > 2 ? atoi(argv[2]) : 0;                     // first layer probed (0 = 64 channels ... 3 = 512)
    struct L { int hw, c; } layers[] = {{32, 64}, {16, 128}, {8, 256}, {4, 512}};
    setvbuf(stdout, nullptr, _IONBF, 0);
    setenv("TRICOLO_HALO_TM3", "0", 0);                                 // the 192-position tiles use all 160 KB of LDS: no room for the stamp area
    hipMalloc(&g_halo_dbg, (size_t)4096 * 256 * 8);
    int li = -1;
    for (auto l : layers) {
        if (++li < first) continue;
        TriConvDesc d = {B, 1, l.hw, l.hw, l.c, 1, l.hw, l.hw, l.c, 1, 3, 3, 1, 0, 1, 1};
        const size_t M = (size_t)B * l.hw * l.hw, K = 9 * l.c;
        void *in, *w, *out; float* stats;
        hipMalloc(&in, M * l.c * 2); hipMalloc(&out, M * l.c * 2); hipMalloc(&w, K * l.c * 2);
        hipMemset(in, 0x11, M * l.c * 2); hipMemset(w, 0x11, K * l.c * 2);
        const int nt = tri_conv_num_mtiles(&d, 0);
        hipMalloc(&stats, (size_t)nt * 2 * l.c * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            hipMemset(g_halo_dbg, 0, (size_t)4096 * 256 * 8);
            hipEventRecord(e0, 0);
            int rc = tri_conv_fwd(&d, in, w, nullptr, out, nullptr, nullptr, 0, 0, stats, TRI_FMT_F16, nullptr, 0, nullptr, nullptr, nullptr);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            if (rc) { printf("tri_conv_fwd: %d %s\n", rc, tri_last_error()); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        std::vector<long long> h((size_t)4096 * 256);
        hipMemcpy(h.data(), g_halo_dbg, h.size() * 8, hipMemcpyDeviceToHost);
        printf("== %dx%d C=%d family %d: %.1f us (event-timed, single launch)\n", l.hw, l.hw, l.c, tri_conv_kernel_family(&d, 0, 0), best * 1e3);
        // workgroup spans and phase sums
        long long tmin = -1, tmax = 0; int nwg = 0;
        std::map<int, double> tot;
        for (int wg = 0; wg < 4096; ++wg) {
            const long long* s = &h[(size_t)wg * 256];
            const int n = (int)s[255];
            if (n < 2) continue;
            ++nwg;
            const long long m = 0xFFFFFFFFFFFFll;
            if (tmin < 0 || (s[0] & m) < tmin) tmin = s[0] & m;
            tmax = std::max(tmax, s[n - 1] & m);
            for (int i = 1; i < n; ++i) tot[(int)(s[i] >> 48)] += (double)((s[i] & m) - (s[i - 1] & m));
        }
        printf("   %d workgroups stamped, first start -> last end %lld cycles; mean cycles per workgroup by phase (time spent reaching that stamp):\n", nwg, tmax - tmin);
        double all = 0;
        for (auto& kv : tot) all += kv.second;
        for (auto& kv : tot) printf("     %-16s %9.0f  (%4.1f %%)\n", kPhase[kv.first], kv.second / nwg, 100.0 * kv.second / all);
        printf("     total            %9.0f\n", all / nwg);
        for (int wg : {0, nwg / 2, nwg - 1}) {
            const long long* s = &h[(size_t)wg * 256];
            const int n = (int)s[255]; const long long m = 0xFFFFFFFFFFFFll;
            printf("   wg %d: start +%lld, end +%lld, stamps %d; first 40 deltas:", wg, (s[0] & m) - tmin, (s[n - 1] & m) - tmin, n);
            for (int i = 1; i < std::min(n, 41); ++i) printf(" %d:%lld", (int)(s[i] >> 48), (s[i] & m) - (s[i - 1] & m));
            printf("\n");
        }
        hipFree(in); hipFree(out); hipFree(w); hipFree(stats);
    }
    return 0;
}
