// Timing ablations of conv_halo_rows_kernel / conv_halo2d_kernel on the bench shape's 3x3 layers (results are WRONG in the ablated
// builds: timing only).  Build one binary per variant:
//   for v in BASE NOMMA NOREAD NODMA; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -DHR_ABL_$v -Iinclude \
//       -o tools/probes/build/halo_abl_$v tools/probes/halo_abl_probe.hip tricolo_amd/csrc/{misc,conv_c64,conv_vox,conv_pw}.hip; done
// Prints per layer the mean of 50 back-to-back launches (forward and data gradient).
#include "../../tricolo_amd/csrc/conv_igemm.hip"
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 192;
    const int first = argc > 2 ? atoi(argv[2]) : 0;
    struct L { int hw, c; } layers[] = {{32, 64}, {16, 128}, {8, 256}, {4, 512}};
    setvbuf(stdout, nullptr, _IONBF, 0);
    int li = -1;
    for (auto l : layers) {
        if (++li < first) continue;
        TriConvDesc d = {B, 1, l.hw, l.hw, l.c, 1, l.hw, l.hw, l.c, 1, 3, 3, 1, 0, 1, 1};
        const size_t M = (size_t)B * l.hw * l.hw, K = 9 * l.c;
        void *in, *w, *out; float* stats;
        hipMalloc(&in, M * l.c * 2); hipMalloc(&out, M * l.c * 2); hipMalloc(&w, K * l.c * 2);
        hipMemset(in, 0x11, M * l.c * 2); hipMemset(w, 0x11, K * l.c * 2);
        const int nt = tri_conv_num_mtiles(&d, 0);
        hipMalloc(&stats, (size_t)(nt + 1024) * 2 * l.c * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int dir = 0; dir < 2; ++dir) {
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                hipEventRecord(e0, 0);
                int rc = 0;
                for (int r = 0; r < 50 && !rc; ++r)
                    rc = dir == 0 ? tri_conv_fwd(&d, in, w, nullptr, out, nullptr, nullptr, 0, 0, stats, TRI_FMT_F16, nullptr, 0, nullptr, nullptr, nullptr)
                                  : tri_conv_dgrad(&d, out, w, nullptr, in, nullptr, 0, TRI_FMT_F16, nullptr, 0, nullptr, nullptr, nullptr);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                if (rc) { printf("launch failed: %d %s\n", rc, tri_last_error()); return 1; }
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%dx%d C=%d %s family %d: %.2f us per launch (50 back to back)\n", l.hw, l.hw, l.c, dir ? "dgrad" : "fwd  ",
                   tri_conv_kernel_family(&d, dir, 0), best * 1e3 / 50);
        }
        hipFree(in); hipFree(out); hipFree(w); hipFree(stats);
    }
    return 0;
}
