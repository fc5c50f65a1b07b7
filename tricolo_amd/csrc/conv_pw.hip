// conv_pw_kernel: the 1x1 / stride-2 shortcut convolutions of the ResNet trunk (torchvision BasicBlock.downsample[0], mv_cnn.py:44), forward
// and data gradient, 16-bit storage.  Through conv_dma_kernel (im2col gather, 64-channel k-steps, a pipeline built for 9 - 27 taps) these
// three layers ran at 45 - 89 TFLOP/s - 9 to 18 us for 0.8 GFLOP each, six launches per step; they are HBM / L2-bound GEMMs with a K of one
// to eight k-steps:
//   * the whole [BN x K] weight tile goes to LDS once per workgroup (16-byte chunks XOR-swizzled by the row), the activation rows go
//     straight from global memory into MFMA fragments (a row's 64 bytes per k-step are one contiguous segment; four k-steps in flight);
//   * forward: row m = output pixel (n, oh, ow), gathered from input pixel (n, 2 oh, 2 ow); BatchNorm column sums in the epilogue (one
//     record per 128-row tile, as the other kernels);
//   * data gradient: row m = dOut pixel (n, ih, iw) -> dIn pixel (n, 2 ih, 2 iw); the three other pixels of its 2 x 2 block get zeros from
//     the same workgroup, so the dense dIn tensor the 3x3 branch accumulates into is complete after one launch.
#include "common.h"
#include "../../include/tricolo_hip.h"
#include "conv_vox.h"

struct PwArgs {
    const void* in;
    const void* w;             // [N][Kpad] packed operand rows
    void* out;
    float* stats;              // forward only (may be NULL): [mtiles][2][N]
    int M, K, N, Kpad;         // GEMM rows (pixels of the SMALL grid), contraction, output channels
    int GH, GW;                // the small grid; the large one is 2 GH x 2 GW
    int transposed;
};

template <int N>
static __device__ __forceinline__ float pw_row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}

template <typename AT, int BN>
__global__ __launch_bounds__(256) void conv_pw_kernel(const PwArgs p) {
    typedef Mma<typename OpOf<AT>::E> MM;
    typedef typename MM::v8 v8;
    constexpr int TN = BN / 16, TM = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
    const int NT = p.N / BN;
    const int mtile = blockIdx.x / NT, ntile = blockIdx.x - mtile * NT;
    const int K = p.K, kchunks = K >> 3;

    {   // weight tile -> LDS
        const AT* w = (const AT*)p.w + (size_t)ntile * BN * p.Kpad;
        for (int e = t; e < BN * kchunks; e += 256) {
            const int n = e / kchunks, c = e - n * kchunks;
            const uint4 v = *(const uint4*)(w + (size_t)n * p.Kpad + c * 8);
            *(uint4*)(smem + ((size_t)n * kchunks + (c ^ (n & 7))) * 16) = v;
        }
    }
    const AT* src[TM];
    bool ok[TM];
    size_t opix[TM];                                                    // element offset of the row's output pixel
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = mtile * 128 + wave * 32 + a * 16 + fr;
        ok[a] = m < p.M;
        const int mm = ok[a] ? m : 0;
        const int x = mm % p.GW, q = mm / p.GW, y = q % p.GH, n = q / p.GH;
        const size_t big = (((size_t)n * 2 * p.GH + 2 * y) * 2 * p.GW + 2 * x);   // pixel (n, 2 y, 2 x) of the large grid
        if (!p.transposed) { src[a] = (const AT*)p.in + big * K; opix[a] = (size_t)mm * p.N; }
        else { src[a] = (const AT*)p.in + (size_t)mm * K; opix[a] = big * p.N; }
    }
    __syncthreads();

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 128) {
        uint4 af[4][TM];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                af[s][a] = make_uint4(0u, 0u, 0u, 0u);
                if (ok[a] && k0 + s * 32 < K) af[s][a] = *(const uint4*)(src[a] + k0 + s * 32 + fq * 8);
            }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (k0 + s * 32 >= K) break;
            const int c = ((k0 + s * 32) >> 3) + fq;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const uint4 bf = *(const uint4*)(smem + ((size_t)(b * 16 + fr) * kchunks + (c ^ (fr & 7))) * 16);
#pragma unroll
                for (int a = 0; a < TM; ++a) acc[a][b] = MM::mma(__builtin_bit_cast(v8, bf), __builtin_bit_cast(v8, af[s][a]), acc[a][b]);
            }
        }
    }

    AT* const out = (AT*)p.out + ntile * BN;
    if (p.transposed) {
        const size_t dx = p.N, dy = (size_t)2 * p.GW * p.N;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if (!ok[a]) continue;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                AT* o = out + opix[a] + b * 16 + fq * 4;
                const f32x4 v = acc[a][b];
                Act<AT>::st4(o, make_float4(v[0], v[1], v[2], v[3]));
                Act<AT>::st4(o + dx, z);
                Act<AT>::st4(o + dy, z);
                Act<AT>::st4(o + dy + dx, z);
            }
        }
        return;
    }
    f32x4 cs[TN], cq[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) { cs[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; cq[b] = cs[b]; }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        if (!ok[a]) continue;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            f32x4 v = acc[a][b];
            Act<AT>::st4(out + opix[a] + b * 16 + fq * 4, make_float4(v[0], v[1], v[2], v[3]));
            v[0] = Act<AT>::rnd(v[0]); v[1] = Act<AT>::rnd(v[1]); v[2] = Act<AT>::rnd(v[2]); v[3] = Act<AT>::rnd(v[3]);   // statistics of what BatchNorm reads back
            cs[b] += v;
            cq[b] += v * v;
        }
    }
    if (!p.stats) return;
    __syncthreads();                                                    // the weight tile is no longer read: its LDS holds the column sums
    float* const red = (float*)smem;                                    // [4 waves][BN][2]
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s_ = cs[b][r], q_ = cq[b][r];
            s_ += pw_row_ror<8>(s_); q_ += pw_row_ror<8>(q_);
            s_ += pw_row_ror<4>(s_); q_ += pw_row_ror<4>(q_);
            s_ += pw_row_ror<2>(s_); q_ += pw_row_ror<2>(q_);
            s_ += pw_row_ror<1>(s_); q_ += pw_row_ror<1>(q_);
            if (fr == 0) {
                const int col = b * 16 + fq * 4 + r;
                red[(wave * BN + col) * 2 + 0] = s_;
                red[(wave * BN + col) * 2 + 1] = q_;
            }
        }
    __syncthreads();
    if (t < BN) {
        float s_ = 0.f, q_ = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) { s_ += red[(w2 * BN + t) * 2]; q_ += red[(w2 * BN + t) * 2 + 1]; }
        p.stats[((size_t)mtile * 2 + 0) * p.N + ntile * BN + t] = s_;
        p.stats[((size_t)mtile * 2 + 1) * p.N + ntile * BN + t] = q_;
    }
}

static bool pw_disabled() {                                            // A/B switch: TRICOLO_NO_PW_CONV=1 keeps conv_dma_kernel for the 1x1 / 2 layers
    static int v = -1;
    if (v < 0) { const char* e = getenv("TRICOLO_NO_PW_CONV"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

// (arguments in the GEMM view of conv_make_plan: for the data gradient the "input" grid is dOut's and the "output" grid twice as large -
//  the direction is read off the two grids)
bool tri_internal_pw_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride, int pd,
                              int ph, int pw, TriPwGeom* g) {
    if (pw_disabled() || KD != 1 || KH != 1 || KW != 1 || stride != 2 || pd || ph || pw || ID != 1 || OD != 1) return false;
    if (cin % 64 || cin > 512 || cout % 64) return false;
    int transposed;
    if (IH == 2 * OH && IW == 2 * OW) transposed = 0;
    else if (OH == 2 * IH && OW == 2 * IW) transposed = 1;
    else return false;
    const int GH = transposed ? IH : OH, GW = transposed ? IW : OW;
    const long M = (long)B * GH * GW;
    if (M * 4 * (cin > cout ? cin : cout) * 2 >= ((long)1 << 31)) return false;
    const int mtiles = (int)((M + 127) / 128);
    const int bn = (cout % 128 == 0 && (long)128 * cin * 2 <= 65536 && (long)mtiles * (cout / 128) >= 256) ? 128 : 64;
    g->bn = bn; g->mtiles = mtiles; g->GH = GH; g->GW = GW; g->transposed = transposed;
    return true;
}

template <typename AT>
static int pw_launch_t(const TriPwGeom& g, const PwArgs& a, hipStream_t stream) {
    const size_t smem = (size_t)g.bn * a.K * 2 > (size_t)4 * g.bn * 2 * 4 ? (size_t)g.bn * a.K * 2 : (size_t)4 * g.bn * 2 * 4;
    const int grid = g.mtiles * (a.N / g.bn);
    if (g.bn == 128) {
        static bool set = false;
        if (!set) { hipFuncSetAttribute((const void*)conv_pw_kernel<AT, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); set = true; }
        conv_pw_kernel<AT, 128><<<grid, 256, smem, stream>>>(a);
    } else {
        static bool set = false;
        if (!set) { hipFuncSetAttribute((const void*)conv_pw_kernel<AT, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); set = true; }
        conv_pw_kernel<AT, 64><<<grid, 256, smem, stream>>>(a);
    }
    return tri_check_launch("tri_conv(pw)");
}

int tri_internal_pw_launch(const TriPwGeom& g, int B, const void* in, const void* w, int K, int N, int Kpad, void* out, float* stats, int transposed,
                           int act_fmt, hipStream_t stream) {
    PwArgs a{};
    a.in = in; a.w = w; a.out = out; a.stats = transposed ? nullptr : stats;
    a.M = B * g.GH * g.GW; a.K = K; a.N = N; a.Kpad = Kpad; a.GH = g.GH; a.GW = g.GW; a.transposed = transposed;
    return act_fmt == TRI_FMT_F16 ? pw_launch_t<f16_t>(g, a, stream) : pw_launch_t<bf16_t>(g, a, stream);
}
