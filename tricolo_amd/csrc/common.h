// Shared device helpers for the TriCoLo gfx950 kernels (CDNA4 only: wave64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define TRI_OK 0
#define TRI_ERR_ARG (-1)
#define TRI_ERR_UNSUPPORTED (-2)

extern "C" void tri_set_error(const char* msg);
// Ablation bits of the timing probes (no MFMA / no epilogue / no statistics ...: WRONG results, timing only).  Compiled in only with
// -DTRI_PROBE_BUILD (tools/probes); the production library ignores TRICOLO_HALO_ABL, so an environment variable can never silently
// corrupt gradients.
static inline int tri_probe_ablation() {
#ifdef TRI_PROBE_BUILD
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("TRICOLO_HALO_ABL"); abl = e ? atoi(e) : 0; }
    return abl;
#else
    return 0;
#endif
}
int tri_check_launch(const char* what);

// Unsigned division by a runtime constant (Granlund-Montgomery round-up form), valid for n < 2^31.
struct FastDiv {
    uint32_t mul, shift, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1u << l) < d) ++l;
    f.shift = l;
    f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    return (__umulhi(f.mul, n) + n) >> f.shift;
}

// ---- LDS operand tiles --------------------------------------------------------------------------------------
// Every MFMA operand tile is [rows][32 k] bf16 = 64 B per row, read with ds_read_b128 as
// lane l -> row (l & 15), 16-byte chunk (l >> 4).  ds_read_b128 is serviced in the 16-lane groups
// {0-3,12-15,20-27} ... (MI355X_MICROARCH.md, LDS table); with 64-B rows the 16-byte slot of (row, pos) is
// (row & 3) * 4 + pos, so the four row-quads a group touches must land on four distinct positions.  Storing
// chunk c of row r at position c ^ f((r >> 2) & 3), f = {0,3,2,1}, makes all four lane groups conflict-free.
__device__ __forceinline__ int tile_off(int row, int chunk) {
    int g = (row >> 2) & 3;
    int f = g ^ ((g & 1) << 1);
    return row * 64 + ((chunk ^ f) << 4);
}

__device__ __forceinline__ void split_bf16(const float4& v, bf16x4& hi, bf16x4& lo) {
    hi[0] = (bf16_t)v.x; hi[1] = (bf16_t)v.y; hi[2] = (bf16_t)v.z; hi[3] = (bf16_t)v.w;
    lo[0] = (bf16_t)(v.x - (float)hi[0]); lo[1] = (bf16_t)(v.y - (float)hi[1]);
    lo[2] = (bf16_t)(v.z - (float)hi[2]); lo[3] = (bf16_t)(v.w - (float)hi[3]);
}
__device__ __forceinline__ bf16x4 to_bf16x4(const float4& v) {
    bf16x4 h;
    h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
    return h;
}

// ---- activation storage ---------------------------------------------------------------------------------------
// Big activation tensors are fp32 (bf16x3 mode), bf16 (bf16 mode) or f16 (f16 mode: the 16-bit modes move half the HBM
// and L2 bytes of every pass; f16 keeps 11 significand bits against bf16's 8, which is what puts the f16 mode inside the
// 1e-3 parity bound).  All arithmetic is fp32 either way; these helpers move 4 consecutive channels.
// The C ABI names the storage type by an int `act_fmt`: 0 fp32, 1 bf16, 2 f16.
#ifndef TRI_FMT_F32
#define TRI_FMT_F32 0
#define TRI_FMT_BF16 1
#define TRI_FMT_F16 2
#endif
// The fp32 value as a separately rounded result: without this the compiler may fold an fp32 FMA and the conversion to f16 that
// follows into one v_fma_mixlo_f16 (ONE rounding of the exact result) in one kernel and not in another that forms the same value -
// kernels that must agree bit for bit (tri_maxpool_bn_bwd_apply / conv_stem_wgrad_kernel's BNF staging) pin the double rounding.
__device__ __forceinline__ float fp32_rounded(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
template <typename T> struct Act;
template <> struct Act<float> {
    static constexpr int BYTES = 4;
    static constexpr int SIG_BITS = 24;                                        // significand bits incl. the hidden one
    static __device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
    static __device__ __forceinline__ void st4(float* p, const float4& v) { *(float4*)p = v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }           // value as it will be read back
};
template <> struct Act<bf16_t> {
    static constexpr int BYTES = 2;
    static constexpr int SIG_BITS = 8;
    static __device__ __forceinline__ float4 ld4(const bf16_t* p) {
        bf16x4 r = *(const bf16x4*)p;
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    }
    static __device__ __forceinline__ void st4(bf16_t* p, const float4& v) { *(bf16x4*)p = to_bf16x4(v); }
    static __device__ __forceinline__ float rnd(float v) { return (float)(bf16_t)v; }
};
template <> struct Act<f16_t> {
    static constexpr int BYTES = 2;
    static constexpr int SIG_BITS = 11;
    static __device__ __forceinline__ float4 ld4(const f16_t* p) {
        f16x4 r = *(const f16x4*)p;
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    }
    static __device__ __forceinline__ void st4(f16_t* p, const float4& v) {
        f16x4 h;
        h[0] = (f16_t)v.x; h[1] = (f16_t)v.y; h[2] = (f16_t)v.z; h[3] = (f16_t)v.w;
        *(f16x4*)p = h;
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(f16_t)v; }
};
// run `...` with T bound to the storage type that act_fmt names
#define TRI_ACT_DISPATCH(fmt, ...)                                       \
    do {                                                                 \
        if ((fmt) == TRI_FMT_BF16) { using T = bf16_t; __VA_ARGS__; }    \
        else if ((fmt) == TRI_FMT_F16) { using T = f16_t; __VA_ARGS__; } \
        else { using T = float; __VA_ARGS__; }                           \
    } while (0)

// ---- MFMA operand element types ---------------------------------------------------------------------------------
// E = bf16_t or f16_t: both issue v_mfma_f32_16x16x32_* at the same rate with fp32 accumulation.  fp32-stored
// activations are converted (and, in the 3-product mode, split) to bf16 operands; 16-bit storage IS the operand type.
template <typename E> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 v8;
    static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mma<f16_t> {
    typedef f16x8 v8;
    static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <typename AT> struct OpOf { typedef bf16_t E; };
template <> struct OpOf<f16_t> { typedef f16_t E; };

// ---- asynchronous global -> LDS copies --------------------------------------------------------------------------
// LDS-DMA issued through inline asm: the compiler then knows nothing about LDS being written asynchronously and does not
// put its own (conservative, vmcnt(0)) wait in front of every ds_read that follows - the counted waits + barriers of the
// kernel are the only synchronisation, which is what lets DMAs stay in flight across several compute units of work.
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i make_rsrc_words(const void* base, unsigned bytes) {
    return (v4i){(int)(unsigned)(size_t)base, (int)(((size_t)base >> 32) & 0xffff), (int)bytes, 0x00020000};
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
#else
    return 0;
#endif
}
// lds_dst must be wave-uniform (the hardware adds lane * 16).  M0 is named in the clobber list; clang treats it as a
// reserved register and only warns (-Wno-inline-asm in the Makefile) - the kernels that use this helper have no other
// M0 consumer (no movrel indexing, no compiler-issued LDS-DMA), and tests/test_gpu_ops.py checks them bit-exactly.
__device__ __forceinline__ void dma16_async(v4i rsrc, unsigned lds_dst, int voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "v"(voff), "s"(rsrc) : "memory", "m0");
#endif
}

// XCD-aware bijective remap (cdna_hip_programming.md T1): blocks that share operand panels get consecutive
// logical ids on one XCD, so the panel is fetched into that XCD's L2 once.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
