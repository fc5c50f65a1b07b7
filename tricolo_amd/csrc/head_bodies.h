// Device bodies of the towers' head passes (global average pool + view max, row L2 normalise), shared by their own launches
// (bn_pool.hip, misc.hip) and by the one-launch head chains of linear_small.hip - same arithmetic, same summation order.
#pragma once
#include "common.h"

// Accesses to the tensors one stage of a head chain hands to the next INSIDE a launch (COH = true): agent-scope relaxed atomics, i.e.
// `global_load / global_store ... sc1` - they go through to memory, past the XCD's own L2, so producers and consumers on different
// XCDs agree without an L2 write-back / invalidate (an agent-scope fence writes back EVERYTHING dirty in the L2: measured 13-40 us per
// stage in the first version of the chain).  COH = false: the plain vector accesses of the single launches.
#ifndef HEAD_COH_SCOPE
#define HEAD_COH_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP                  /* EXPERIMENT: every chain workgroup on ONE XCD, exchange through its L2 (sc0) */
#endif
template <bool COH> __device__ __forceinline__ float ld1c(const float* p) {
    if constexpr (!COH) return *p;
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, HEAD_COH_SCOPE);
}
template <bool COH> __device__ __forceinline__ float4 ld4c(const float* p) {
    if constexpr (!COH) return *(const float4*)p;
    else return make_float4(ld1c<true>(p), ld1c<true>(p + 1), ld1c<true>(p + 2), ld1c<true>(p + 3));
}
template <bool COH> __device__ __forceinline__ void st1c(float* p, float v) {
    if constexpr (!COH) *p = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, HEAD_COH_SCOPE);
}
template <bool COH> __device__ __forceinline__ void st4c(float* p, const float4& v) {
    if constexpr (!COH) *(float4*)p = v;
    else { st1c<true>(p, v.x); st1c<true>(p + 1, v.y); st1c<true>(p + 2, v.z); st1c<true>(p + 3, v.w); }
}

// ----------------------------------------------------- global average pool + max over the views of one shape
// x [B*V, HW, C] -> out [B, C], argmax view index [B, C] (first maximum, as torch.max(dim=1)); mv_cnn.py:29-31
// work item = (shape b, 128 channels): 32 channel quads x 8 view slots; a thread sums its views over HW in the k order of a plain loop
// (same fp32 result as one thread per (b, c) - which ran 96 dependent-latency loads per thread on 64 workgroups, 25 us for 3 MB),
// then the 8 slots are merged through LDS with torch.max's tie rule (first maximum).  256 threads; ends on a barrier-free tail.
template <typename T, bool COH = false>
__device__ __forceinline__ void avgpool_viewmax_fwd_body(const T* __restrict__ x, int V, int HW, int C, float* __restrict__ out,
                                                         int* __restrict__ arg, int b, int cblock, float4 (*sb)[32], int4 (*si)[32]) {
    const int q = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const int c = cblock * 128 + q * 4;
    const float inv = 1.0f / (float)HW;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int4 bi = make_int4(0, 0, 0, 0);
    if (c < C) {
        for (int v = slot; v < V; v += 8) {
            const T* p = x + ((long)(b * V + v) * HW) * C + c;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8                                                   // (rolled: one load round trip per pixel, 16 in a row for a 4x4 map)
            for (int k = 0; k < HW; ++k) {
                const float4 e = Act<T>::ld4(p + (long)k * C);
                s.x += e.x; s.y += e.y; s.z += e.z; s.w += e.w;
            }
            s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
            if (s.x > best.x) { best.x = s.x; bi.x = v; }
            if (s.y > best.y) { best.y = s.y; bi.y = v; }
            if (s.z > best.z) { best.z = s.z; bi.z = v; }
            if (s.w > best.w) { best.w = s.w; bi.w = v; }
        }
    }
    sb[slot][q] = best; si[slot][q] = bi;
    __syncthreads();
    if (slot == 0 && c < C) {
#pragma unroll
        for (int s2 = 1; s2 < 8; ++s2) {
            const float4 o = sb[s2][q];
            const int4 oi = si[s2][q];
            // a slot's candidate is its FIRST maximum; between slots the smaller view index wins a tie
            if (o.x > best.x || (o.x == best.x && o.x > -INFINITY && oi.x < bi.x)) { best.x = o.x; bi.x = oi.x; }
            if (o.y > best.y || (o.y == best.y && o.y > -INFINITY && oi.y < bi.y)) { best.y = o.y; bi.y = oi.y; }
            if (o.z > best.z || (o.z == best.z && o.z > -INFINITY && oi.z < bi.z)) { best.z = o.z; bi.z = oi.z; }
            if (o.w > best.w || (o.w == best.w && o.w > -INFINITY && oi.w < bi.w)) { best.w = o.w; bi.w = oi.w; }
        }
        st4c<COH>(out + (long)b * C + c, best);
        *(int4*)(arg + (long)b * C + c) = bi;
    }
}
// backward: the winning view of (b, c) receives dout / HW at every pixel (times scale: the f16 mode's gradient scale), the others zero
template <typename T, bool COH = false>
__device__ __forceinline__ void avgpool_viewmax_bwd_body(const float* __restrict__ dout, const int* __restrict__ arg, int B, int V, int HW, int C,
                                                         T* __restrict__ dx, float scale, long first, long stride) {
    const long total = (long)B * V * HW * C;
    const float inv = scale / (float)HW;
    for (long i = first; i < total; i += stride) {
        int c = (int)(i % C);
        long r = i / C / HW;
        int v = (int)(r % V), b = (int)(r / V);
        dx[i] = (T)((arg[(long)b * C + c] == v) ? ld1c<COH>(dout + (long)b * C + c) * inv : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ row L2 normalise
// z = x / max(||x||, eps)   (F.normalize(dim=1), sparse_cnn.py:51, mv_cnn.py:33, bigru.py:18);  one wave per row
template <bool COH = false>
__device__ __forceinline__ void l2norm_fwd_body(const float* __restrict__ x, int D, float eps, float* __restrict__ z, float* __restrict__ norm,
                                                int row, int lane) {
    const float* p = x + (long)row * D;
    float s = 0.f;
    // (unrolled: rolled, each of the D / 64 steps of a row waits out its own load - 8 round trips per pass at D = 512 for a 64 KB tensor)
#pragma unroll 8
    for (int i = lane; i < D; i += 64) { const float e = ld1c<COH>(p + i); s += e * e; }
    s = wave_sum(s);
    float nrm = sqrtf(s);
    float inv = 1.0f / fmaxf(nrm, eps);
#pragma unroll 8
    for (int i = lane; i < D; i += 64) z[(long)row * D + i] = ld1c<COH>(p + i) * inv;
    if (lane == 0 && norm) norm[row] = nrm;
}
// dx = (dz - z * <dz, z>) / max(norm, eps)
template <bool COH = false>
__device__ __forceinline__ void l2norm_bwd_body(const float* __restrict__ z, const float* __restrict__ norm, const float* __restrict__ dz, int D,
                                                float eps, float* __restrict__ dx, int row, int lane) {
    const float* zp = z + (long)row * D;
    const float* dp = dz + (long)row * D;
    float s = 0.f;
#pragma unroll 8
    for (int i = lane; i < D; i += 64) s += zp[i] * dp[i];
    s = wave_sum(s);
    float inv = 1.0f / fmaxf(norm[row], eps);
#pragma unroll 8
    for (int i = lane; i < D; i += 64) st1c<COH>(dx + (long)row * D + i, (dp[i] - zp[i] * s) * inv);
}
