// Small kernels of the TriCoLo step (gfx950): error plumbing, layout converters (COO voxels -> dense masked grid,
// NCHW images -> NHWC4), row L2-normalise forward/backward, column sums (bias gradients), fused Adam.
#include "common.h"
#include "../../include/tricolo_hip.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";
extern "C" void tri_set_error(const char* msg) { strncpy(g_err, msg, sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
extern "C" const char* tri_last_error() { return g_err; }
int tri_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return TRI_OK;
}
extern "C" int tri_version() { return 1; }

static inline int ew_grid(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ------------------------------------------------------------------------------------------------ voxel scatter
// COO batch of data_module.py:52-64 (locs [n,4] = (b,i0,i1,i2), feats [n,3] in [0,1]) -> dense channels-last grid
// [B,V,V,V,4] = (r,g,b,0) and site mask [B,V,V,V].  Both outputs must be zero-filled first (tri_fill_zero).
template <typename T>
__global__ void voxel_scatter_kernel(const int* __restrict__ locs, const float* __restrict__ feats, int n, int B, int V,
                                     T* __restrict__ dense, uint8_t* __restrict__ mask) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = locs[i * 4], z = locs[i * 4 + 1], y = locs[i * 4 + 2], x = locs[i * 4 + 3];
    if ((unsigned)b >= (unsigned)B || (unsigned)z >= (unsigned)V || (unsigned)y >= (unsigned)V || (unsigned)x >= (unsigned)V) return;
    long pos = (((long)b * V + z) * V + y) * V + x;
    Act<T>::st4(dense + pos * 4, make_float4(feats[i * 3], feats[i * 3 + 1], feats[i * 3 + 2], 0.f));
    mask[pos] = 1;
}
extern "C" int tri_voxel_scatter(const int* locs, const float* feats, int n, int B, int V, void* dense, uint8_t* mask, int act_fmt,
                                 void* stream) {
    hipStream_t s = (hipStream_t)stream;
    size_t sites = (size_t)B * V * V * V;
    // the mask is zeroed up to its 32-byte padding (callers need not pre-fill anything); one fill when it directly follows `dense`
    const size_t dense_bytes = sites * 4 * (act_fmt ? 2 : 4), mask_bytes = (sites + 31) / 32 * 32;
    if ((char*)dense + dense_bytes == (char*)mask) hipMemsetAsync(dense, 0, dense_bytes + mask_bytes, s);
    else { hipMemsetAsync(dense, 0, dense_bytes, s); hipMemsetAsync(mask, 0, mask_bytes, s); }
    if (n > 0) TRI_ACT_DISPATCH(act_fmt, voxel_scatter_kernel<T><<<(n + 255) / 256, 256, 0, s>>>(locs, feats, n, B, V, (T*)dense, mask));
    return tri_check_launch("tri_voxel_scatter");
}

// Dense RGBA u8 grids (the dataset's on-disk format, general_dataset.py:47-51,92-93) straight to the tower's input: site
// active <=> alpha != 0, feats = RGB / 255.  Replaces the CPU nonzero / COO build + tri_voxel_scatter with one coalesced
// pass: 4 plane bytes in, one (r,g,b,0) quad + one mask byte out per voxel; 4x less H2D than the f32 COO features.
template <typename T>
__global__ void voxel_from_rgba_kernel(const uint8_t* __restrict__ rgba, long V3, long total, T* __restrict__ dense,
                                       uint8_t* __restrict__ mask) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / V3, p = i - b * V3;
        const uint8_t* g = rgba + b * 4 * V3 + p;
        const bool on = g[3 * V3] != 0;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) v = make_float4((float)g[0] / 255.f, (float)g[V3] / 255.f, (float)g[2 * V3] / 255.f, 0.f);
        Act<T>::st4(dense + i * 4, v);
        mask[i] = on ? 1 : 0;
    }
}
extern "C" int tri_voxel_from_rgba_u8(const uint8_t* rgba, int B, int V, void* dense, uint8_t* mask, int act_fmt, void* stream) {
    const long V3 = (long)V * V * V, total = (long)B * V3;
    TRI_ACT_DISPATCH(act_fmt, voxel_from_rgba_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(rgba, V3, total, (T*)dense, mask));
    if (total % 32) hipMemsetAsync(mask + total, 0, 32 - total % 32, (hipStream_t)stream);      // padding bytes of the mask
    return tri_check_launch("tri_voxel_from_rgba_u8");
}

// Step timeline (tools/step_timeline.py): one lane writes the constant 100 MHz wall clock into *slot.  Launched between the
// kernels of a (captured) step it timestamps the critical path without a profiler attached - rocprofv3's kernel trace inflates
// and re-orders the gaps between the ~330 launches of a step.
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
extern "C" int tri_debug_stamp(unsigned long long* slot, void* stream) {
    stamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(slot);
    return tri_check_launch("tri_debug_stamp");
}

// Toolchain regression probe (round 6): the 16-byte raw-buffer builtins in the forms the conv kernels use them
// (form 0: load with a per-lane offset, scalar offset 0 - the 16 load sites of conv_igemm / conv_vox / conv_voxg / conv_c64 / conv_s2g /
// conv_wgrad) and in the forms that MISCOMPILED on ROCm 7.2 when gru.hip tried them (profiles/r5/NOTES_voxel.md: store with an SGPR scalar
// offset - data registers rewritten 0-3 instructions behind the store; load with a scalar offset the compiler did not see as uniform -
// split into four dword loads, the fourth returning the first's value).  Every form copies src to dst, 16 bytes per lane, four
// back-to-back elements per thread so that the data registers are reused right behind each access; tests/test_gpu_ops.py compares.
//   form 0  load (voffset) -> plain store        form 1  load (voffset + SGPR soffset) -> plain store
//   form 2  plain load -> store (voffset)        form 3  plain load -> store (voffset + SGPR soffset)
__global__ __launch_bounds__(256) void buffer_b128_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16, int form) {
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(n16 * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)(n16 * 16), 0x00020000);
    const long tile = (long)blockIdx.x * 1024;                       // 1024 elements per workgroup: 4 per thread, 256 apart
    const unsigned sbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(tile * 16));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = tile + u * 256 + threadIdx.x;
        if (i >= n16) continue;
        const unsigned lane_off = (unsigned)((u * 256 + threadIdx.x) * 16);
        uint4 v;
        if (form == 0) v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srs, (unsigned)(i * 16), 0, 0));
        else if (form == 1) v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srs, lane_off, sbase, 0));
        else v = src[i];
        if (form == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), drs, (unsigned)(i * 16), 0, 0);
        else if (form == 3) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), drs, lane_off, sbase, 0);
        else dst[i] = v;
    }
}
extern "C" int tri_debug_buffer_b128_probe(const void* src, void* dst, long n16, int form, void* stream) {
    if (form < 0 || form > 3 || n16 <= 0 || n16 * 16 >= ((long)1 << 31)) { tri_set_error("tri_debug_buffer_b128_probe: form 0..3, 0 < n16 * 16 < 2^31"); return TRI_ERR_ARG; }
    buffer_b128_probe_kernel<<<(unsigned)((n16 + 1023) / 1024), 256, 0, (hipStream_t)stream>>>((const uint4*)src, (uint4*)dst, n16, form);
    return tri_check_launch("tri_debug_buffer_b128_probe");
}

// count of non-zero mask bytes -> *count (device int)
__global__ void mask_count_kernel(const uint8_t* __restrict__ mask, long n, int* __restrict__ count) {
    int local = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) local += mask[i] ? 1 : 0;
    float f = wave_sum((float)local);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = f;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(count, (int)(ws[0] + ws[1] + ws[2] + ws[3] + 0.5f));
}
extern "C" int tri_mask_count(const uint8_t* mask, long n, int* count, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipMemsetAsync(count, 0, sizeof(int), s);
    long b = (n + 256 * 64 - 1) / (256 * 64);
    mask_count_kernel<<<(int)(b < 1 ? 1 : (b > 256 ? 256 : b)), 256, 0, s>>>(mask, n, count);
    return tri_check_launch("tri_mask_count");
}

// Active-site list of a level: row_pos[0 .. count) = the positions p with mask[p] != 0 in ASCENDING order (deterministic),
// *count = their number.  The submanifold convolutions then visit only these rows (tri_conv_fwd / tri_conv_dgrad with
// row_pos + row_count): executed work = active work, instead of every 128-site tile that contains one active site.
// Three small launches: per-block counts (2,048 sites per block), one-block exclusive scan, ordered write.
#define CMP_SITES 2048
__device__ __forceinline__ unsigned cmp_load8(const uint8_t* __restrict__ mask, long base, long n, unsigned long long* bits) {
    unsigned long long v = 0;
    if (base + 8 <= n) v = *(const unsigned long long*)(mask + base);            // mask buffers are 32-byte padded and aligned
    else
        for (int k = 0; k < 8; ++k)
            if (base + k < n) v |= (unsigned long long)mask[base + k] << (8 * k);
    unsigned cnt = 0;
    unsigned long long nz = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if ((v >> (8 * k)) & 0xff) { ++cnt; nz |= 1ull << k; }
    *bits = nz;
    return cnt;
}
__global__ __launch_bounds__(256) void mask_block_count_kernel(const uint8_t* __restrict__ mask, long n, int* __restrict__ block_count) {
    __shared__ int ws[4];
    unsigned long long bits;
    int c = (int)cmp_load8(mask, (long)blockIdx.x * CMP_SITES + threadIdx.x * 8, n, &bits);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// exclusive scan of block_count[0 .. nb) in place; *count = total
__global__ __launch_bounds__(1024) void mask_scan_kernel(int* __restrict__ block_count, int nb, int* __restrict__ count) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int b0 = t * per, b1 = min(nb, b0 + per);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += block_count[b];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                         // Hillis-Steele inclusive scan over the 1,024 partial sums
        int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int b = b0; b < b1; ++b) { int c = block_count[b]; block_count[b] = run; run += c; }
    if (t == 1023) *count = part[1023];
}
__global__ __launch_bounds__(256) void mask_block_write_kernel(const uint8_t* __restrict__ mask, long n, const int* __restrict__ block_base,
                                                               int* __restrict__ row_pos) {
    __shared__ int ws[4];
    unsigned long long bits;
    const long base = (long)blockIdx.x * CMP_SITES + threadIdx.x * 8;
    const int c = (int)cmp_load8(mask, base, n, &bits);
    int incl = c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    int off = block_base[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) off += ws[w];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if ((bits >> k) & 1ull) row_pos[off++] = (int)(base + k);
}
// mask_block_write_kernel with the scan folded in, for lists of at most 1,024 blocks (2 M sites): every block sums the counts of the
// blocks before it itself (<= 4 KiB of L2-resident reads) and the last block also writes the total - two launches instead of three
// on a chain of ~15 short launches per voxel level.
__global__ __launch_bounds__(256) void mask_block_write_scan_kernel(const uint8_t* __restrict__ mask, long n, const int* __restrict__ block_count,
                                                                    int* __restrict__ row_pos, int* __restrict__ count) {
    __shared__ int ws[4], bs[4];
    unsigned long long bits;
    const long base = (long)blockIdx.x * CMP_SITES + threadIdx.x * 8;
    const int c = (int)cmp_load8(mask, base, n, &bits);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pre = 0;                                                     // counts of the blocks before this one, 256 threads striding
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) pre += block_count[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) ws[wave] = incl;
    if (lane == 0) bs[wave] = pre;
    __syncthreads();
    const int block_base = bs[0] + bs[1] + bs[2] + bs[3];
    int off = block_base + incl - c;
    for (int w = 0; w < wave; ++w) off += ws[w];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if ((bits >> k) & 1ull) row_pos[off++] = (int)(base + k);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *count = block_base + ws[0] + ws[1] + ws[2] + ws[3];
}
extern "C" size_t tri_mask_compact_scratch(long n) { return (size_t)((n + CMP_SITES - 1) / CMP_SITES + 1) * sizeof(int); }
extern "C" int tri_mask_compact(const uint8_t* mask, long n, int* row_pos, int* count, void* scratch, void* stream) {
    if (n < 1 || n >= ((long)1 << 31)) { tri_set_error("tri_mask_compact: 1 <= n < 2^31 sites"); return TRI_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)((n + CMP_SITES - 1) / CMP_SITES);
    mask_block_count_kernel<<<nb, 256, 0, s>>>(mask, n, (int*)scratch);
    static int fused = -1;                                           // A/B switch: TRICOLO_MASK_SCAN_FUSED=0 keeps the separate scan launch
    if (fused < 0) fused = 1;
    if (fused && nb <= 1024) {
        mask_block_write_scan_kernel<<<nb, 256, 0, s>>>(mask, n, (const int*)scratch, row_pos, count);
        return tri_check_launch("tri_mask_compact");
    }
    mask_scan_kernel<<<1, 1024, 0, s>>>((int*)scratch, nb, count);
    mask_block_write_kernel<<<nb, 256, 0, s>>>(mask, n, (const int*)scratch, row_pos);
    return tri_check_launch("tri_mask_compact");
}

// (round 5) All site masks of the voxel tower and all their row lists up front.  The occupancy of level l + 1 is the 2x2x2 OR-pool of level l
// and depends on nothing but the input grid, so the five masks and the five compact lists need not be produced level by level between the
// convolutions (10 short launches on the tower's forward chain, ~3 us each as nodes of the replayed graph): one launch builds the masks of
// levels 1-4 from level 0 (one workgroup per level-4 site = 16^3 level-0 sites), two launches compact all levels together.
__global__ __launch_bounds__(512) void mask_pyramid_kernel(const uint8_t* __restrict__ m0, int B, int V, uint8_t* __restrict__ m1,
                                                           uint8_t* __restrict__ m2, uint8_t* __restrict__ m3, uint8_t* __restrict__ m4) {
    __shared__ uint8_t s1[512], s2[64], s3[8];
    const int t = threadIdx.x, V1 = V >> 1, V2 = V >> 2, V3 = V >> 3, V4 = V >> 4;
    int r = blockIdx.x;
    const int cx = r % V4; r /= V4;
    const int cy = r % V4; r /= V4;
    const int cz = r % V4, b = r / V4;
    {
        const int z = cz * 8 + (t >> 6), y = cy * 8 + ((t >> 3) & 7), x = cx * 8 + (t & 7);
        const size_t base = (((size_t)b * V + z * 2) * V + y * 2) * V + x * 2;
        const unsigned any = *(const unsigned short*)(m0 + base) | *(const unsigned short*)(m0 + base + V) |
                             *(const unsigned short*)(m0 + base + (size_t)V * V) | *(const unsigned short*)(m0 + base + (size_t)V * V + V);
        const uint8_t v = any ? 1 : 0;
        s1[t] = v;
        m1[(((size_t)b * V1 + z) * V1 + y) * V1 + x] = v;
    }
    __syncthreads();
    if (t < 64) {
        const int lz = t >> 4, ly = (t >> 2) & 3, lx = t & 3;
        unsigned any = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) any |= s1[((lz * 2 + (k >> 2)) * 8 + ly * 2 + ((k >> 1) & 1)) * 8 + lx * 2 + (k & 1)];
        s2[t] = (uint8_t)any;
        m2[(((size_t)b * V2 + cz * 4 + lz) * V2 + cy * 4 + ly) * V2 + cx * 4 + lx] = (uint8_t)any;
    }
    __syncthreads();
    if (t < 8) {
        const int lz = t >> 2, ly = (t >> 1) & 1, lx = t & 1;
        unsigned any = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) any |= s2[((lz * 2 + (k >> 2)) * 4 + ly * 2 + ((k >> 1) & 1)) * 4 + lx * 2 + (k & 1)];
        s3[t] = (uint8_t)any;
        m3[(((size_t)b * V3 + cz * 2 + lz) * V3 + cy * 2 + ly) * V3 + cx * 2 + lx] = (uint8_t)any;
    }
    __syncthreads();
    if (t == 0) {
        unsigned any = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) any |= s3[k];
        m4[(((size_t)b * V4 + cz) * V4 + cy) * V4 + cx] = (uint8_t)any;
    }
    if (blockIdx.x == 0 && t < 32) {                                // padding of every mask to 32 bytes (the brick kernels read whole dwords)
        const size_t n[4] = {(size_t)B * V1 * V1 * V1, (size_t)B * V2 * V2 * V2, (size_t)B * V3 * V3 * V3, (size_t)B * V4 * V4 * V4};
        uint8_t* m[4] = {m1, m2, m3, m4};
#pragma unroll
        for (int l = 0; l < 4; ++l)
            if (n[l] + t < (n[l] + 31) / 32 * 32) m[l][n[l] + t] = 0;
    }
}
// masks[0 .. 3] <- levels 1-4 of the level-0 mask of [B, V, V, V] grids; V % 16 == 0; every output holds its site count rounded up to 32 bytes
extern "C" int tri_mask_pyramid(const uint8_t* mask0, int B, int V, uint8_t* const* masks, void* stream) {
    if (B < 1 || V < 16 || V % 16 || ((uintptr_t)mask0 & 1)) { tri_set_error("tri_mask_pyramid: V must be a multiple of 16, mask 2-byte aligned"); return TRI_ERR_ARG; }
    const long blocks = (long)B * (V / 16) * (V / 16) * (V / 16);
    if (blocks >= (1L << 31)) { tri_set_error("tri_mask_pyramid: too many level-4 sites"); return TRI_ERR_ARG; }
    mask_pyramid_kernel<<<(int)blocks, 512, 0, (hipStream_t)stream>>>(mask0, B, V, masks[0], masks[1], masks[2], masks[3]);
    return tri_check_launch("tri_mask_pyramid");
}
#define CMP_MAX_LISTS 8
struct CompactJobs {
    const uint8_t* mask[CMP_MAX_LISTS];
    long n[CMP_MAX_LISTS];
    int* rows[CMP_MAX_LISTS];
    int* count[CMP_MAX_LISTS];
    int blk0[CMP_MAX_LISTS + 1];                                     // first block of every list in the launch
    int nlist;
};
template <typename T, int N>
static __device__ __forceinline__ T cmp_pick(const T (&a)[N], int l) {      // a[l] without a run-time index into the kernel-argument struct
    T v = a[0];
#pragma unroll
    for (int i = 1; i < N; ++i) if (l == i) v = a[i];
    return v;
}
static __device__ __forceinline__ int compact_list_of(const CompactJobs& j, int blk) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < CMP_MAX_LISTS; ++i) if (i < j.nlist && blk >= j.blk0[i]) l = i;
    return l;
}
__global__ __launch_bounds__(256) void mask_multi_count_kernel(CompactJobs j, int* __restrict__ block_count) {
    __shared__ int ws[4];
    const int l = compact_list_of(j, blockIdx.x);
    unsigned long long bits;
    int c = (int)cmp_load8(cmp_pick(j.mask, l), (long)(blockIdx.x - cmp_pick(j.blk0, l)) * CMP_SITES + threadIdx.x * 8, cmp_pick(j.n, l), &bits);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void mask_multi_write_scan_kernel(CompactJobs j, const int* __restrict__ block_count) {
    __shared__ int ws[4], bs[4];
    const int l = compact_list_of(j, blockIdx.x), first = cmp_pick(j.blk0, l), lb = blockIdx.x - first;
    unsigned long long bits;
    const long base = (long)lb * CMP_SITES + threadIdx.x * 8;
    const int c = (int)cmp_load8(cmp_pick(j.mask, l), base, cmp_pick(j.n, l), &bits);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pre = 0;                                                     // counts of the list's blocks before this one (as mask_block_write_scan_kernel)
    for (int b = threadIdx.x; b < lb; b += 256) pre += block_count[first + b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) ws[wave] = incl;
    if (lane == 0) bs[wave] = pre;
    __syncthreads();
    const int block_base = bs[0] + bs[1] + bs[2] + bs[3];
    int off = block_base + incl - c;
    for (int w = 0; w < wave; ++w) off += ws[w];
    int* __restrict__ row_pos = cmp_pick(j.rows, l);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if ((bits >> k) & 1ull) row_pos[off++] = (int)(base + k);
    if ((int)blockIdx.x == cmp_pick(j.blk0, l + 1) - 1 && threadIdx.x == 0) *cmp_pick(j.count, l) = block_base + ws[0] + ws[1] + ws[2] + ws[3];
}
// tri_mask_compact of up to 8 masks in two launches (every list at most 1,024 blocks = 2 M sites; larger ones go through tri_mask_compact).
// scratch: tri_mask_compact_multi_scratch(n, nlist) bytes.  Same lists, same counts.
extern "C" size_t tri_mask_compact_multi_scratch(const long* n, int nlist) {
    size_t nb = 0;
    for (int l = 0; l < nlist; ++l) nb += (size_t)((n[l] + CMP_SITES - 1) / CMP_SITES);
    return (nb + 1) * sizeof(int);
}
extern "C" int tri_mask_compact_multi(const uint8_t* const* masks, const long* n, int nlist, int* const* rows, int* const* counts, void* scratch,
                                      void* stream) {
    if (nlist < 1 || nlist > CMP_MAX_LISTS) { tri_set_error("tri_mask_compact_multi: 1..8 lists"); return TRI_ERR_ARG; }
    CompactJobs j;
    int nb = 0;
    for (int l = 0; l < nlist; ++l) {
        const long b = (n[l] + CMP_SITES - 1) / CMP_SITES;
        if (n[l] < 1 || b > 1024) { tri_set_error("tri_mask_compact_multi: every list needs 1 <= n <= 2,097,152 sites"); return TRI_ERR_ARG; }
        j.mask[l] = masks[l]; j.n[l] = n[l]; j.rows[l] = rows[l]; j.count[l] = counts[l]; j.blk0[l] = nb;
        nb += (int)b;
    }
    for (int l = nlist; l <= CMP_MAX_LISTS; ++l) j.blk0[l] = nb;
    j.nlist = nlist;
    hipStream_t s = (hipStream_t)stream;
    mask_multi_count_kernel<<<nb, 256, 0, s>>>(j, (int*)scratch);
    mask_multi_write_scan_kernel<<<nb, 256, 0, s>>>(j, (const int*)scratch);
    return tri_check_launch("tri_mask_compact_multi");
}

// ------------------------------------------------------------------------------------------------ token embedding
// emb[l][b][:] = W[tok[b][l]][:]   (bigru.py:15: embedding_layer(x).transpose(0, 1); padding row 0 of W is zero by init)
__global__ void embedding_fwd_kernel(const int* __restrict__ tok, const float4* __restrict__ w, int B, int L, int D4,
                                     float4* __restrict__ out) {
    const long total = (long)B * L * D4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D4);
        const long r = i / D4;
        const int l = (int)(r / B), b = (int)(r - (long)l * B);
        out[i] = w[(long)tok[b * L + l] * D4 + d];
    }
}
extern "C" int tri_embedding_fwd(const int* tokens, const float* weight, int B, int L, int D, float* out, void* stream) {
    if (D % 4) { tri_set_error("tri_embedding_fwd: D must be a multiple of 4"); return TRI_ERR_ARG; }
    embedding_fwd_kernel<<<ew_grid((long)B * L * (D / 4)), 256, 0, (hipStream_t)stream>>>(tokens, (const float4*)weight, B, L, D / 4,
                                                                                         (float4*)out);
    return tri_check_launch("tri_embedding_fwd");
}
// dW[v][:] = sum over the occurrences (l, b) of token v, in ascending (l, b) order, of dout[l][b][:]; dW[padding_idx] = 0.
// One workgroup per vocabulary row: the 256 threads mark the row's occurrences in an LDS bit mask (time-major positions), then every thread
// walks the (few) set bits in ascending order and accumulates its own column - deterministic, no global atomics, no sort (ATen's
// sort-based embedding_dense_backward took 143 us for 3,072 tokens).
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int* __restrict__ tok, const float* __restrict__ dout, int B, int L, int D,
                                                            int padding_idx, float* __restrict__ dw) {
    extern __shared__ unsigned long long masks[];                // [ceil(B*L / 64)]
    const int v = blockIdx.x, t = threadIdx.x, n = B * L;
    const int nm = (n + 63) >> 6;
    // The token list is scanned in MEMORY order (coalesced; eight loads in flight per thread) and a hit sets its bit at the time-major
    // position l * B + b with an LDS atomic - hits are rare (n / V per workgroup), the bit mask does not depend on their order.  Scanned in
    // time-major order with a ballot per 256 positions, every wave load touched 64 cache lines (stride L), in every one of the V
    // workgroups: 22 M line requests, 52 us at 6,144 tokens - the tail of the text tower and, in the voxel + text configurations, of the step.
    for (int w = t; w < nm; w += 256) masks[w] = 0ull;
    __syncthreads();
    for (int base = 0; base < n; base += 256 * 8) {
        int tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = base + u * 256 + t;                     // memory order: r = b * L + l
            tv[u] = r < n ? tok[r] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (tv[u] == v) {
                const int r = base + u * 256 + t;
                const int b = r / L, l = r - b * L;
                const int pos = l * B + b;
                atomicOr(&masks[pos >> 6], 1ull << (pos & 63));
            }
    }
    __syncthreads();
    for (int d = t; d < D; d += 256) {
        float acc = 0.f;
        if (v != padding_idx)
            for (int w = 0; w < nm; ++w) {
                unsigned long long m = masks[w];
                while (m) {
                    const int bit = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    acc += dout[(size_t)(w * 64 + bit) * D + d];
                }
            }
        dw[(size_t)v * D + d] = acc;
    }
}
extern "C" int tri_embedding_bwd(const int* tokens, const float* dout, int B, int L, int V, int D, int padding_idx, float* dweight,
                                 void* stream) {
    size_t smem = (size_t)((B * L + 63) / 64) * 8;
    if (smem > 60 * 1024) { tri_set_error("tri_embedding_bwd: more than 491,520 tokens per call"); return TRI_ERR_UNSUPPORTED; }
    embedding_bwd_kernel<<<V, 256, smem, (hipStream_t)stream>>>(tokens, dout, B, L, D, padding_idx, dweight);
    return tri_check_launch("tri_embedding_bwd");
}

// ------------------------------------------------------------------------------------------------ image layout
// x [N,3,H,W] f32 (tricolo_net.py:51 flatten of data_dict["images"]) -> [N,H,W,4] with a zero 4th channel
template <typename T>
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ x, long HW, long total, T* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long n = i / HW, p = i - n * HW;
        const float* b = x + n * 3 * HW + p;
        Act<T>::st4(out + i * 4, make_float4(b[0], b[HW], b[2 * HW], 0.f));
    }
}
// four pixels per thread: three 16-byte plane loads, 32 (16-bit storage) / 64 contiguous output bytes - the one-pixel form above moved
// 20 bytes per thread and ran at 2 TB/s (32 us for the bench batch, at the head of the image tower's chain)
template <typename T>
__global__ void nchw3_to_nhwc4_x4_kernel(const float* __restrict__ x, long HW4, long total4, T* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW4, p = i - n * HW4;
        const float4* b = (const float4*)(x + n * 3 * HW4 * 4) + p;
        const float4 r = b[0], g = b[HW4], bl = b[2 * HW4];
        T* o = out + i * 16;
        Act<T>::st4(o, make_float4(r.x, g.x, bl.x, 0.f));
        Act<T>::st4(o + 4, make_float4(r.y, g.y, bl.y, 0.f));
        Act<T>::st4(o + 8, make_float4(r.z, g.z, bl.z, 0.f));
        Act<T>::st4(o + 12, make_float4(r.w, g.w, bl.w, 0.f));
    }
}
extern "C" int tri_nchw3_to_nhwc4(const float* x, int N, int H, int W, void* out, int act_fmt, void* stream) {
    long HW = (long)H * W, total = (long)N * HW;
    if (HW % 4 == 0 && ((size_t)x & 15) == 0) {
        TRI_ACT_DISPATCH(act_fmt, nchw3_to_nhwc4_x4_kernel<T><<<ew_grid(total / 4), 256, 0, (hipStream_t)stream>>>(x, HW / 4, total / 4, (T*)out));
        return tri_check_launch("tri_nchw3_to_nhwc4");
    }
    TRI_ACT_DISPATCH(act_fmt, nchw3_to_nhwc4_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(x, HW, total, (T*)out));
    return tri_check_launch("tri_nchw3_to_nhwc4");
}

// u8 [N,3,H,W] -> [N,H,W,4] normalised exactly like torchvision's ToTensor + Normalize in general_dataset.py:87-89:
// (u8 / 255 - mean[c]) / std[c] in fp32, zero 4th channel.  The f32 image batch never exists (4x less H2D and HBM).
template <typename T>
__global__ void nchw3_u8_to_nhwc4_kernel(const uint8_t* __restrict__ x, long HW, long total, float m0, float m1, float m2, float s0,
                                         float s1, float s2, T* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long n = i / HW, p = i - n * HW;
        const uint8_t* b = x + n * 3 * HW + p;
        Act<T>::st4(out + i * 4, make_float4(((float)b[0] / 255.f - m0) / s0, ((float)b[HW] / 255.f - m1) / s1,
                                             ((float)b[2 * HW] / 255.f - m2) / s2, 0.f));
    }
}
extern "C" int tri_nchw3_u8_to_nhwc4(const uint8_t* x, int N, int H, int W, const float* mean3, const float* std3, void* out, int act_fmt,
                                     void* stream) {
    long HW = (long)H * W, total = (long)N * HW;
    TRI_ACT_DISPATCH(act_fmt, nchw3_u8_to_nhwc4_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(
        x, HW, total, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (T*)out));
    return tri_check_launch("tri_nchw3_u8_to_nhwc4");
}

// ------------------------------------------------------------------------------------------------ row L2 normalise
// z = x / max(||x||, eps)   (F.normalize(dim=1), sparse_cnn.py:51, mv_cnn.py:33, bigru.py:18);  one wave per row
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, int rows, int D, float eps, float* __restrict__ z, float* __restrict__ norm) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (long)row * D;
    float s = 0.f;
    // (unrolled: rolled, each of the D / 64 steps of a row waits out its own load - 8 round trips per pass at D = 512 for a 64 KB tensor)
#pragma unroll 8
    for (int i = lane; i < D; i += 64) s += p[i] * p[i];
    s = wave_sum(s);
    float nrm = sqrtf(s);
    float inv = 1.0f / fmaxf(nrm, eps);
#pragma unroll 8
    for (int i = lane; i < D; i += 64) z[(long)row * D + i] = p[i] * inv;
    if (lane == 0 && norm) norm[row] = nrm;
}
extern "C" int tri_l2norm_fwd(const float* x, int rows, int D, float eps, float* z, float* norm, void* stream) {
    l2norm_fwd_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, rows, D, eps, z, norm);
    return tri_check_launch("tri_l2norm_fwd");
}
// dx = (dz - z * <dz, z>) / max(norm, eps)
__global__ void l2norm_bwd_kernel(const float* __restrict__ z, const float* __restrict__ norm, const float* __restrict__ dz, int rows,
                                  int D, float eps, float* __restrict__ dx) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* zp = z + (long)row * D;
    const float* dp = dz + (long)row * D;
    float s = 0.f;
#pragma unroll 8
    for (int i = lane; i < D; i += 64) s += zp[i] * dp[i];
    s = wave_sum(s);
    float inv = 1.0f / fmaxf(norm[row], eps);
#pragma unroll 8
    for (int i = lane; i < D; i += 64) dx[(long)row * D + i] = (dp[i] - zp[i] * s) * inv;
}
extern "C" int tri_l2norm_bwd(const float* z, const float* norm, const float* dz, int rows, int D, float eps, float* dx, void* stream) {
    l2norm_bwd_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(z, norm, dz, rows, D, eps, dx);
    return tri_check_launch("tri_l2norm_bwd");
}

// ------------------------------------------------------------------------------------------------ column sums
// out[c] = sum_m g[m, c]   (bias gradients); one block per 64 columns, fixed summation order
__global__ void colsum_kernel(const float* __restrict__ g, long M, int C, float* __restrict__ out) {
    __shared__ float sh[4][64];
    int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (long m = r; m < M; m += 4) s += g[m * C + c];
    sh[r][threadIdx.x & 63] = s;
    __syncthreads();
    if (r == 0 && c < C) out[c] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}
extern "C" int tri_colsum(const float* g, long M, int C, float* out, void* stream) {
    colsum_kernel<<<(C + 63) / 64, 256, 0, (hipStream_t)stream>>>(g, M, C, out);
    return tri_check_launch("tri_colsum");
}

// elementwise helpers on flat fp32 buffers
__global__ void axpy_kernel(const float* __restrict__ x, float a, float* y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += a * x[i];
}
extern "C" int tri_axpy(const float* x, float a, float* y, long n, void* stream) {
    axpy_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(x, a, y, n);
    return tri_check_launch("tri_axpy");
}
__global__ void act_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = x[i];
        y[i] = act == 1 ? fmaxf(v, 0.f) : (act == 2 ? tanhf(v) : v);
    }
}
// g = dout * act'(out):  relu: out > 0;  tanh: 1 - out^2
__global__ void act_bwd_kernel(const float* dout, const float* __restrict__ out, float* g, long n, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float o = out[i], d = dout[i];
        g[i] = act == 1 ? (o > 0.f ? d : 0.f) : (act == 2 ? d * (1.f - o * o) : d);
    }
}
extern "C" int tri_act_bwd(const float* dout, const float* out, float* g, long n, int act, void* stream) {
    act_bwd_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(dout, out, g, n, act);
    return tri_check_launch("tri_act_bwd");
}

// Up to 8 contiguous fp32 segments copied in one launch (blockIdx.y = segment): the per-step concatenations of the nn.GRU
// parameters ([w_ih_f; w_ih_r], [b_ih_f; b_ih_r], stack(w_hh), stack(b_hh)) without four ATen cat kernels.
struct SegCopy { const float* src[8]; float* dst[8]; long n[8]; };
__global__ void seg_copy_kernel(SegCopy s) {
    const float* src = s.src[blockIdx.y];
    float* dst = s.dst[blockIdx.y];
    const long n = s.n[blockIdx.y];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
// srcs / dsts / n are HOST arrays of `count` (<= 8) device pointers / element counts
extern "C" int tri_copy_segments(const void* const* srcs, void* const* dsts, const long* n, int count, void* stream) {
    if (count < 1 || count > 8) { tri_set_error("tri_copy_segments: 1..8 segments"); return TRI_ERR_ARG; }
    SegCopy s{};
    long nmax = 0;
    for (int i = 0; i < count; ++i) { s.src[i] = (const float*)srcs[i]; s.dst[i] = (float*)dsts[i]; s.n[i] = n[i]; if (n[i] > nmax) nmax = n[i]; }
    int gx = (int)((nmax + 255) / 256);
    seg_copy_kernel<<<dim3(gx < 1 ? 1 : (gx > 256 ? 256 : gx), count), 256, 0, (hipStream_t)stream>>>(s);
    return tri_check_launch("tri_copy_segments");
}

// nn.GRU bias gradients from tri_gru_bwd's per-chunk sums dbias [nchunk][2 dir][4 (dr, dz, dn_input, dn_hidden)][128]:
// db_ih[dir] = (dr, dz, dn_input), db_hh[dir] = (dr, dz, dn_hidden)  ([384] each; fixed summation order over the chunks)
__global__ void gru_bias_grads_kernel(const float* __restrict__ dbias, int nchunk, float* __restrict__ db_ih_f, float* __restrict__ db_hh_f,
                                      float* __restrict__ db_ih_r, float* __restrict__ db_hh_r) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;        // [2][4][128]
    if (e >= 1024) return;
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += dbias[(size_t)c * 1024 + e];
    const int d = e >> 9, gate = (e >> 7) & 3, u = e & 127;
    float* ih = d ? db_ih_r : db_ih_f;
    float* hh = d ? db_hh_r : db_hh_f;
    if (gate < 2) { ih[gate * 128 + u] = s; hh[gate * 128 + u] = s; }
    else if (gate == 2) ih[256 + u] = s;
    else hh[256 + u] = s;
}
extern "C" int tri_gru_bias_grads(const float* dbias, int nchunk, float* db_ih_f, float* db_hh_f, float* db_ih_r, float* db_hh_r, void* stream) {
    gru_bias_grads_kernel<<<4, 256, 0, (hipStream_t)stream>>>(dbias, nchunk, db_ih_f, db_hh_f, db_ih_r, db_hh_r);
    return tri_check_launch("tri_gru_bias_grads");
}

// fp32 <-> activation-storage casts at the boundary between the fp32 heads and a 16-bit tower
template <typename T>
__global__ void cast_from_f32_kernel(const float4* __restrict__ src, T* __restrict__ dst, long n4, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = src[i];
        Act<T>::st4(dst + i * 4, make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale));
    }
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ src, float4* __restrict__ dst, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) dst[i] = Act<T>::ld4(src + i * 4);
}
extern "C" int tri_cast_from_f32(const float* src, void* dst, long n, float scale, int act_fmt, void* stream) {
    if (n % 4) { tri_set_error("tri_cast_from_f32: n must be a multiple of 4"); return TRI_ERR_ARG; }
    TRI_ACT_DISPATCH(act_fmt, cast_from_f32_kernel<T><<<ew_grid(n / 4), 256, 0, (hipStream_t)stream>>>((const float4*)src, (T*)dst, n / 4, scale));
    return tri_check_launch("tri_cast_from_f32");
}
extern "C" int tri_cast_to_f32(const void* src, float* dst, long n, int act_fmt, void* stream) {
    if (n % 4) { tri_set_error("tri_cast_to_f32: n must be a multiple of 4"); return TRI_ERR_ARG; }
    TRI_ACT_DISPATCH(act_fmt, cast_to_f32_kernel<T><<<ew_grid(n / 4), 256, 0, (hipStream_t)stream>>>((const T*)src, (float4*)dst, n / 4));
    return tri_check_launch("tri_cast_to_f32");
}

// ------------------------------------------------------------------------------------------------ fused Adam
// torch.optim.Adam(lr, betas, eps, weight_decay) single-tensor update, L2-in-gradient (config.yaml:50-53,
// tricolo_net.py:43-44).  `step` lives on the device (incremented by tri_adam_tick) so the launch is graph-replayable.
// `step` is a device int[4]: [0] optimizer steps APPLIED (the t of the bias corrections), [1] gradient elements skipped by the per-element
// guard below, [2] the attempt number (applied + skipped steps + 1) of the last step whose gradient tri_adam_guard* found non-finite,
// [3] steps skipped WHOLE for that reason.  Per-step guard (VERDICT r3 item 9 / ADVICE r3): one inf / NaN in an f16 activation gradient
// poisons the BatchNorm-backward sums and through them every gradient further down that tower, so skipping the bad ELEMENTS trains
// the healthy towers on while one tower stands still.  tri_adam_guard* scans the step's gradient once (before the tick); a bad
// gradient makes tri_adam_tick count a skipped step instead of an applied one and the update kernels leave p, m and v untouched -
// what torch.cuda.amp.GradScaler does with an overflowed step.  All device-side: the decision replays inside a captured HIP graph.
__global__ void adam_tick_kernel(int* step) {
    const int attempt = step[0] + step[3] + 1;
    if (step[2] == attempt) step[3] += 1;                          // the guard flagged this attempt: nothing is applied
    else step[0] += 1;
}
extern "C" int tri_adam_tick(int* step, void* stream) {
    adam_tick_kernel<<<1, 1, 0, (hipStream_t)stream>>>(step);
    return tri_check_launch("tri_adam_tick");
}
__device__ __forceinline__ bool adam_step_skipped(const int* step) { return step[2] == step[0] + step[3]; }   // (after the tick)
__device__ __forceinline__ unsigned nonfinite_bits(const float4& v) {
    const unsigned e = 0x7f800000u;
    return (unsigned)((__builtin_bit_cast(unsigned, v.x) & e) == e) | (unsigned)((__builtin_bit_cast(unsigned, v.y) & e) == e) |
           (unsigned)((__builtin_bit_cast(unsigned, v.z) & e) == e) | (unsigned)((__builtin_bit_cast(unsigned, v.w) & e) == e);
}
__global__ __launch_bounds__(256) void adam_guard_kernel(const float* __restrict__ g, long n, int* __restrict__ step) {
    // scalar head up to the first 16-byte boundary, float4 body, scalar tail
    long head = (long)(((16 - ((uintptr_t)g & 15)) & 15) >> 2);
    if (head > n) head = n;
    const long n4 = (n - head) >> 2, tail0 = head + n4 * 4;
    const float4* body = (const float4*)(g + head);
    unsigned bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) bad |= nonfinite_bits(body[i]);
    if (blockIdx.x == 0) {
        if ((long)threadIdx.x < head) bad |= (unsigned)!__builtin_isfinite(g[threadIdx.x]);
        if ((long)threadIdx.x < n - tail0) bad |= (unsigned)!__builtin_isfinite(g[tail0 + threadIdx.x]);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMax(step + 2, step[0] + step[3] + 1);
}
extern "C" int tri_adam_guard(const float* g, long n, int* step, void* stream) {
    if ((uintptr_t)g & 3) { tri_set_error("tri_adam_guard: gradient not 4-byte aligned"); return TRI_ERR_ARG; }
    adam_guard_kernel<<<ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(g, n, step);
    return tri_check_launch("tri_adam_guard");
}
// bias corrections as torch computes them (Python doubles): 1 - beta^t in double, then step_size / bias_correction2_sqrt
struct AdamCoef { float step_size, inv_sqrt_bc2; };
__device__ __forceinline__ AdamCoef adam_coef(int t, float lr, const float* lr_dev, float b1, float b2) {
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    const double l = lr_dev ? (double)*lr_dev : (double)lr;
    AdamCoef c;
    c.step_size = (float)(l / bc1);
    c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    return c;
}
// Overflow guard of the 16-bit modes: activation gradients are stored in f16 under a static 2^12 scale; should one
// overflow, the inf / NaN reaches a parameter gradient.  An element whose gradient is not finite is NOT applied - its parameter and
// both moments keep their values, so a non-finite value can never enter the fp32 master weights or the Adam state, also inside a
// replayed HIP graph - and counted in step[1] (FusedAdam.nonfinite_skipped() reads it).  This per-ELEMENT guard is the fallback of
// callers that do not run tri_adam_guard* first; with the guard pass a bad gradient skips the whole step (see adam_tick_kernel).
__device__ __forceinline__ void adam_note_bad(const int* step, int bad) {
    if (bad) atomicAdd(const_cast<int*>(step) + 1, bad);
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                            const int* __restrict__ step, float lr, const float* __restrict__ lr_dev, float b1, float b2, float eps,
                            float wd, float gscale) {
    if (adam_step_skipped(step)) return;
    const AdamCoef co = adam_coef(*step, lr, lr_dev, b1, b2);
    const float step_size = co.step_size, inv_sqrt_bc2 = co.inv_sqrt_bc2;
    int bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float pi = p[i];
        float gi = g[i] * gscale + wd * pi;
        if (!__builtin_isfinite(gi)) { ++bad; continue; }          // overflow guard: p, m, v of that element stay (see adam_note_bad)
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
    adam_note_bad(step, bad);
}
// Same update, gradients read in place from the per-parameter tensors autograd produced (segment table: device pointer and
// flat start offset of every parameter, ascending) - no concatenation pass over the 74 MB of gradients first.  Four elements
// per thread; segment sizes / starts are multiples of 4 and the gradient tensors 16-byte aligned (checked by the caller).
#define ADAM_MAX_SEG 1024
__global__ __launch_bounds__(256) void adam_seg_kernel(float* __restrict__ p, const float* const* __restrict__ gptr,
                                                       const long* __restrict__ gstart, int nseg, float* __restrict__ m,
                                                       float* __restrict__ v, long n4, const int* __restrict__ step, float lr,
                                                       const float* __restrict__ lr_dev, float b1, float b2, float eps, float wd,
                                                       float gscale) {
    __shared__ long sstart[ADAM_MAX_SEG];
    __shared__ const float* sptr[ADAM_MAX_SEG];
    if (adam_step_skipped(step)) return;
    for (int i = threadIdx.x; i < nseg; i += 256) { sstart[i] = gstart[i]; sptr[i] = gptr[i]; }
    __syncthreads();
    const AdamCoef co = adam_coef(*step, lr, lr_dev, b1, b2);
    const float step_size = co.step_size, inv_sqrt_bc2 = co.inv_sqrt_bc2;
    int bad = 0;
    for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += (long)gridDim.x * blockDim.x) {
        const long i = i4 * 4;
        int lo = 0, hi = nseg - 1;                                  // last segment whose start is <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sstart[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const float* gs = sptr[lo];
        if (!gs) continue;                                          // no gradient: torch.optim.Adam leaves p, m, v untouched
        const float4 gv = *(const float4*)(gs + (i - sstart[lo]));
        float4 pv = *(const float4*)(p + i), mv = *(const float4*)(m + i), vv = *(const float4*)(v + i);
#define ADAM1(P, G, M, V)                                   \
    {                                                       \
        const float gi = G * gscale + wd * P;               \
        if (__builtin_isfinite(gi)) {                       \
            M = b1 * M + (1.f - b1) * gi;                   \
            V = b2 * V + (1.f - b2) * gi * gi;              \
            P = P - step_size * M / (sqrtf(V) * inv_sqrt_bc2 + eps); \
        } else ++bad;                                       \
    }
        ADAM1(pv.x, gv.x, mv.x, vv.x) ADAM1(pv.y, gv.y, mv.y, vv.y) ADAM1(pv.z, gv.z, mv.z, vv.z) ADAM1(pv.w, gv.w, mv.w, vv.w)
#undef ADAM1
        *(float4*)(p + i) = pv; *(float4*)(m + i) = mv; *(float4*)(v + i) = vv;
    }
    adam_note_bad(step, bad);
}
__global__ __launch_bounds__(256) void adam_guard_seg_kernel(const float* const* __restrict__ gptr, const long* __restrict__ gstart, int nseg,
                                                             long n4, int* __restrict__ step) {
    __shared__ long sstart[ADAM_MAX_SEG];
    __shared__ const float* sptr[ADAM_MAX_SEG];
    for (int i = threadIdx.x; i < nseg; i += 256) { sstart[i] = gstart[i]; sptr[i] = gptr[i]; }
    __syncthreads();
    unsigned bad = 0;
    for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += (long)gridDim.x * blockDim.x) {
        const long i = i4 * 4;
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (sstart[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const float* gs = sptr[lo];
        if (gs) bad |= nonfinite_bits(*(const float4*)(gs + (i - sstart[lo])));
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMax(step + 2, step[0] + step[3] + 1);
}
extern "C" int tri_adam_guard_segments(const void* grad_ptrs, const long* grad_starts, int nseg, long n, int* step, void* stream) {
    if (nseg < 1 || nseg > ADAM_MAX_SEG || n % 4) { tri_set_error("tri_adam_guard_segments: 1..1024 segments, n % 4 == 0"); return TRI_ERR_ARG; }
    adam_guard_seg_kernel<<<ew_grid(n / 4), 256, 0, (hipStream_t)stream>>>((const float* const*)grad_ptrs, grad_starts, nseg, n / 4, step);
    return tri_check_launch("tri_adam_guard_segments");
}
extern "C" int tri_adam_step_segments(float* p, const void* grad_ptrs, const long* grad_starts, int nseg, float* m, float* v, long n,
                                      const int* step, float lr, const float* lr_dev, float b1, float b2, float eps, float wd, float gscale,
                                      void* stream) {
    if (nseg < 1 || nseg > ADAM_MAX_SEG || n % 4) { tri_set_error("tri_adam_step_segments: 1..1024 segments, n % 4 == 0"); return TRI_ERR_ARG; }
    adam_seg_kernel<<<ew_grid(n / 4), 256, 0, (hipStream_t)stream>>>(p, (const float* const*)grad_ptrs, grad_starts, nseg, m, v, n / 4, step, lr,
                                                                      lr_dev, b1, b2, eps, wd, gscale);
    return tri_check_launch("tri_adam_step_segments");
}

extern "C" int tri_adam_step(float* p, const float* g, float* m, float* v, long n, const int* step, float lr, const float* lr_dev,
                             float b1, float b2, float eps, float wd, float gscale, void* stream) {
    adam_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, step, lr, lr_dev, b1, b2, eps, wd, gscale);
    return tri_check_launch("tri_adam_step");
}
