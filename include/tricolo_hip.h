/* tricolo_hip.h - C ABI of libtricolo_hip.so: the MI355X (gfx950) kernels under the TriCoLo training step.
 *
 * The reference (3dlg-hcvc/tricolo) is pure Python and has no FFI; its hot path bottoms out in third-party native
 * code (spconv, cuDNN via torchvision / torch.nn, cuBLAS).  This library is what sits under the reference's
 * encoder / loss modules instead (last row).  Each entry point names the reference call site
 * whose native work it replaces (paths relative to /root/reference).
 *
 * Conventions: plain pointers + sizes, all pointers are DEVICE pointers unless noted; `stream` is a hipStream_t
 * passed as void*; every call is asynchronous on that stream, allocates nothing, keeps no global state and returns
 * 0 on success, a hipError_t (>0) or a negative TRI_ERR_* code otherwise (text via tri_last_error()).  Activations
 * are channels-last [B, D, H, W, C] (2D tensors use D = 1), C a multiple of 4, stored as `act_fmt` says:
 * TRI_FMT_F32 (0: bf16x3 mode), TRI_FMT_BF16 (1: bf16 mode) or TRI_FMT_F16 (2: f16 mode; the 16-bit formats move half
 * the bytes of every pass and ARE the MFMA operand type of the conv kernels); accumulation / statistics are fp32 always.
 */
#ifndef TRICOLO_HIP_H
#define TRICOLO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRI_FMT_F32 0
#define TRI_FMT_BF16 1
#define TRI_FMT_F16 2

int tri_version(void);
const char* tri_last_error(void);

/* Geometry of one convolution / dense layer.  Linear(in,out) is KD=KH=KW=1 on a 1x1x1 grid with B = rows. */
typedef struct TriConvDesc {
    int B, ID, IH, IW, Cin;   /* input grid, stored input channels (multiple of 4; 3-channel inputs are stored as 4) */
    int OD, OH, OW, Cout;     /* output grid, output channels (multiple of 32) */
    int KD, KH, KW, stride, pad_d, pad_h, pad_w;
} TriConvDesc;

/* ---- weight packing -----------------------------------------------------------------------------------------
 * fp32 parameter in the reference's own layout (addressed by element strides) -> MFMA operand rows
 * dst[row][tap * inner_pad + i], bf16 (fmt TRI_FMT_BF16 or 0) or f16 (fmt TRI_FMT_F16), zero padded to
 * tri_conv_kpad(ntaps, inner_pad).  w_lo != NULL (bf16 only) additionally receives the residual x - bf16(x)
 * (3-product split mode, fp32-grade accuracy).
 *   forward operand : rows = Cout, inner = Cin     dgrad operand : rows = Cin, inner = Cout (swap the strides)
 * Layouts: spconv SubMConv3d weight [Cout,kd,kh,kw,Cin] (sparse_cnn.py:12-32), torchvision conv [Cout,Cin,kh,kw]
 * (mv_cnn.py:44), nn.Linear [out,in].
 * frag != 0 (rows % 16 == 0): the same elements in MFMA-FRAGMENT-MAJOR order for the kernels that load their weight fragments straight
 * into registers (tri_conv_kernel_family == 13 / 14 / 15): the [16 rows x 32 k] block (row tile rt, k-step ks) is 1 KiB at
 * ((rt * kpad / 32) + ks) * 512 elements, element (r, k) of it at ((k % 32) / 8 * 16 + r % 16) * 8 + k % 8 - lane (r, k / 8) of a
 * 16x16x32 A fragment reads its 8 elements as ONE contiguous 16-byte piece and a wave reads 1 KiB contiguously. */
int tri_conv_kpad(int ntaps, int cin_stored);
/* one packing job; tri_weight_prep_multi runs an array of them (DEVICE memory) in a single launch - one per tower and step */
typedef struct TriPrepDesc {
    const float* w;
    void* hi;
    void* lo;            /* NULL in the single-operand modes (bf16, f16) */
    long s_row, s_tap, s_inner;
    int rows, ntaps, inner, inner_pad, kpad, fmt;
    int frag;            /* 1: fragment-major order (see above) */
} TriPrepDesc;
int tri_weight_prep_multi(const TriPrepDesc* descs_dev, int n, void* stream);
int tri_weight_prep(const float* w, long s_row, long s_tap, long s_inner, int rows, int ntaps, int inner, int inner_pad,
                    void* w_hi, void* w_lo, int fmt, int frag, void* stream);

/* ---- token embedding (bigru.py:10,15) ------------------------------------------------------------------------------
 * fwd: emb[L,B,D] = W[tokens[B,L]] (already in the GRU's time-major order); bwd: dense dW[V,D], occurrences summed in
 * ascending (l, b) order (deterministic), row padding_idx zero (nn.Embedding(padding_idx=0) semantics). */
int tri_embedding_fwd(const int* tokens, const float* weight, int B, int L, int D, float* out, void* stream);
int tri_embedding_bwd(const int* tokens, const float* dout, int B, int L, int V, int D, int padding_idx, float* dweight, void* stream);

/* ---- text -> shape retrieval ------------------------------------------------------------------------
 * Device-side replacement of eval_retrieval.py:70-82,184-187: float64 similarities text[Nq,D] . shape[Ns,D]^T, the k best
 * shapes per query (descending, equal similarities -> higher index first, i.e. numpy's ascending argsort flipped) and the
 * 0-based rank of the query's own shape label[q] in the full ordering (first_hit, may be NULL together with label). */
int tri_retrieval_topk(const float* text, const float* shape, const int* label, int Nq, int Ns, int D, int k, int* topk_idx /* [Nq,k] */,
                       double* topk_sim /* [Nq,k] */, int* first_hit /* [Nq] */, void* stream);

/* ---- dense layers with <= 64 rows (MLP heads at the per-GPU batch) ---------------------------------------------
 * nn.Linear forward / backward of sparse_cnn.py:39-44, mv_cnn.py:21-26, bigru.py:12, clip_text.py:9-14 in three launches,
 * reading the fp32 parameter directly (no packed copy).  x [M,K], w [N,K] (torch layout), y / dout [M,N]; act 0 none /
 * 1 relu / 2 tanh; the backward entry points take the forward OUTPUT y and apply act' themselves.  split3 = bf16x3. */
int tri_linear_small_supported(int M, int K, int N);               /* M <= 64, K % 128 == 0, N % 128 == 0 */
int tri_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act, int split3, void* stream);
int tri_linear_small_dgrad(const float* dout, const float* y, const float* w, float* dx, int M, int K, int N, int act, int split3,
                           void* stream);
int tri_linear_small_wgrad(const float* x, const float* dout, const float* y, float* dw, float* db /* may be NULL */, int M, int K, int N,
                           int act, int split3, void* stream);
/* tri_linear_small_wgrad + tri_linear_small_dgrad of one layer in a single launch (same results) */
int tri_linear_small_bwd(const float* x, const float* dout, const float* y, const float* w, float* dx, float* dw, float* db, int M, int K,
                         int N, int act, int split3, void* stream);

/* ---- implicit-GEMM convolution on MFMA ------------------------------------------------------------------------
 * tri_conv_fwd replaces spconv.SubMConv3d (sparse_cnn.py:12,17,22,27,32; row_mask = active-site mask gives the
 * submanifold rule), every torchvision conv2d of net_1 (mv_cnn.py:29) and nn.Linear (sparse_cnn.py:39-44,
 * mv_cnn.py:21-26, bigru.py:12, clip_text.py:9-14; bias + act 0 none / 1 relu / 2 tanh fused).
 * stats (optional) receives [tri_conv_num_mtiles][2][Cout] per-tile column sums / sums of squares for BatchNorm.
 * tri_conv_dgrad / tri_conv_wgrad replace the autograd backward of the same call sites; `d` is always the FORWARD
 * descriptor.  wgrad writes dw through element strides, i.e. directly in the reference's parameter layout. */
/* split3: 0 = bf16 operands / fp32 activations, 1 = bf16x3 (hi + lo operands), 2 = 16-bit operands and activation storage (bf16 or f16) */
int tri_conv_num_mtiles(const TriConvDesc* d, int split3);
/* the same for a call that passes a compact row list (row_list = 1: row_pos + row_count): such launches are planned for the voxel
 * grids' occupancy (<= 25 %), so a small level may run split-K over its list where the dense call would not */
int tri_conv_num_records(const TriConvDesc* d, int split3, int row_list);
/* kernel family tri_conv_fwd (transposed = 0) / tri_conv_dgrad (transposed = 1) dispatches for this layer and mode:
 * 0 conv_igemm_kernel (register-staged im2col), 2 conv_dma_kernel (LDS-DMA staging); bits 8.. hold the output-channel
 * tile width; bit 16 set = the dense call of the layer runs split-K in this mode; 4 conv_stem_kernel, 3 / 5 the halo kernels,
 * 6 = the brick kernel of conv_vox.hip (voxel level 0; takes the site mask as row_mask, refuses a row list), 9 / 10 = the
 * register-stationary filter-bank kernels of conv_c64.hip (9 conv_c64_kernel: 64 -> 64 channels 3x3 / 1, both directions; 10 conv_s2d_kernel:
 * data gradient of the 64 -> 128 channel 3x3 / 2 layer), 12 conv_pw_kernel (1x1 / 2 shortcuts; 7 and 11 were A/B partners dropped in round 6),
 * 13 conv_voxg_kernel (conv_voxg.hip: SubMConv3d on 2^3 / 4^3 / 8^3 grids, both directions; takes the site mask as row_mask, refuses a
 * row list; bits 8..15 = output channels per workgroup, bits 24.. = samples per unit = per BatchNorm record), 14 conv_voxb_kernel (voxel
 * level 1 on 16^3 / 32^3 grids, forward only; site mask as row_mask), 15 conv_s2g_kernel (conv_s2g.hip: forward of the 3x3 / 2 layers with >= 128 input
 * channels - layer3 / layer4's opening convolutions; refuses row mask / list, bias, activation, accumulate; one BatchNorm record per unit of images).
 * Families 13 / 14 / 15 read FRAGMENT-MAJOR packed operands.  For profilers. */
int tri_conv_kernel_family(const TriConvDesc* d, int transposed, int split3);
/* same for tri_conv_wgrad: 0 conv_wgrad_kernel, 2 conv_wgrad_dma_kernel (taken when act_fmt != 0 and the layer qualifies) */
int tri_conv_wgrad_kernel_family(const TriConvDesc* d, int act_fmt);
/* split-K scratch a small-M layer needs (0 = none): pass at least this many bytes to tri_conv_fwd / tri_conv_dgrad */
size_t tri_conv_workspace(const TriConvDesc* d, int transposed);
int tri_conv_fwd(const TriConvDesc* d, const void* in, const void* w_hi, const void* w_lo, void* out, const uint8_t* row_mask,
                 const float* bias, int act, int accumulate, float* stats, int act_fmt, void* workspace, size_t workspace_bytes,
                 const int* row_pos /* optional compact row list, see tri_mask_compact */, const int* row_count /* DEVICE int */,
                 void* stream);
int tri_conv_dgrad(const TriConvDesc* d, const void* dout, const void* wt_hi, const void* wt_lo, void* din, const uint8_t* row_mask,
                   int accumulate, int act_fmt, void* workspace, size_t workspace_bytes,
                   const int* row_pos /* optional [B*ID*IH*IW]: a permutation of the input positions, the order in which the
                                         kernel's row tiles visit them.  For stride 2 pass the positions sorted by the parity of
                                         (coordinate + pad) per axis: a tile then only runs the 1-4 taps of 9 its rows can use
                                         (the other products are structurally zero).  Purely an execution-order hint. */,
                   const int* row_count /* optional DEVICE int: row_pos is a COMPACT list, only rows [0, *row_count) are computed
                                           (submanifold layers: the active sites, tri_mask_compact); other rows of din are NOT written */,
                   void* stream);
/* Data gradient that also takes the BatchNorm-backward sums of the BatchNorm whose OUTPUT gradient it produces: din is the gradient
 * w.r.t. relu(bn(y)) (relu_scale / relu_shift: the mask y * scale + shift > 0 is recomputed), w.r.t. relu(bn(y) + shortcut) whose
 * saved output is relu_out (mask relu_out > 0), or w.r.t. bn(y) itself (neither).  partial[records][2][C] receives sum g', sum g' * y
 * over the values AS STORED in din (g' = masked), the layout tri_bn_bwd_reduce writes and tri_bn_bwd_finalize sums: the separate
 * reduce pass over din / y (module/img_encoder/mv_cnn.py:44 BasicBlock backward: one per BatchNorm2d) is not launched at all.
 * tri_conv_dgrad_bn_records: number of records, or 0 when this layer's data-gradient kernel has no fused form (then call
 * tri_conv_dgrad and tri_bn_bwd_reduce). */
typedef struct TriConvBnSums {
    const void* y;            /* [B, ID, IH, IW, Cin] the BatchNorm's input, storage format of din */
    const float* relu_scale;  /* optional [Cin] */
    const float* relu_shift;
    const void* relu_out;     /* optional, shape of y */
    float* partial;           /* out */
} TriConvBnSums;
int tri_conv_dgrad_bn_records(const TriConvDesc* d, int accumulate, int act_fmt);
int tri_conv_dgrad_bn(const TriConvDesc* d, const void* dout, const void* wt_hi, const void* wt_lo, void* din, int accumulate,
                      int act_fmt, void* workspace, size_t workspace_bytes, const int* row_pos, const TriConvBnSums* sums, void* stream);
size_t tri_conv_wgrad_workspace(const TriConvDesc* d);
/* gather plan of a layer geometry (origin offset + tap validity bits per output position): build once, reuse every step */
size_t tri_conv_plan_bytes(const TriConvDesc* d);
int tri_conv_plan_build(const TriConvDesc* d, void* plan, void* stream);
int tri_conv_wgrad(const TriConvDesc* d, const void* in, const void* dout, const uint8_t* row_mask, const void* plan /* required */,
                   void* workspace, size_t workspace_bytes, float* dw, long s_co, long s_tap, long s_ci, int cin_real, int split3,
                   int act_fmt, float out_scale /* dw = out_scale * sum: undoes the f16 mode's gradient scaling */,
                   const int* row_pos, const int* row_count /* optional compact list of the output positions to contract over
                       (tri_mask_compact: the active sites of a submanifold layer) - executed work = active rows; row_mask unused */,
                   void* stream);
/* The same in two halves, so that a tower's backward pays ONE reduce launch instead of one per layer: tri_conv_wgrad_partial runs
 * the position-split partial kernel into `workspace` (which must stay untouched until the reduce) and fills *pending;
 * tri_wgrad_reduce_grouped sums the slabs of n pending layers into their dw (any n; TRI_WGRAD_GROUP_MAX layers per launch).
 * Same arithmetic and summation order as tri_conv_wgrad: results are bitwise identical. */
#ifndef TRI_WGRAD_GROUP_MAX
#define TRI_WGRAD_GROUP_MAX 24
#endif
typedef struct TriWgradReduce {
    const float* slab;
    float* dw;
    long s_co, s_tap, s_ci;
    int splits, Cout, Kpad, ntaps, cin_stored, cin_real, zlanes, nblocks;   /* zlanes 0: row form (one output channel per block, the
                                                                               parameter row written as one contiguous run) */
    float out_scale;
    int kw_real;                       /* 0, or (slabs whose kernel rows are padded to 2^kw_shift taps: stem 8, voxel level 0 4) the real kernel width */
    int kw_shift;
} TriWgradReduce;
int tri_conv_wgrad_partial(const TriConvDesc* d, const void* in, const void* dout, const uint8_t* row_mask, const void* plan /* required */,
                           void* workspace, size_t workspace_bytes, float* dw, long s_co, long s_tap, long s_ci, int cin_real, int split3,
                           int act_fmt, float out_scale, const int* row_pos, const int* row_count, TriWgradReduce* pending /* HOST, out */,
                           void* stream);
int tri_wgrad_reduce_grouped(const TriWgradReduce* pending /* HOST array */, int n, void* stream);
/* Same, and every workgroup that STORED an inf / NaN gradient element flags the optimizer's current attempt in `note` (the int32[4]
 * device record of tri_adam_guard / tri_adam_tick: note[2] = max(note[2], note[0] + note[3] + 1)) - the overflow scan of these tensors
 * (torch.cuda.amp.GradScaler's found_inf; tricolo_amd/optim.py) then needs no pass of its own.  note may be NULL. */
int tri_wgrad_reduce_grouped_noted(const TriWgradReduce* pending /* HOST array */, int n, int* note, void* stream);
/* Several layers' partial kernels in ONE launch (16-bit activation storage).  The launch's resident
 * workgroups are shared by the jobs, so each layer is cut into fewer, longer splits than alone: the fp32 slab traffic of the step
 * (splits x Cout x K per layer, written here and re-read by the reduce) shrinks by about the number of jobs.
 * tri_conv_wgrad_group_info: family 0 = the layer is not groupable (use tri_conv_wgrad_partial), else jobs of equal family may share
 * a launch; tiles = workgroups per split, steps = 64-position steps (the caller's budget: ~448 workgroups per launch).
 * tri_conv_wgrad_partial_group: n <= TRI_WGRAD_JOBS_MAX jobs of one family (dense position ranges or compact row lists, no row mask);
 * workspace sized by tri_conv_wgrad_workspace as before;
 * pending[i] (HOST, out) is job i's reduce descriptor for tri_wgrad_reduce_grouped. */
#define TRI_WGRAD_JOBS_MAX 12
typedef struct TriWgradJob {
    const TriConvDesc* d;
    const void* in;
    const void* dout;
    const void* plan;
    void* workspace;
    size_t workspace_bytes;
    float* dw;
    long s_co, s_tap, s_ci;
    int cin_real;
    float out_scale;
    const int* row_pos;                /* optional compact list of the output positions to contract over (tri_mask_compact) ... */
    const int* row_count;              /* ... and its device-side length, as tri_conv_wgrad_partial takes them */
} TriWgradJob;
/* Stem weight gradient fused with tri_maxpool_bn_bwd_apply (conv 7x7/2 -> BN -> ReLU -> MaxPool2d(3,2,1), mv_cnn.py:44): dW from the
 * conv output y, the pool's winning-tap map / pooled gradient and the BatchNorm-backward coefficients; the gradient w.r.t. the conv
 * output is never stored.  TRI_ERR_UNSUPPORTED when the layer does not run conv_stem_wgrad_kernel (then: apply + tri_conv_wgrad_partial). */
int tri_conv_stem_wgrad_bn(const TriConvDesc* d, const void* in, const void* y, const uint8_t* arg, const void* dpool, const float* c1,
                           const float* c2, const float* c3, const float* relu_scale, const float* relu_shift, void* workspace,
                           size_t workspace_bytes, float* dw, long s_co, long s_tap, long s_ci, int cin_real, int act_fmt, float out_scale,
                           TriWgradReduce* pending /* HOST, out */, void* stream);
int tri_conv_wgrad_group_info(const TriConvDesc* d, int act_fmt, int* family, int* tiles, int* steps);
int tri_conv_wgrad_partial_group(const TriWgradJob* jobs /* HOST array */, int n, int act_fmt, TriWgradReduce* pending /* HOST array, out */,
                                 void* stream);

/* ---- BatchNorm (train-mode statistics, eps / momentum as torch.nn.BatchNorm1d/2d) ----------------------------------
 * Replaces nn.BatchNorm1d over active voxels (sparse_cnn.py:13,18,23,28,33; count from a device counter) and the 20
 * BatchNorm2d of ResNet-18 (mv_cnn.py:29; count_host = N*H*W). */
int tri_bn_finalize(const float* partial, int ntiles, int C, const int* count_dev, int count_host, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                    float eps, float* mean, float* invstd, float* scale, float* shift, void* stream);
int tri_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* mean,
                       float* invstd, float* scale, float* shift, void* stream);
int tri_bn_act(const void* y, const float* scale, const float* shift, const void* res, const float* rscale, const float* rshift,
               void* out, long M, int C, int relu, int act_fmt, void* stream);
int tri_relu_bwd(const void* dout, const void* out, void* g, long n, int act_fmt, void* stream);
int tri_bn_bwd_num_blocks(long M);
/* relu_scale / relu_shift (optional): g is the gradient w.r.t. relu(bn(y)); the mask y*scale+shift > 0 is recomputed */
/* relu_out (optional): g is the gradient w.r.t. relu(bn(y) + residual) and relu_out that ReLU's saved output (mask out > 0) */
int tri_bn_bwd_reduce(const void* y, const void* g, long M, int C, float* partial, const float* relu_scale, const float* relu_shift,
                      const void* relu_out, const uint8_t* row_mask /* optional: rows with 0 are skipped (never read) */, int act_fmt,
                      void* stream);
/* BatchNorm backward of a SMALL tensor (accepted up to M = 16,384 rows; the Python gate ops.bn_bwd uses it for M <= 512 only - from
 * ~1 k rows on the three-pass form is faster: this kernel's loads are 16 bytes per cache line) in ONE launch: a workgroup owns 4 / 8 channels for all positions - sums,
 * coefficients (double) and the apply pass without records or a finalize launch.  Arguments as tri_bn_bwd_reduce / _finalize / _apply
 * (dy may alias g); TRI_ERR_UNSUPPORTED for other shapes. */
int tri_bn_bwd_small(const void* y, const void* g, long M, int C, const int* count_dev, int count_host, const float* gamma,
                     const float* mean, const float* invstd, const float* relu_scale, const float* relu_shift, const void* relu_out,
                     void* g_masked, const uint8_t* row_mask, int keep_inactive, void* dy, float* dgamma, float* dbeta, float out_scale,
                     int act_fmt, void* stream);
int tri_bn_bwd_finalize(const float* partial, int nblk, int C, const int* count_dev, int count_host, const float* gamma,
                        const float* mean, const float* invstd, float* dgamma, float* dbeta, float* c1, float* c2, float* c3,
                        float out_scale /* dgamma, dbeta *= out_scale (g carries the f16 mode's gradient scale; dy keeps it) */,
                        void* stream);
int tri_bn_bwd_apply(const void* y, const void* g, const float* c1, const float* c2, const float* c3, const uint8_t* row_mask,
                     void* dy, long M, int C, const float* relu_scale, const float* relu_shift, const void* relu_out,
                     void* g_masked /* optional: receives g * (relu_out > 0), may alias g */,
                     int keep_inactive /* 0: rows with row_mask == 0 come out as zeros (a data gradient gathers them); 1: they are left
                                          untouched - for layers whose dy only feeds a weight gradient over the same mask (voxel level 0) */,
                     int act_fmt, void* stream);
/* BatchNorm backward of TWO tensors of one shape that share their upstream gradient - bn2 and the shortcut's BatchNorm of a down-sampling
 * BasicBlock (mv_cnn.py:20; out = relu(bn_a(ya) + bn_b(yb)), g = gradient of out, relu_out = out): the three passes serve both, g and
 * relu_out are read once.  partial_* [tri_bn_bwd_num_blocks(M)][2][C]; buf_* [5][C] = dgamma, dbeta, c1, c2, c3; g_masked (may alias g)
 * receives g * (out > 0). */
int tri_bn_bwd_pair_reduce(const void* ya, const void* yb, const void* g, const void* relu_out, long M, int C, float* partial_a,
                           float* partial_b, int act_fmt, void* stream);
int tri_bn_bwd_pair_finalize(const float* partial_a, const float* partial_b, int nblk, int C, int count_host, const float* gamma_a,
                             const float* mean_a, const float* invstd_a, float* buf_a, const float* gamma_b, const float* mean_b,
                             const float* invstd_b, float* buf_b, float out_scale, void* stream);
int tri_bn_bwd_pair_apply(const void* ya, const void* yb, const void* g, const void* relu_out, const float* buf_a, const float* buf_b,
                          void* dya, void* dyb, void* g_masked, long M, int C, int act_fmt, void* stream);

/* ---- pooling -------------------------------------------------------------------------------------------------
 * BN + ReLU + mask + spconv.SparseMaxPool3d(2,2) fused (sparse_cnn.py:13-15 ...), its backward routing;
 * nn.MaxPool2d(3,2,1) of the ResNet stem; AdaptiveAvgPool2d + torch.max over views (mv_cnn.py:29-31). */
int tri_bn_relu_pool3d_fwd(const void* y, const float* scale, const float* shift, const uint8_t* mask, int B, int D, int C,
                           void* pooled, uint8_t* mask_out, int act_fmt, void* stream);
/* Row-list forms of the voxel tower's backward passes (round 4): the routing pass over the ACTIVE pooled sites (out_pos / out_count = the
 * next level's tri_mask_compact list) and the whole BatchNorm backward (reduce + finalize + apply, three launches) over the level's own
 * list, rows outside the list neither read nor written (tri_bn_bwd_apply's keep_inactive contract).  Same values as the dense forms. */
int tri_pool3d_bwd_route_rows(const void* y, const float* scale, const float* shift, const uint8_t* mask, const void* pooled,
                              const void* dpooled, int B, int D, int C, void* g, const int* out_pos, const int* out_count, int act_fmt,
                              void* stream);
/* (round 5) the routing walk with the level's BatchNorm-backward sums folded in, as tri_pool3d_bwd_route_reduce does for the dense walk:
 * partial [tri_pool3d_bwd_route_rows_num_blocks][2][C] feeds tri_bn_bwd_finalize; the apply pass is then tri_bn_bwd_apply (site mask) or
 * tri_bn_bwd_apply_rows (dy = c1 * g + c2 + c3 * y on the rows of the level's own list only; dy may alias g).  C / 4 must divide 256.
 * All 2x2x2-window kernels need fewer than 2^31 sites and a 2-byte aligned mask (TRI_ERR_ARG otherwise). */
int tri_pool3d_bwd_route_rows_num_blocks(int B, int D, int C);
int tri_pool3d_bwd_route_rows_reduce(const void* y, const float* scale, const float* shift, const uint8_t* mask, const void* pooled,
                                     const void* dpooled, int B, int D, int C, void* g, const int* out_pos, const int* out_count,
                                     float* partial, int act_fmt, void* stream);
int tri_bn_bwd_apply_rows(const void* y, const void* g, const float* c1, const float* c2, const float* c3, void* dy, int C,
                          const int* row_pos, const int* row_count, long max_rows, int act_fmt, void* stream);
size_t tri_bn_bwd_rows_scratch(int C);
int tri_bn_bwd_rows(const void* y, const void* g, int C, const int* row_pos, const int* row_count, long max_rows, const float* gamma,
                    const float* mean, const float* invstd, void* dy, float* dgamma, float* dbeta, float out_scale, void* scratch,
                    int act_fmt, void* stream);
int tri_pool3d_bwd_route(const void* y, const float* scale, const float* shift, const uint8_t* mask, const void* pooled,
                         const void* dpooled, int B, int D, int C, void* g, int act_fmt, void* stream);
/* the same with the level's BatchNorm-backward sums folded in (sparse_cnn.py:12-35 backward: SparseMaxPool3d -> ReLU -> BatchNorm1d):
 * partial [tri_pool3d_bwd_route_reduce_num_blocks][2][C] = per-workgroup sums of g and g * y over the active sites, what
 * tri_bn_bwd_reduce would produce from a second pass; continue with tri_bn_bwd_finalize / tri_bn_bwd_apply.  C / 4 must divide 256. */
int tri_pool3d_bwd_route_reduce_num_blocks(int B, int D, int C);
int tri_pool3d_bwd_route_reduce(const void* y, const float* scale, const float* shift, const uint8_t* mask, const void* pooled,
                                const void* dpooled, int B, int D, int C, void* g, float* partial, int act_fmt, void* stream);
/* bn_scale / bn_shift (optional, [C]): pools relu(x * scale + shift), i.e. BatchNorm + ReLU + MaxPool2d of the ResNet stem in one pass */
int tri_maxpool2d_fwd(const void* x, int N, int H, int W, int C, void* out, uint8_t* arg /* [N,Ho,Wo,C] winning tap, may be NULL */,
                      const float* bn_scale, const float* bn_shift, int act_fmt, void* stream);
int tri_maxpool2d_bwd(const uint8_t* arg, const void* dout, int N, int H, int W, int C, void* dx, int act_fmt, void* stream);
/* stem backward without materialising the max-pool gradient (conv -> BN -> ReLU -> MaxPool2d(3,2,1), mv_cnn.py:44): the BatchNorm
 * backward passes read (y [N,H,W,C], arg [N,H/2,W/2,C], dpool [N,H/2,W/2,C]) and route the pooled gradient to the winning taps
 * on the fly.  reduce -> tri_bn_bwd_finalize(partial, tri_maxpool_bn_bwd_num_blocks(...), C, count = N*H*W) -> apply.  Same
 * values as tri_maxpool2d_bwd followed by tri_bn_bwd_reduce / _apply with relu_scale / relu_shift.  H, W even. */
int tri_maxpool_bn_bwd_num_blocks(int N, int H, int W);
int tri_maxpool_bn_bwd_reduce(const void* y, const uint8_t* arg, const void* dpool, int N, int H, int W, int C, float* partial,
                              const float* relu_scale, const float* relu_shift, int act_fmt, void* stream);
/* round 4: the same [blocks][2][C] sums from the POOLED tensors (pooled = the forward's max-pool output, which the next layer keeps):
 * sum(g) and sum(g y) are sums over windows, and an active window's winning activation is (pooled - shift) / scale; 50 MB instead of
 * 137 MB read at the bench shape.  Windows whose recovered activation would be ill-conditioned (|pooled| > 64 |gamma|) read the
 * stored y through the tap map.  16-bit storage, C % 8 == 0 and C / 8 a divisor of 256; else TRI_ERR_UNSUPPORTED. */
int tri_maxpool_bn_bwd_pooled_num_blocks(int N, int H, int W);
int tri_maxpool_bn_bwd_reduce_pooled(const void* pooled, const void* dpool, const void* y, const uint8_t* arg, int N, int H, int W, int C,
                                     float* partial, const float* relu_scale, const float* relu_shift, const float* gamma, int act_fmt,
                                     void* stream);
int tri_maxpool_bn_bwd_apply(const void* y, const uint8_t* arg, const void* dpool, int N, int H, int W, int C, const float* c1,
                             const float* c2, const float* c3, const float* relu_scale, const float* relu_shift, void* dy, int act_fmt,
                             void* stream);
int tri_avgpool_viewmax_fwd(const void* x, int B, int V, int HW, int C, float* out, int* arg, int act_fmt, void* stream);
int tri_avgpool_viewmax_bwd(const float* dout, const int* arg, int B, int V, int HW, int C, void* dx, int act_fmt, float scale, void* stream);

/* ---- layout converters (batch layout of tricolo/data/data_module.py:40-65) ---------------------------------------- */
/* dense [B,V,V,V,4] and mask [B*V^3 padded to 32 bytes] are zero-filled by the call (one fill when mask directly follows dense) */
int tri_voxel_scatter(const int* locs, const float* feats, int n, int B, int V, void* dense, uint8_t* mask, int act_fmt, void* stream);
/* the dataset's dense RGBA u8 grids [B,4,V,V,V] straight to the tower input (active <=> alpha != 0, feats =
 * RGB / 255; general_dataset.py:47-51,92-93) - no CPU COO build, no scatter.  mask must hold B*V^3 bytes (padded to 32). */
int tri_voxel_from_rgba_u8(const uint8_t* rgba, int B, int V, void* dense, uint8_t* mask, int act_fmt, void* stream);
int tri_mask_count(const uint8_t* mask, long n, int* count, void* stream);
/* instrumentation: *slot = wall_clock64() (100 MHz) at this point of the stream; tools/step_timeline.py */
int tri_debug_stamp(unsigned long long* slot, void* stream);
/* toolchain regression probe: copies n16 16-byte elements src -> dst through the raw-buffer builtins (form 0 load with a per-lane offset -
 * what the conv kernels use -, 1 load with an SGPR scalar offset, 2 / 3 the same for stores; csrc/misc.hip buffer_b128_probe_kernel) */
int tri_debug_buffer_b128_probe(const void* src, void* dst, long n16, int form, void* stream);
/* active-site list of a submanifold level: row_pos[0 .. *count) = positions with mask != 0, ascending; *count = how many.
 * Hand row_pos + count to tri_conv_fwd / tri_conv_dgrad: they then compute (and write) ONLY those rows - executed work =
 * active work (spconv's rulebook idea on a dense index space).  Split-K layers take the list too (slab row = list row).  scratch: tri_mask_compact_scratch(n) bytes. */
size_t tri_mask_compact_scratch(long n);
int tri_mask_compact(const uint8_t* mask, long n, int* row_pos, int* count, void* scratch, void* stream);
/* (round 5) the voxel tower's five site masks and five row lists up front: tri_mask_pyramid builds the masks of levels 1-4 (2x2x2 OR-pool,
 * what tri_bn_relu_pool3d_fwd writes as mask_out level by level - that argument may now be NULL) from the level-0 mask of [B, V, V, V]
 * grids in one launch (V % 16 == 0; masks[l - 1] holds B * (V >> l)^3 bytes rounded up to 32, padding zeroed); tri_mask_compact_multi is
 * tri_mask_compact of up to 8 masks in two launches (each list <= 2,097,152 sites).  Same masks, same lists, same counts. */
int tri_mask_pyramid(const uint8_t* mask0, int B, int V, uint8_t* const* masks, void* stream);
size_t tri_mask_compact_multi_scratch(const long* n, int nlist);
int tri_mask_compact_multi(const uint8_t* const* masks, const long* n, int nlist, int* const* rows, int* const* counts, void* scratch,
                           void* stream);
int tri_nchw3_to_nhwc4(const float* x, int N, int H, int W, void* out, int act_fmt, void* stream);
/* u8 images [N,3,H,W] -> channels-last [N,H,W,4], (u8/255 - mean[c]) / std[c] as ToTensor + Normalize of
 * general_dataset.py:87-89; mean3 / std3 are HOST pointers to three floats. */
int tri_nchw3_u8_to_nhwc4(const uint8_t* x, int N, int H, int W, const float* mean3, const float* std3, void* out, int act_fmt, void* stream);

/* ---- row ops ------------------------------------------------------------------------------------------------------
 * F.normalize(dim=1) (sparse_cnn.py:51, mv_cnn.py:33, bigru.py:18), bias gradients, activation backward. */
int tri_l2norm_fwd(const float* x, int rows, int D, float eps, float* z, float* norm, void* stream);
int tri_l2norm_bwd(const float* z, const float* norm, const float* dz, int rows, int D, float eps, float* dx, void* stream);
int tri_colsum(const float* g, long M, int C, float* out, void* stream);
int tri_axpy(const float* x, float a, float* y, long n, void* stream);
int tri_act_bwd(const float* dout, const float* out, float* g, long n, int act, void* stream);
/* dst (act_fmt storage) = scale * src (fp32), and back: the fp32 heads <-> 16-bit tower boundary of the voxel tower */
int tri_cast_from_f32(const float* src, void* dst, long n, float scale, int act_fmt, void* stream);
int tri_cast_to_f32(const void* src, float* dst, long n, int act_fmt, void* stream);

/* ---- persistent bidirectional GRU recurrence (nn.GRU(256,128,bidirectional) of text_encoder/bigru.py:11,17) -----------
 * xproj [L][B][768] = x_t W_ih^T + b_ih for both directions (768 = dir*384 + gate*128 + unit, gate order r,z,n);
 * w_hh [2][384][128], b_hh [2][384].  Outputs: hs [2][L][B][128], gates [2][L][B][128][4] (r,z,n,hn per unit) saved for backward,
 * hfinal [B][256] = cat(forward final state, reverse final state) (bigru.py:18).  tri_gru_bwd returns the gate
 * pre-activation gradients dgi [L][B][768], dgh [2][L][B][384] and h_{t-1} rows hprev [2][L][B][128]; the weight
 * gradients are then plain GEMMs (tri_conv_wgrad with 1x1 geometry).
 * split3: MFMA operand mode of the recurrence - 0 single bf16 products, 1 the 3-product bf16 hi/lo split (fp32-grade), 2 single f16
 * products (the f16 mode; the backward carries its gate gradients through LDS times 2^12). */
int tri_gru_fwd(const float* xproj, const float* w_hh, const float* b_hh, int B, int L, float* hs, float* gates, float* hfinal,
                int split3, void* stream);
int tri_gru_bwd(const float* dhfinal, const float* w_hh, const float* hs, const float* gates, int B, int L, float* dgi, float* dgh,
                float* hprev, float* dbias /* [ceil(B/16)][2][4][128] per-chunk sums of (dr, dz, dn_input, dn_hidden) */,
                int split3, void* stream);

/* up to 8 contiguous fp32 segments copied in one launch; srcs / dsts / n are HOST arrays of device pointers / element counts
 * (the per-step [w_ih_f; w_ih_r] ... concatenations of the nn.GRU parameters, bigru.py:11) */
int tri_copy_segments(const void* const* srcs, void* const* dsts, const long* n, int count, void* stream);
/* nn.GRU bias gradients from tri_gru_bwd's dbias: db_ih[dir] = (dr, dz, dn_input), db_hh[dir] = (dr, dz, dn_hidden), [384] each */
int tri_gru_bias_grads(const float* dbias, int nchunk, float* db_ih_f, float* db_hh_f, float* db_ih_r, float* db_hh_r, void* stream);

/* ---- NT-Xent loss, forward + backward fused (tricolo/loss/nt_xent.py:24-74) ------------------------------------------ */
size_t tri_ntxent_workspace(int B, int D);
int tri_ntxent_fwd_bwd(const float* za, const float* zb, int B, int D, float temperature, float alpha, int norm, float* loss,
                       float* dza, float* dzb, void* workspace, size_t workspace_bytes, void* stream);
/* gradient half alone, from the workspace a tri_ntxent_fwd_bwd(dza = dzb = NULL) call filled; dloss (optional DEVICE scalar)
 * = upstream d(total)/d(loss), folded in */
int tri_ntxent_bwd(const float* za, const float* zb, int B, int D, float temperature, float alpha, int norm, const float* dloss,
                   float* dza, float* dzb, const void* workspace, size_t workspace_bytes, void* stream);
/* All pairs of a step at once (tricolo_net.py:56-63: the loss of every pair of modalities, summed): M = 2 or 3 embeddings
 * z[m] [B, D] (HOST array of device pointers, in the reference's modality order - the earlier one of a pair is the alpha
 * side), 4 launches forward, 1 backward.  losses [P + 1] (device) = the pair losses in itertools.combinations order, then
 * their sum ((l0 + l1) + l2 in fp32, as Python's sum()).  The backward takes the upstream gradients of the pair losses
 * (HOST array of device scalars, NULL entries / NULL array = 0) and of the total (device scalar or NULL) and writes
 * dz[m] = sum over m's pairs.  B <= 512, D % 4 == 0, D <= 2048 (TRI_ERR_UNSUPPORTED otherwise: use the per-pair calls). */
size_t tri_ntxent_multi_workspace(int M, int B, int D);
int tri_ntxent_multi_fwd(const float* const* z, int M, int B, int D, float temperature, float alpha, int norm, float* losses,
                         void* workspace, size_t workspace_bytes, void* stream);
int tri_ntxent_multi_bwd(const float* const* z, int M, int B, int D, float temperature, float alpha, int norm,
                         const float* const* dpair, const float* dtotal, float* const* dz, const void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---- Adam (torch.optim.Adam as instantiated by config/config.yaml:50-53, tricolo_net.py:43-44) ------------------------ */
/* `step` is a DEVICE int[4]: [0] optimizer steps applied (the t of the bias corrections; tri_adam_tick adds one), [1] the running count
 * of gradient elements the update kernels skipped because they were inf / NaN (per-element fallback: such an element leaves p, m and v
 * untouched), [2] attempt number of the last step whose gradient tri_adam_guard* found non-finite, [3] steps skipped whole.
 * Per-step overflow guard of the 16-bit modes (torch.cuda.amp.GradScaler's rule, decided on the device so that it replays inside a HIP
 * graph): call tri_adam_guard / tri_adam_guard_segments over the step's gradient BEFORE tri_adam_tick; when it finds an inf / NaN the
 * tick counts a skipped step instead of an applied one and the update kernels return without touching p, m, v. */
int tri_adam_guard(const float* g, long n, int* step, void* stream);
int tri_adam_guard_segments(const void* grad_ptrs, const long* grad_starts, int nseg, long n, int* step, void* stream);
int tri_adam_tick(int* step, void* stream);
/* lr_dev (optional, DEVICE float): when not NULL the learning rate is read from it at run time, so a captured HIP graph
 * follows a schedule (LrDecayCallback of train.py) without re-capture; `lr` is used otherwise.  Bias corrections in double. */
int tri_adam_step(float* p, const float* g, float* m, float* v, long n, const int* step, float lr, const float* lr_dev, float b1, float b2,
                  float eps, float wd, float gscale, void* stream);
/* the same update with the gradients read in place: grad_ptrs = DEVICE array of nseg device pointers (NULL = no gradient: the
 * segment is SKIPPED like torch.optim.Adam skips parameters whose .grad is None), grad_starts = DEVICE array of the flat start offset of each segment (ascending, first 0); segment sizes and n are
 * multiples of 4, gradient tensors 16-byte aligned.  Saves the flat-gradient concatenation pass of the single-GPU step. */
int tri_adam_step_segments(float* p, const void* grad_ptrs, const long* grad_starts, int nseg, float* m, float* v, long n,
                           const int* step, float lr, const float* lr_dev, float b1, float b2, float eps, float wd, float gscale,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif
