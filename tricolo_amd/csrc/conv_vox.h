// Internal interface between conv_igemm.hip (plans + dispatch of tri_conv_fwd / tri_conv_dgrad) and conv_vox.hip (brick kernels of the
// voxel tower).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct TriVox0Geom { int V, TY, grid; };
// true when the layer is level 0 of the voxel tower in a 16-bit storage mode (4 stored input channels, 32 outputs, 3x3x3 / 1 / pad 1 on a
// 32^3 / 64^3 / 128^3 grid); g->grid = workgroups = BatchNorm records of the launch
bool tri_internal_vox0_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                                int pd, int ph, int pw, TriVox0Geom* g);
int tri_internal_vox0_launch(const TriVox0Geom& g, int B, const void* in, const void* w, int kpad, void* out, const uint8_t* mask, float* stats,
                             int act_fmt, hipStream_t stream);

// level-0 weight gradient (conv_vox0_wgrad_kernel): persistent workgroups of the launch (0: switched off), each writes one fp32 slab [32][144]
int tri_internal_vox0_wgrad_grid(const TriVox0Geom& g);
int tri_internal_vox0_wgrad_launch(const TriVox0Geom& g, int grid, int B, const void* in, const void* dout, const uint8_t* mask, float* slab,
                                   int act_fmt, hipStream_t stream);

// the SubMConv3d layers on the coarse grids (2^3 / 4^3 / 8^3 sites per sample, 64 | cin, 16 | cout, 16-bit storage): conv_voxg_kernel
// (conv_voxg.hip), forward and data gradient over the SITE MASK; spu samples per unit, nunits units = BatchNorm records, ct output
// channels per workgroup
struct TriVoxgGeom { int D, spu, nunits, ct, grid, smem; };
bool tri_internal_voxg_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                                int pd, int ph, int pw, TriVoxgGeom* g);
int tri_internal_voxg_launch(const TriVoxgGeom& g, int B, int cin, int cout, int kpad, const void* in, const void* w, void* out, const uint8_t* mask,
                             float* stats, int transposed, int act_fmt, hipStream_t stream);

// level 1 of the voxel tower (32 -> 64 channels on 16^3 / 32^3 grids, 16-bit storage): conv_voxb_kernel (conv_voxg.hip) - bricks of 256
// sites, ranked active rows, filter bank stationary; forward only, over the SITE MASK; g->grid = persistent workgroups = BatchNorm records
struct TriVoxbGeom { int D, nbricks, grid; };
bool tri_internal_voxb_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                                int pd, int ph, int pw, TriVoxbGeom* g);
int tri_internal_voxb_launch(const TriVoxbGeom& g, int B, const void* in, const void* w, void* out, const uint8_t* mask, float* stats, int act_fmt,
                             hipStream_t stream);

// the 3x3 / stride 2 / pad 1 layers that open layer3 / layer4 of the ResNet trunk (Cin >= 128, Cin % 64 == 0, Cout % 64 == 0, <= 192 output
// positions per image, 16-bit storage): conv_s2g_kernel (conv_s2g.hip), forward only; units of whole images, g->nunits BatchNorm records
struct TriS2gGeom { int ipu, nunits, nrt, PW, PS, NSP, HB, grid, smem; };
bool tri_internal_s2g_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                               int pd, int ph, int pw, TriS2gGeom* g);
int tri_internal_s2g_launch(const TriS2gGeom& g, int B, int IH, int IW, int cin, int cout, int kpad, const void* in, const void* w, void* out,
                            float* stats, int act_fmt, hipStream_t stream);

struct TriC64Geom { int W, TY, nbricks, grid; };
// a 64 -> 64 channel 3x3 / 1 / pad 1 2D layer on 16- / 32- / 64-pixel-wide images in a 16-bit storage mode (layer1 of the ResNet trunk):
// conv_c64_kernel (conv_c64.hip), forward and data gradient; g->grid = persistent workgroups = BatchNorm records of the launch
bool tri_internal_c64_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                               int pd, int ph, int pw, TriC64Geom* g);
// data gradient of a 64 -> 128 channel 3x3 / 2 / pad 1 2D layer (GEMM view: 128 -> 64 channels onto a grid twice as large), dOut rows of
// 16 / 32 pixels, 16-bit storage: conv_s2d_kernel (conv_c64.hip)
bool tri_internal_s2d_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                               int pd, int ph, int pw, TriC64Geom* g);
int tri_internal_s2d_launch(const TriC64Geom& g, int B, int H, const void* in, const void* w, void* out, int accumulate, int act_fmt,
                            hipStream_t stream);
struct TriConvBnSums;                                                         // include/tricolo_hip.h
int tri_internal_c64_launch(const TriC64Geom& g, int B, int H, const void* in, const void* w, void* out, float* stats, int transposed,
                            int accumulate, int act_fmt, const TriConvBnSums* bs, hipStream_t stream);

// the 1x1 / stride-2 / pad-0 shortcut convolutions (64 | cin <= 512, 64 | cout, 16-bit storage): conv_pw_kernel (conv_pw.hip), forward
// (BatchNorm records: one per 128-row tile) and data gradient (writes the dense dIn tensor, zeros included; no accumulate form)
struct TriPwGeom { int bn, mtiles, GH, GW, transposed; };
bool tri_internal_pw_geometry(int B, int ID, int IH, int IW, int cin, int OD, int OH, int OW, int cout, int KD, int KH, int KW, int stride,
                              int pd, int ph, int pw, TriPwGeom* g);
int tri_internal_pw_launch(const TriPwGeom& g, int B, const void* in, const void* w, int K, int N, int Kpad, void* out, float* stats, int transposed,
                           int act_fmt, hipStream_t stream);
