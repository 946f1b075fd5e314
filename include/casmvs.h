/*
 * casmvs.h - C ABI of libcasmvs_hip.so: the MI355X (gfx950) cascade-MVS depth engine.
 *
 * This is the drop-in boundary for the hot path of kwea123/CasMVSNet_pl
 * (`models/mvsnet.py::CascadeMVSNet.forward` and the `models/modules.py` functions it calls).
 * The reference has no FFI layer of its own (it is 100 % Python on torch ops), so every entry
 * point below names the reference function (file:line under /root/reference) it replaces; the
 * Python host side (`casmvsnet_pl_amd/`) binds them with ctypes and mirrors the reference's
 * Python API on top.
 *
 * Conventions (all entry points):
 *   - return 0 (CASMVS_OK) or a negative CASMVS_ERR_* code; never throw across the ABI.
 *     `casmvs_last_error()` returns a thread-local human readable message for the last failure.
 *   - every tensor is fp32, contiguous, row-major in the reference's own layout (NCHW / NCDHW);
 *     the *_nhwc entry points (pixel-major feature maps) say so explicitly.
 *   - the CALLER owns every buffer (inputs, outputs, workspace, packed weights).  The library
 *     allocates no device memory and keeps no pointer after return.
 *   - device pointers are raw `hipDeviceptr`-style `float*` on the calling thread's current
 *     device; `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream).
 *   - fully asynchronous on `stream`: no internal synchronisation, re-entrant; the only global state
 *     is a mutex-protected cache of per-(device, kernel) launch constants (occupancy, LDS opt-in); the
 *     library never reads the environment.  Inputs
 *     are `const`; outputs must not alias inputs.
 *   - gfx950 only.  No CPU fallback exists: on a machine without a gfx950 device the launch entry
 *     points fail with CASMVS_ERR_HIP.
 *
 * BUILD.  The SHIPPED library is NOT a plain `hipcc -c` of csrc/: casmvsnet_pl_amd/build.py compiles every source to device assembly
 * (`hipcc --offload-arch=gfx950 -O3 -ffp-contract=off --cuda-device-only -S`), exchanges the two commuting sources of every packed-float32
 * instruction of the form the MI355X gets wrong beside f16 / bf16 matrix instructions (see casmvs_packed_opsel_safe below), assembles, links and
 * bundles the code object back into the host object, and defines CASMVS_PACKED_OPSEL_SAFE=1; the linked .so is then disassembled and linted.
 * A plain `hipcc -shared` build of the same sources is a working library with the same results ON ONE STREAM, but casmvs_packed_opsel_safe()
 * returns 0 for it, and a DIRECT C-ABI caller (one that does not go through casmvsnet_pl_amd/streams.py, which applies the rule for Python callers)
 * must then serialise by itself: no kernel of this library that issues f16 / bf16 matrix instructions (the *_splitf16_* / *_splitbf16_* / zmarch /
 * zfused entry points and the whole-net calls that select them) may overlap, on another stream of the same GPU, any float32 kernel of this library -
 * order them with events, or run one stream.  Nothing in the library detects a violation; the wrong values are silent.
 */
#ifndef CASMVS_H
#define CASMVS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CASMVS_ABI_VERSION 6

#define CASMVS_OK 0
#define CASMVS_ERR_INVALID_ARG (-1) /* null pointer / non-positive size / unsupported combination */
#define CASMVS_ERR_UNSUPPORTED (-2) /* shape outside what the kernels were built for             */
#define CASMVS_ERR_HIP (-3)         /* HIP runtime / launch error (message has hipGetErrorString) */

/* ---- housekeeping ------------------------------------------------------------------------ */

/* ABI version of the loaded library (== CASMVS_ABI_VERSION of the header it was built with). */
int casmvs_abi_version(void);

/* Thread-local message describing the last non-zero return on this thread ("" if none). */
const char *casmvs_last_error(void);

/* 1 when the library's device code was assembled with the packed-float32 operand exchange of casmvsnet_pl_amd/build.py (rewrite_unsafe_packed), 0 for
 * a plain `hipcc -c` build.  gfx950: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel:[0,1,..] (low result from src0's low and a vector src1's
 * high half) read that half as zero in lanes 48-63 while another wave of the SIMD issues f16 / bf16 matrix instructions - another kernel on another
 * stream, or other waves of the same kernel (tools/probes/pk_fma_opsel_repro.hip).  A library that returns 1 contains no such
 * instruction (tests/test_device_code_lints.py disassembles the shipped .so): its kernels may run beside each other on any number of streams.  With
 * 0 the caller keeps kernels with f16 matrix instructions from overlapping float32 kernels of other streams (casmvsnet_pl_amd/streams.py does). */
int casmvs_packed_opsel_safe(void);

/* ---- (a3)/(a4) depth hypotheses ------------------------------------------------------------
 * Replaces: models/mvsnet.py:213-229 (coarsest level: d_k = init_depth_min + k*interval) and
 *           models/mvsnet.py:231-235 + models/modules.py:34-49 (finer levels: bilinear x2
 *           upsample of the previous depth with align_corners=True, then
 *           d_min = max(u - half_range, 1e-7), d_k = d_min + k*interval).
 *
 * prev_depth  : device (B, hp, wp) previous-level depth, or NULL for the coarsest level.
 * depth_min_b : device (B) per-sample init_depth_min (used only when prev_depth == NULL).
 * interval_b  : device (B) per-sample depth interval of THIS level (depth_interval*ratio[l]).
 * half_range_b: device (B) per-sample (n_depths/2)*interval (used only when prev_depth != NULL).
 * out         : device (B, D, h, w).  When prev_depth != NULL, (h, w) is the x2 grid of
 *               (hp, wp) in the reference; any h >= 2, w >= 2 works (ATen align_corners rule).
 */
int casmvs_depth_hypotheses_f32(const float *prev_depth, const float *depth_min_b,
                                const float *interval_b, const float *half_range_b, float *out,
                                int B, int D, int h, int w, int hp, int wp, void *stream);

/* ---- (a5) homo_warp ------------------------------------------------------------------------
 * Replaces: models/modules.py:52-92 `homo_warp(src_feat, proj_mat, depth_values)`.
 * src   : device (B, C, H, W)     source-view feature map
 * proj  : device (B, 3, 4)        (P_src @ inv(P_ref))[:3]
 * depth : device (B, D, H, W)     per-pixel depth hypotheses of the reference view
 * out   : device (B, C, D, H, W)  warped source volume (bilinear, zeros padding, align_corners)
 * Coordinates follow the reference's fp32 operation order; its seven divisions per tap set are
 * evaluated to <= 1 ulp (reciprocal + one Newton step), everything else is separately rounded.
 */
int casmvs_homo_warp_f32(const float *src, const float *proj, const float *depth, float *out,
                         int B, int C, int H, int W, int D, void *stream);

/* ---- (a5)+(a6) fused plane sweep + variance cost volume ------------------------------------
 * Replaces: models/mvsnet.py:134-168 for G == 1 (ref broadcast, the V-1 homo_warp calls, the
 *           sum / sum-of-squares accumulation and `var = sq/V - (sum/V)^2`), without ever
 *           materialising a warped volume.
 * feats : device (B, V, C, h, w)   view 0 is the reference view
 * proj  : device (B, V-1, 3, 4)
 * depth : device (B, D, h, w)
 * out   : device (B, C, D, h, w)
 */
int casmvs_costvol_var_f32(const float *feats, const float *proj, const float *depth, float *out,
                           int B, int V, int C, int h, int w, int D, void *stream);

/* ---- (a5)+(a7) fused plane sweep + group-wise correlation cost volume ----------------------
 * Replaces: models/mvsnet.py:142-144,157-162,169-172 for G > 1.
 * out   : device (B, G, D, h, w); channels are split contiguously C -> (G, C/G).
 * Supported: C in {8, 16, 32}, G divides C.
 */
int casmvs_costvol_gwc_f32(const float *feats, const float *proj, const float *depth, float *out,
                           int B, int V, int C, int G, int h, int w, int D, void *stream);

/* Channel-last variants of the two cost-volume builders: feats is (B, V, h, w, C) - every pixel's C
 * features contiguous, 16-byte aligned - so that a bilinear tap of 4 channels is one 16-byte load
 * (half the gather instructions of the NCHW kernels, whole cache lines per gather; bit-identical
 * results).  C in {8, 16, 32}.  casmvs_nchw_to_nhwc_f32 converts a (N, C, h, w) map, C in {8,16,32}. */
int casmvs_nchw_to_nhwc_f32(const float *in, float *out, int N, int C, int h, int w, void *stream);
int casmvs_costvol_var_nhwc_f32(const float *feats, const float *proj, const float *depth, float *out,
                                int B, int V, int C, int h, int w, int D, void *stream);
int casmvs_costvol_gwc_nhwc_f32(const float *feats, const float *proj, const float *depth, float *out,
                                int B, int V, int C, int G, int h, int w, int D, void *stream);

/* ---- LDS-staged plane sweep (the fast path) ---------------------------------------------------
 * Same operations and results (bit-identical) as the *_nhwc gather kernels above, but the box of source pixels
 * a tile of 256 reference pixels reaches over a chunk of 8 depth planes is staged once in LDS and every
 * bilinear tap is an LDS read (the gather kernels are bound by the CU's texture path, not by HBM).
 * feats (B, V, h, w, C) pixel-major, C in {8, 16, 32}; D % 8 == 0; at most 8 source views per call.
 * casmvs_costvol_lds_supported: 1 when a shape has an LDS plan (otherwise use the *_nhwc kernels);
 * casmvs_costvol_lds_preferred: 1 when that plan is also the faster kernel on the MI355X (measured per channel count).
 * casmvs_homo_warp_nhwc_f32 replaces models/modules.py:52-92 for a pixel-major source map (B, H, W, C):
 *   out (B, C, D, H, W) exactly as casmvs_homo_warp_f32. */
int casmvs_costvol_lds_supported(int C, int w, int D, int n_src_views, int G);
int casmvs_costvol_lds_preferred(int C, int w, int D, int n_src_views, int G);
int casmvs_costvol_var_lds_f32(const float *feats, const float *proj, const float *depth, float *out,
                               int B, int V, int C, int h, int w, int D, void *stream);
int casmvs_costvol_gwc_lds_f32(const float *feats, const float *proj, const float *depth, float *out,
                               int B, int V, int C, int G, int h, int w, int D, void *stream);
int casmvs_homo_warp_nhwc_f32(const float *src, const float *proj, const float *depth, float *out,
                              int B, int C, int H, int W, int D, void *stream);
/* The same LDS-staged sweep on the reference's own layout, src (B, C, H, W) (modules.py:52-92 takes exactly this): the source box is staged from the
 * channel planes (whole quads of x, transposed in registers) - no pixel-major copy of the map exists.  Bit-identical to the two entries above.
 * C in {8, 16, 32}, W % 4 == 0, 16-byte aligned tensors; casmvs_homo_warp_lds_supported says whether the shape has an LDS plan. */
int casmvs_homo_warp_lds_supported(int C, int W, int D);
int casmvs_homo_warp_lds_f32(const float *src, const float *proj, const float *depth, float *out, int B,
                             int C, int H, int W, int D, void *stream);

/* ---- view-sharded build (BASELINE configs 4/5) --------------------------------------
 * The sums of models/mvsnet.py:147-167 are linear in the source views: a rank warps the source views
 * [view_begin, view_end) (1-based indices into feats' view axis, view 0 = reference) and produces
 *   variance:    sum = [ref] + SUM_v warped_v,  sq = [ref^2] + SUM_v warped_v^2     (B, C, D, h, w) each;
 *                include_ref != 0 on exactly one rank adds the bracketed terms;
 *   correlation: out = mean_{c in group}( (SUM_v warped_v) * ref )                   (B, G, D, h, w).
 * After an all-reduce(SUM) of those buffers every rank finalises:
 *   var = sq / V - (sum / V)^2   (casmvs_costvol_var_finalize_f32; may run in place, out == sum or sq)
 *   cor = out / (V - 1)          (casmvs_costvol_gwc_finalize_f32; may run in place).
 * With a single rank (views [1, V), include_ref = 1) partial + finalise equals the fused kernels bit for bit.
 * n = number of floats (a multiple of 4). */
int casmvs_costvol_partial_var_f32(const float *feats, const float *proj, const float *depth, float *sum,
                                   float *sq, int B, int V, int C, int h, int w, int D, int view_begin,
                                   int view_end, int include_ref, void *stream);
int casmvs_costvol_partial_gwc_f32(const float *feats, const float *proj, const float *depth, float *out,
                                   int B, int V, int C, int G, int h, int w, int D, int view_begin,
                                   int view_end, void *stream);
int casmvs_costvol_var_finalize_f32(const float *sum, const float *sq, float *out, size_t n, int V, void *stream);
int casmvs_costvol_gwc_finalize_f32(const float *in, float *out, size_t n, int V, void *stream);

/* ---- (a8) CostRegNet 3D convolutions (fp32 MFMA) -------------------------------------------
 * Replaces: models/mvsnet.py:60-104 `CostRegNet` and models/modules.py:21-31 `ConvBnReLU3D`
 *           (nn.Conv3d / nn.ConvTranspose3d, k=3, pad=1, followed by eval-mode ABN =
 *           BatchNorm(running stats) + leaky_relu, optionally followed by a skip add).
 *
 * Layer kinds.  Every kind computes
 *      y = lrelu_slope( conv(x) * scale[co] + shift[co] ) (+ skip)
 * where (scale, shift) is the folded eval-mode ABN (or scale = 1, shift = bias for `prob`).
 */
#define CASMVS_CONV_S1 0     /* Conv3d k3 s1 p1              : (B,Cin,D,H,W) -> (B,Cout,D,H,W)       */
#define CASMVS_CONV_S2 1     /* Conv3d k3 s2 p1              : (B,Cin,D,H,W) -> (B,Cout,D/2,H/2,W/2) */
#define CASMVS_CONV_T2 2     /* ConvTranspose3d k3 s2 p1 op1 : (B,Cin,D,H,W) -> (B,Cout,2D,2H,2W)    */

/* Number of floats of the packed (device-layout) image of one layer's parameters. */
size_t casmvs_conv3d_packed_floats(int kind, int cin, int cout);

/* HOST-side packing of one layer (pure CPU, no HIP calls): permutes the torch-layout weight into
 * the MFMA A-operand images the kernels stream, and appends scale[cout], shift[cout].
 * weight : host; Conv3d (cout, cin, 3,3,3) for S1/S2, ConvTranspose3d (cin, cout, 3,3,3) for T2
 * scale, shift : host (cout) or NULL (=> 1 and 0)
 * packed : host, casmvs_conv3d_packed_floats(kind, cin, cout) floats
 */
int casmvs_conv3d_pack_f32(int kind, int cin, int cout, const float *weight, const float *scale,
                           const float *shift, float *packed);

/* One layer.  `packed` is the device copy of the image produced by casmvs_conv3d_pack_f32.
 * in   : device (B, cin, D, H, W);  skip : device, shape of out, or NULL;  out : device.
 * D, H, W are the INPUT dims.  For S2 they must be even.
 * slope: leaky-relu negative slope (0.01 for ABN, 1.0 for no activation).
 * Supported cout: S1 {1, 8, 16, 32, 64}; S2 {16, 32, 64}; T2 {8, 16, 32}.  Any cin >= 1.
 */
int casmvs_conv3d_forward_f32(int kind, const float *packed, const float *in, const float *skip,
                              float *out, int B, int cin, int cout, int D, int H, int W,
                              float slope, void *stream);

/* CostRegNet.conv0 (Conv3d cin -> 8, k3 s1 p1, folded ABN, leaky-relu; mvsnet.py:63,91) on the bf16 matrix cores with
 * float32-grade arithmetic: every float32 operand = the exact sum of three bf16 slices, a product = six (terms = 0 / 6) or
 * all nine (terms = 9) exact bf16 x bf16 partial products accumulated in float32 (csrc/conv0_splitbf16.hip).
 * cin in {8, 16, 32}, W % 4 == 0, 16-byte aligned tensors.  `packed`: HOST image from casmvs_conv0_splitbf16_pack
 * (weight (8, cin, 3, 3, 3), scale / shift (8) or NULL), copied to the device by the caller.
 * casmvs_selftest_mfma_bf16: lane-semantics probe of v_mfma_f32_16x16x32_bf16 (dump: NULL or 256 floats). */
size_t casmvs_conv0_splitbf16_packed_bytes(int cin);
int casmvs_conv0_splitbf16_pack(int cin, const float *weight, const float *scale, const float *shift, void *packed);
int casmvs_conv0_splitbf16_supported(int cin, int W);
int casmvs_conv0_splitbf16_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int D, int H, int W,
                                       float slope, int terms, void *stream);
int casmvs_selftest_mfma_bf16(float *dump);

/* The same layer on the f16 matrix cores (csrc/conv0_splitf16.hip): every float32 operand = the sum of two float16 numbers
 * (22 of its 24 significand bits) after an exact power-of-two scaling into float16's range - one per weight tensor (host),
 * one per staged (tile, 8-channel chunk) of the input (device, from the tile's largest magnitude) - three (terms = 0 / 3)
 * or four (terms = 4) exact f16 x f16 partial products accumulated in float32.  Half the matrix instructions and 2/3 of the
 * LDS bytes of the bf16 variant: two workgroups share a CU.  Same argument rules as casmvs_conv0_splitbf16_*; weights must be
 * finite.  casmvs_selftest_mfma_f16: lane-semantics probe of v_mfma_f32_16x16x32_f16 (dump: NULL or 256 floats). */
size_t casmvs_conv0_splitf16_packed_bytes(int cin);
int casmvs_conv0_splitf16_pack(int cin, const float *weight, const float *scale, const float *shift, void *packed);
int casmvs_conv0_splitf16_supported(int cin, int W);
int casmvs_conv0_splitf16_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int D, int H, int W,
                                      float slope, int terms, void *stream);
/* conv0 in the same split-f16 arithmetic, input-stationary along z (csrc/conv0_zmarch.hip): a workgroup owns a 16 x 32 (y, x) patch and
 * marches along z, staging every input plane once per chunk of 8 channels (per-plane power-of-two scaling) and feeding the three
 * output planes it touches - half the staged voxels per output voxel of casmvs_conv0_splitf16_forward_f32, whose L1 -> L2 request
 * stream bounds it.  `packed`: the image of casmvs_conv0_splitf16_pack.  cin = 8, 16 (two workgroups per CU) or 32 (one), W % 4 == 0.  Results agree with the other
 * entry to ~1e-6 of the range (both ~3e-7 from a float64 convolution), not bit for bit.
 * Patches are 8 x 64 for cin = 8 / 16 (fewer partial cache lines per staged row: the kernel is bound by its read traffic), 16 x 32 for cin = 32.
 * Measured on the MI355X (tools/native/conv0_zm_check.cpp, batch 8, dirtied caches, profiles/r04_conv0_zm_wide_ab.txt): 1.5x the tiled kernel at cin = 16
 * (cascade level 1), 1.26x at cin = 8, 0.9-1.04x at cin = 32 - casmvs_costreg_regress_f32 uses it for cin = 8 and 16. */
int casmvs_conv0_zmarch_supported(int cin, int W);
int casmvs_conv0_zmarch_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int D, int H, int W, float slope,
                                    void *stream);
/* CostRegNet.conv11 = ConvTranspose3d(16 -> 8, k3 s2 p1 op1) + ABN + leaky-relu, plus the `conv0 + ...` skip (models/mvsnet.py:84-86, 101), on the
 * f16 matrix cores in the same split arithmetic (csrc/deconv11_splitf16.hip): the x parities of the output are the two halves of the MFMA rows,
 * K = 2 input x positions x 16 input channels, one MFMA set per (kz, ky) pair and 32 output x; output tile 4 x 8 x 32 from a 3 x 5 x 18 input box.
 * in (B, 16, Di, Hi, Wi); skip (B, 8, 2 Di, 2 Hi, 2 Wi) or NULL; out like skip.  weight (16, 8, 3, 3, 3) = the torch ConvTranspose3d layout.
 * casmvs_conv3d_forward_f32(CASMVS_CONV_T2, ...) is the float32-MFMA form of the same layer (0.9 ms of the 8.5 ms step).
 * Measured on the MI355X (tools/native/deconv11_check.cpp): 1.0-1.03x the float32 kernel, equal bits run to run; casmvs_costreg_regress_f32 takes its image as split_layers[5]. */
size_t casmvs_deconv11_splitf16_packed_bytes(void);
int casmvs_deconv11_splitf16_pack(const float *weight, const float *scale, const float *shift, void *packed);
int casmvs_deconv11_splitf16_supported(int Wi);
int casmvs_deconv11_splitf16_forward_f32(const void *packed, const float *in, const float *skip, float *out, int B, int Di, int Hi, int Wi,
                                         float slope, void *stream);

/* CostRegNet.conv9 = ConvTranspose3d(32 -> 16, k3 s2 p1 op1) + ABN + leaky-relu, plus the `conv2 + ...` skip (models/mvsnet.py:80-82, 99) the same
 * way (csrc/deconv9_splitf16.hip): MFMA rows = the 16 output channels, K = the 32 input channels of one input voxel, three sets per (kz, ky) pair
 * (kx = 1 -> even outputs; kx = 2 and kx = 0 from the next input -> odd outputs).  in (B, 32, Di, Hi, Wi); skip / out (B, 16, 2 Di, 2 Hi, 2 Wi).
 * Measured on the MI355X (tools/native/deconv9_check.cpp): 1.2x the float32 kernel at the engine's shapes; casmvs_costreg_regress_f32 takes its image as split_layers[4]. */
size_t casmvs_deconv9_splitf16_packed_bytes(void);
int casmvs_deconv9_splitf16_pack(const float *weight, const float *scale, const float *shift, void *packed);
int casmvs_deconv9_splitf16_supported(int Wi);
int casmvs_deconv9_splitf16_forward_f32(const void *packed, const float *in, const float *skip, float *out, int B, int Di, int Hi, int Wi,
                                        float slope, void *stream);

int casmvs_selftest_mfma_f16(float *dump);

/* CostRegNet's stride-1 layers with equal channel counts (conv2: 16 -> 16, conv4: 32 -> 32, conv6: 64 -> 64; Conv3d k3 s1 p1 + folded ABN +
 * leaky-relu, mvsnet.py:66,69,72) in the arithmetic of casmvs_conv0_splitf16_forward_f32, channel-inner matrix form
 * (csrc/conv_ci_splitf16.hip).  (cin, cout) in {(16, 16), (32, 32), (64, 64)} (64 -> 64: D >= 3), W % 2 == 0, tensors 8-byte aligned.  `packed`: HOST image from casmvs_conv_ci_splitf16_pack
 * (weight (cout, cin, 3, 3, 3) finite, scale / shift (cout) or NULL), copied to the device (16-byte aligned) by the caller. */
size_t casmvs_conv_ci_splitf16_packed_bytes(int cin, int cout);
int casmvs_conv_ci_splitf16_pack(int cin, int cout, const float *weight, const float *scale, const float *shift, void *packed);
int casmvs_conv_ci_splitf16_supported(int cin, int cout, int W);
int casmvs_conv_ci_splitf16_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int cout, int D, int H, int W,
                                        float slope, void *stream);

/* CostRegNet's stride-2 layers conv1 (8 -> 16) and conv3 (16 -> 32) (Conv3d k3 s2 p1 + folded ABN + leaky-relu, mvsnet.py:64-65,67-68) in the same
 * arithmetic, input-stationary along z (csrc/conv_s2_splitf16.hip): an output patch of 6 x 32 marches over the input planes, every plane patch staged
 * once.  in (B, cin, D, H, W) -> out (B, cout, (D - 1) / 2 + 1, (H - 1) / 2 + 1, (W - 1) / 2 + 1); (cin, cout) in {(8, 16), (16, 32)}, W % 4 == 0, `in` and
 * the image 16-byte aligned.  `packed`: HOST image from casmvs_conv_s2_splitf16_pack (weight (cout, cin, 3, 3, 3) finite, scale / shift (cout) or NULL),
 * copied to the device by the caller.  casmvs_costreg_regress_f32 takes the images as split_layers[6] (conv1) and [7] (conv3). */
size_t casmvs_conv_s2_splitf16_packed_bytes(int cin, int cout);
int casmvs_conv_s2_splitf16_pack(int cin, int cout, const float *weight, const float *scale, const float *shift, void *packed);
int casmvs_conv_s2_splitf16_supported(int cin, int cout, int W);
int casmvs_conv_s2_splitf16_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int cout, int D, int H, int W,
                                        float slope, void *stream);

/* Whole CostRegNet (mvsnet.py:91-104).  `packed_layers[11]` are the device images of
 * conv0..conv6, conv7, conv9, conv11, prob (in that order).  `workspace` holds the intermediate
 * activations; its size comes from casmvs_costreg_workspace_bytes.
 * vol : device (B, cin, D, h, w)  ->  cost : device (B, D, h, w)   (the `prob` head, 1 channel)
 * D, h, w must be divisible by 8.  slope: leaky-relu slope of every ABN (0.01 in the reference).
 * layer_events: NULL, or 12 caller-created `hipEvent_t` handles: event i is recorded on `stream`
 *               right before layer i is launched and event 11 after the last layer, so a caller
 *               can time each kernel (hipEventElapsedTime) without any synchronisation in here.
 */
size_t casmvs_costreg_workspace_bytes(int B, int D, int h, int w);

/* HOST-side packing of the WHOLE CostRegNet (models/mvsnet.py:60-89) in one call: the eleven float32 layer images of `packed_layers`
 * (conv0..conv6, conv7, conv9, conv11, prob) back to back in ONE blob, every image 16-byte aligned.  Equal, image by image, to eleven
 * casmvs_conv3d_pack_f32 calls (the per-layer form stays: a caller that re-packs one changed layer uses it).
 * casmvs_costreg_packed_floats(cin, layer_offsets): floats of the blob; layer_offsets (NULL or 11 entries) receives every image's float
 *   offset inside it - after ONE host-to-device copy of the blob, packed_layers[i] = device_blob + layer_offsets[i].
 * weights[11]: host, torch layouts (Conv3d (cout,cin,3,3,3); ConvTranspose3d (cin,cout,3,3,3) for conv7 / conv9 / conv11);
 * scales[11] / shifts[11]: host (cout) per layer, the folded eval-mode ABN (prob: NULL scale, shift = bias); NULL arrays / entries => 1 and 0. */
size_t casmvs_costreg_packed_floats(int cin, size_t *layer_offsets);
int casmvs_costreg_pack_f32(int cin, const float *const *weights, const float *const *scales, const float *const *shifts, float *packed);

int casmvs_costreg_forward_f32(const float *const *packed_layers, const float *vol, float *cost,
                               void *workspace, int B, int cin, int D, int h, int w, float slope,
                               void *const *layer_events, void *stream);

/* ---- (f-1) FeatureNet: 2D convolutions on the same MFMA kernels ------------------------------
 * Replaces: models/mvsnet.py:7-57 (FeatureNet) and models/modules.py:5-18 (ConvBnReLU):
 *   y = lrelu_slope( conv2d(x) * scale[co] + shift[co] ) (+ up), x (N, cin, H, W), "same" padding.
 * The 2D layers run on the 3D kernels with a 1-deep kernel (kz = 1, D = 1).
 */
#define CASMVS_CONV2D_K3 3     /* Conv2d k3 s1 p1 : (N,Cin,H,W) -> (N,Cout,H,W);    cout 8 or % 16 == 0 */
#define CASMVS_CONV2D_K5S2 4   /* Conv2d k5 s2 p2 : (N,Cin,H,W) -> (N,Cout,H/2,W/2); cout % 16 == 0     */
#define CASMVS_CONV2D_K1 5     /* Conv2d k1       : (N,Cin,H,W) -> (N,Cout,H,W);    cout % 16 == 0      */
#define CASMVS_CONV2D_K1_UP 6  /* Conv2d k1 + bilinear x2 (align_corners) upsampling of `up` added:
                                  the FPN top-down step of mvsnet.py:36-38; `up` is (N,Cout,H/2,W/2)  */

/* Packing: as casmvs_conv3d_pack_f32 with weight (cout, cin, k, k).  Host only. */
size_t casmvs_conv2d_packed_floats(int kind, int cin, int cout);
int casmvs_conv2d_pack_f32(int kind, int cin, int cout, const float *weight, const float *scale,
                           const float *shift, float *packed);

/* One 2D layer.  in : device (N, cin, H, W) with H, W the INPUT dims (even for K5S2 and K1_UP);
 * up : device (N, cout, H/2, W/2) for K1_UP, otherwise NULL;  out : device. */
int casmvs_conv2d_forward_f32(int kind, const float *packed, const float *in, const float *up,
                              float *out, int N, int cin, int cout, int H, int W, float slope,
                              void *stream);

/* Whole FeatureNet (mvsnet.py:40-57).  packed_layers[13]: conv0.0, conv0.1, conv1.0, conv1.1, conv1.2,
 * conv2.0, conv2.1, conv2.2 (ABN folded), toplayer, lat1, lat0, smooth1, smooth0 (bias as shift).
 * imgs : device (N, 3, H, W), H % 4 == 0, W % 4 == 0
 * feat0 (N, 8, H, W), feat1 (N, 16, H/2, W/2), feat2 (N, 32, H/4, W/4) : device outputs
 * feat{0,1,2}_nhwc : NULL, or device (N, H, W, 8), (N, H/2, W/2, 16), (N, H/4, W/4, 32), 16-byte aligned:
 *                    the same maps pixel-major, written by the same kernels, for casmvs_costvol_*_nhwc_f32
 * layer_events: NULL or 14 hipEvent_t handles (event i before layer i, event 13 after the last). */
size_t casmvs_featurenet_workspace_bytes(int N, int H, int W);
int casmvs_featurenet_forward_f32(const float *const *packed_layers, const float *imgs, float *feat0,
                                  float *feat1, float *feat2, float *feat0_nhwc, float *feat1_nhwc,
                                  float *feat2_nhwc, void *workspace, int N, int H, int W, float slope,
                                  void *const *layer_events, void *stream);

/* FeatureNet's full-resolution FPN tail as one kernel: feat0 = smooth0(lat0(conv0) + upsample2x(feat1_sum))
 * (mvsnet.py:36-38,50-51,54) without materialising the 32-channel sum.  The three operators are linear:
 *   packed40 : device image of casmvs_conv2d_pack_f32(CASMVS_CONV2D_K3, cin = 40, cout = 8) of the weight
 *              [ (smooth0.weight o lat0.weight) (8, 8, 3, 3) | smooth0.weight (8, 32, 3, 3) ] along the input channels
 *              (scale = 1, shift = 0), 16-byte aligned;
 *   bias9    : device (3, 3, 8): smooth0.bias + the sum over smooth0's taps that lie INSIDE the image of
 *              smooth0.weight[:, :, ky, kx] . lat0.bias, for the row classes (first / inner / last row) x column classes;
 *   conv0 (N, 8, H, W); feat1_sum (N, 32, H/2, W/2) = lat1(conv1) + up(feat2); feat0 (N, 8, H, W); feat0_nhwc NULL or
 *   (N, H, W, 8).  H even, W % 4 == 0, W >= 8 (casmvs_fpn_tail0_supported).
 */
int casmvs_fpn_tail0_supported(int H, int W);
int casmvs_fpn_tail0_f32(const float *packed40, const float *bias9, const float *conv0, const float *feat1_sum,
                         float *feat0, float *feat0_nhwc, int N, int H, int W, void *stream);

/* casmvs_featurenet_forward_f32 with the full-resolution tail (lat0, upsample-add, smooth0) run by casmvs_fpn_tail0_f32
 * (packed_layers[10] / [12] are then unused but must still be non-NULL).  layer_events: the `lat0` interval times the fused
 * kernel, the `smooth0` interval is empty.  fused0_arith: 0 = fused0_packed is the float32 image (casmvs_conv2d_pack_f32 of the
 * composed 40-channel layer, casmvs_fpn_tail0_f32), 1 = the split-f16 image (casmvs_fpn_tail0_splitf16_pack,
 * casmvs_fpn_tail0_splitf16_f32: the same kernel on the f16 matrix cores in the arithmetic of casmvs_conv0_splitf16_forward_f32).
 * casmvs_fpn_tail0_splitf16_pack: HOST, weight40 (8, 40, 3, 3) float32 finite -> casmvs_fpn_tail0_splitf16_packed_bytes() bytes.
 * ci_layers: NULL, or EIGHT pointers (ABI version 6; versions 3-5 read seven, version 2 five) { conv1.1, conv1.2, conv2.1, conv2.2, smooth1: DEVICE copies of
 * casmvs_conv2d_ci_splitf16_pack's images; conv1.0, conv2.0: of casmvs_conv2d_k5s2_splitf16_pack's; conv0 (= conv0.0 + conv0.1): of casmvs_fnet_conv0_mm_pack's }
 * (an entry may be NULL): those layers then run on the f16 matrix cores (casmvs_conv2d_ci_splitf16_forward_f32 / casmvs_conv2d_k5s2_splitf16_forward_f32 /
 * casmvs_fnet_conv0_mm_f32: one launch for the two layers - the `conv0.0` interval of layer_events times it, `conv0.1` is empty).
 * feat0 / feat1 may be NULL when feat0_nhwc / feat1_nhwc are given (ABI version 3): nothing downstream of FeatureNet reads the (N, C, h, w) layout of
 * levels 0 / 1 - the plane sweep gathers pixel-major - and the engine's own call drops those stores.  feat2 feeds lat1: never NULL. */
/* FeatureNet's 3x3 stride-1 layers with 16 / 32 channels (conv1.1, conv1.2: 16 -> 16; conv2.1, conv2.2: 32 -> 32: ConvBnReLU, mvsnet.py:19-20,24-25;
 * smooth1: Conv2d 32 -> 16 with bias, mvsnet.py:32,53) in the split-f16 arithmetic (csrc/conv2d_ci_splitf16.hip).  (cin, cout) in {(16, 16),
 * (32, 32), (32, 16)}, W % 2 == 0, tensors 8-byte aligned.  `packed`: HOST image from casmvs_conv2d_ci_splitf16_pack (weight (cout, cin, 3, 3)
 * finite, scale / shift (cout) or NULL), copied to the device (16-byte aligned).  out_nhwc: NULL or the pixel-major copy (N, H, W, cout),
 * 16-byte aligned; `out` may be NULL when out_nhwc is given (casmvs_fpn_tail0_splitf16_f32: feat0 likewise).  slope: 0.01 for the ABN layers, 1.0 for smooth1. */
size_t casmvs_conv2d_ci_splitf16_packed_bytes(int cin, int cout);
int casmvs_conv2d_ci_splitf16_pack(int cin, int cout, const float *weight, const float *scale, const float *shift, void *packed);
int casmvs_conv2d_ci_splitf16_supported(int cin, int cout, int W);
int casmvs_conv2d_ci_splitf16_forward_f32(const void *packed, const float *in, float *out, float *out_nhwc, int N, int cin, int cout, int H, int W,
                                          float slope, void *stream);
/* FeatureNet's stride-2 layers conv1.0 (8 -> 16) and conv2.0 (16 -> 32): Conv2d k5 s2 p2 + folded ABN + leaky-relu (mvsnet.py:18,23) in the same arithmetic
 * (csrc/conv2d_k5s2_splitf16.hip).  in (N, cin, H, W) -> out (N, cout, H / 2, W / 2); H even, W % 4 == 0, `in` and the image 16-byte aligned.  `packed`: HOST image
 * from casmvs_conv2d_k5s2_splitf16_pack (weight (cout, cin, 5, 5) finite, scale / shift (cout) or NULL), copied to the device by the caller. */
size_t casmvs_conv2d_k5s2_splitf16_packed_bytes(int cin, int cout);
int casmvs_conv2d_k5s2_splitf16_pack(int cin, int cout, const float *weight, const float *scale, const float *shift, void *packed);
int casmvs_conv2d_k5s2_splitf16_supported(int cin, int cout, int H, int W);
int casmvs_conv2d_k5s2_splitf16_forward_f32(const void *packed, const float *in, float *out, int N, int cin, int cout, int H, int W, float slope, void *stream);
/* FeatureNet.conv0 = ConvBnReLU(3, 8, 3) -> ConvBnReLU(8, 8, 3) (models/mvsnet.py:14-16) as ONE kernel, both layers on the f16 matrix cores in the same
 * arithmetic (csrc/fnet_conv0_mm.hip): the 8-channel map between the two layers never reaches memory.  imgs (N, 3, H, W) -> out (N, 8, H, W); W even, both
 * 8-byte aligned.  `packed`: HOST image from casmvs_fnet_conv0_mm_pack (w0 (8, 3, 3, 3), w1 (8, 8, 3, 3) finite; scale / shift (8) = the layers' folded
 * eval-mode ABN, or NULL = 1 / 0), copied to the device by the caller (16-byte aligned).  slope: the ABN leaky-relu slope of both layers. */
size_t casmvs_fnet_conv0_mm_packed_bytes(void);
int casmvs_fnet_conv0_mm_pack(const float *w0, const float *scale0, const float *shift0, const float *w1, const float *scale1, const float *shift1, void *packed);
int casmvs_fnet_conv0_mm_supported(int W);
int casmvs_fnet_conv0_mm_f32(const void *packed, const float *imgs, float *out, int N, int H, int W, float slope, void *stream);
size_t casmvs_fpn_tail0_splitf16_packed_bytes(void);
int casmvs_fpn_tail0_splitf16_pack(const float *weight40, void *packed);
int casmvs_fpn_tail0_splitf16_f32(const void *packed, const float *bias9, const float *conv0, const float *feat1_sum,
                                  float *feat0, float *feat0_nhwc, int N, int H, int W, void *stream);
int casmvs_featurenet_forward_fused_f32(const float *const *packed_layers, const void *fused0_packed, int fused0_arith,
                                        const float *fused0_bias9, const void *const *ci_layers, const float *imgs, float *feat0, float *feat1,
                                        float *feat2, float *feat0_nhwc, float *feat1_nhwc, float *feat2_nhwc,
                                        void *workspace, int N, int H, int W, float slope,
                                        void *const *layer_events, void *stream);

/* ---- (a9) softmax over depth + soft-argmin regression + confidence --------------------------
 * Replaces: models/mvsnet.py:174-193 and models/modules.py:95-104:
 *   p = softmax_D(cost); depth = sum_k p_k d_k; idx = clamp(trunc(sum_k p_k k), 0, D-1);
 *   confidence = p[idx-1] + p[idx] + p[idx+1] + p[idx+2]  (zeros outside [0, D)).
 * cost, depth_values : device (B, D, h, w);  depth, confidence : device (B, h, w)
 * index : device (B, h, w) int32 or NULL (optional debug/parity output of idx).
 */
int casmvs_softmax_regress_f32(const float *cost, const float *depth_values, float *depth,
                               float *confidence, int32_t *index, int B, int D, int h, int w,
                               void *stream);

/* ---- (a8 tail + a9) the `prob` head walking the depth axis, fused with the regression ---------------
 * Replaces: models/mvsnet.py:89,104 (`prob`: Conv3d 8 -> 1, k3 p1, bias) followed by models/mvsnet.py:174-193
 *           (softmax, depth_regression, confidence) - one launch when the whole depth range is one chunk.
 * packed : device image of the head's parameters (casmvs_conv3d_pack_f32, kind S1, cout 1)
 * in     : device (B, 8, D, h, w), 16-byte aligned, w % 4 == 0 (casmvs_prob_regress_supported)
 * cost   : device (B, D, h, w), always written (the regularised cost, mvsnet.py:174), 16-byte aligned
 * depth_values (B, D, h, w), depth / confidence (B, h, w), index (B, h, w) int32 or NULL: as
 *          casmvs_softmax_regress_f32; depth == NULL: only `cost` is produced (all four may then be NULL)
 * slope  : leaky-relu slope of the head (1.0 = none, the reference)
 * zchunk : output planes per workgroup along D; 0 = chosen by the library (the whole range when the pixel
 *          tiles fill the chip: then the regression runs inside the same kernel, else as a second launch).
 */
int casmvs_prob_regress_supported(int cin, int w);
int casmvs_prob_regress_f32(const float *packed, const float *in, const float *depth_values, float *cost,
                            float *depth, float *confidence, int32_t *index, int B, int cin, int D, int h,
                            int w, float slope, int zchunk, void *stream);

#ifdef CASMVS_TRACE
/* Debug tooling, -DCASMVS_TRACE builds only (tools/build_trace_lib.sh; tools/debug/disturber.py, DESIGN.md "co-residency"): a neighbour kernel of a
 * chosen kind - 0 f32 MFMA, 1 f16 MFMA, 2 bf16 MFMA, 3 LDS traffic over its whole allocation, 4 VALU - with `lds_bytes` (16 .. 163840) of dynamic LDS
 * per 256-thread workgroup.  The production library does not contain it. */
int casmvs_debug_disturb(int kind, int blocks, int iters, int lds_bytes, float *sink, void *stream);
#endif

/* CostRegNet's tail as ONE kernel that walks the depth axis (csrc/conv11_prob_zfused.hip): conv11 = ConvTranspose3d(16 -> 8, k3 s2 p1 op1) + ABN +
 * leaky-relu + skip (mvsnet.py:84-86,101), `prob` = Conv3d(8 -> 1, k3 p1, bias) (:89,104) and the softmax / regression / confidence (:174-193).  The
 * 8-channel full-resolution tensor between the two layers never reaches memory.  deconv11_packed: DEVICE copy of casmvs_deconv11_splitf16_pack's
 * image (16-byte aligned); prob_packed: device image of casmvs_conv3d_pack_f32(CASMVS_CONV_S1, 8, 1); in (B, 16, Di, Hi, Wi) (conv9's output), skip
 * (B, 8, 2 Di, 2 Hi, 2 Wi) (conv0's output), depth_values / cost (B, 2 Di, 2 Hi, 2 Wi), depth / confidence (B, 2 Hi, 2 Wi), index: NULL or int32.
 * Wi even; tensors 8-byte aligned.  slope: conv11's leaky-relu slope; prob_slope: 1 (no activation).  casmvs_costreg_regress_f32 selects it where the
 * 16 x 60 pixel tiles fill the chip. */
int casmvs_conv11_prob_zfused_supported(int Di, int Hi, int Wi);
int casmvs_conv11_prob_zfused_f32(const void *deconv11_packed, const float *prob_packed, const float *in, const float *skip, const float *depth_values,
                                  float *cost, float *depth, float *confidence, int32_t *index, int B, int Di, int Hi, int Wi, float slope,
                                  float prob_slope, void *stream);

/* Whole CostRegNet + regression: casmvs_costreg_forward_f32 with the head replaced by casmvs_prob_regress_f32.
 * `cost` (B, D, h, w) is still produced.  layer_events: as casmvs_costreg_forward_f32 (event 10 before the head,
 * event 11 after the head INCLUDING the regression).  conv0_arith selects conv0's arithmetic: CASMVS_CONV0_F32 (the float32
 * MFMA kernel on packed_layers[0]), CASMVS_CONV0_SPLIT_BF16 / CASMVS_CONV0_SPLIT_F16 with split_layers[0] = the DEVICE copy of
 * casmvs_conv0_splitbf16_pack's / casmvs_conv0_splitf16_pack's image of conv0 (cin 8 / 16 / 32; any other shape falls back to the
 * float32 kernel; with the split-f16 image cin = 16 runs casmvs_conv0_zmarch_forward_f32).  split_layers: NULL, or EIGHT pointers { conv0, conv2, conv4,
 * conv6, conv9, conv11, conv1, conv3 images, each or NULL } (ABI version 3; version 2 read six, version 1 four); a non-NULL conv2 / conv4 / conv6 entry (DEVICE copy of
 * casmvs_conv_ci_splitf16_pack's image) runs that layer on the f16 matrix cores too (conv4 / conv6 only where the volume gives >= 100 tiles; conv6
 * only with >= 3 planes), a non-NULL conv9 / conv11 entry (casmvs_deconv9_splitf16_pack / casmvs_deconv11_splitf16_pack) the transposed layers, a non-NULL conv1 / conv3
 * entry (casmvs_conv_s2_splitf16_pack) the stride-2 layers. */
#define CASMVS_CONV0_F32 0
#define CASMVS_CONV0_SPLIT_BF16 1
#define CASMVS_CONV0_SPLIT_F16 2
int casmvs_costreg_regress_f32(const float *const *packed_layers, const void *const *split_layers, int conv0_arith, const float *vol,
                               const float *depth_values, float *cost, float *depth, float *confidence, int32_t *index,
                               void *workspace, int B, int cin, int D, int h, int w, float slope,
                               void *const *layer_events, void *stream);

/* Replaces: models/modules.py:95-104 `depth_regression(p, depth_values)` on its own (the engine's forward uses the
 * fused kernel above): out[b, y, x] = sum_k prob[b, k, y, x] * d, with d = depth_values[b, k, y, x]
 * (depth_values_per_plane == 0, shape (B, D, h, w)) or depth_values[k] (depth_values_per_plane != 0, shape (D)). */
int casmvs_depth_regression_f32(const float *prob, const float *depth_values, float *out, int B, int D,
                                int h, int w, int depth_values_per_plane, void *stream);

/* ---- (f-4) input images ------------------------------------------------------------------------
 * Replaces: datasets/dtu.py:134-137 (T.ToTensor + T.Normalize) on the device: images (N,H,W,3) uint8 RGB ->
 * out (N,3,H,W) float32 = (u8 / 255 - mean[c]) / std[c]; mean3 / std3 are HOST arrays of 3 floats. */
int casmvs_normalize_images_u8(const unsigned char *images, float *out, int N, int H, int W, const float *mean3,
                               const float *std3, void *stream);

/* ---- (f-3) depth filtering / fusion of one reference view --------------------------------------
 * Replaces: eval.py:113-182 (xy_ref2src, xy_src2ref, check_geo_consistency) and eval.py:273-326 (confidence mask,
 * geometric-consistency count, depth / colour averaging, back-projection) for ONE reference view against S source
 * views; the surrounding scan loop / PFM / PLY I/O stays on the host.  All maps are full resolution (H, W).
 * depth_ref (H,W) f32; image_ref (H,W,3) u8; proba_quarter (H/4,W/4) f32 = confidence_2, or NULL (no confidence mask);
 * depth_src (S,H,W) f32; image_src (S,H,W,3) u8; m_ref2src / m_src2ref (S,3,4) f32 = (P_src inv(P_ref))[:3] /
 * (P_ref inv(P_src))[:3]; m_ref2world (3,4) = rows of inv(P_ref) or NULL.
 * Outputs: depth_refined (H,W) f32; image_refined (H,W,3) f64; mask_geo_sum (H,W) i32; mask_final (H,W) u8 =
 * (proba > conf) & (mask_geo_sum >= min_geo_consistent); optional xyz_world (H,W,3) f32 for every pixel, mask_geo
 * (S,H,W) u8, depth_reproj (S,H,W) f32 and image_s2r (S,H,W,3) u8 (the per-view results of check_geo_consistency). */
int casmvs_fuse_reference_view(const float *depth_ref, const unsigned char *image_ref, const float *proba_quarter,
                               const float *depth_src, const unsigned char *image_src, const float *m_ref2src,
                               const float *m_src2ref, const float *m_ref2world, float *depth_refined,
                               double *image_refined, int32_t *mask_geo_sum, unsigned char *mask_final,
                               float *xyz_world, unsigned char *mask_geo, float *depth_reproj, unsigned char *image_s2r,
                               int S, int H, int W, float conf, int min_geo_consistent, void *stream);

/* The same function with a quarter of the vector-memory instructions: the two taps of a row come from one 8-byte load (depth
 * pair; six colour bytes), the view matrices from LDS.  Same expressions in the same order - results are bit-identical to
 * casmvs_fuse_reference_view (checked on the MI355X by tools/native/fusion_check.cpp: every output incl. the per-view ones; 95 -> 83 us
 * at 1152 x 864 with 10 source views).  Needs W, H >= 2 and S <= 64.  What casmvsnet_pl_amd.fusion calls by default. */
int casmvs_fuse_reference_view_paired(const float *depth_ref, const unsigned char *image_ref, const float *proba_quarter,
                                      const float *depth_src, const unsigned char *image_src, const float *m_ref2src,
                                      const float *m_src2ref, const float *m_ref2world, float *depth_refined,
                                      double *image_refined, int32_t *mask_geo_sum, unsigned char *mask_final,
                                      float *xyz_world, unsigned char *mask_geo, float *depth_reproj, unsigned char *image_s2r,
                                      int S, int H, int W, float conf, int min_geo_consistent, void *stream);

/* ---- (f-2) backward of the plane sweep and of the depth regression (training, op by op) -----------
 * casmvs_homo_warp_backward_f32: the gradient of models/modules.py:52-92 with respect to src_feat (the grid depends on
 *   the DETACHED depth hypotheses only, mvsnet.py:231): grad_src (B,C,H,W) = scatter-add of grad_out (B,C,D,H,W) with the
 *   forward's bilinear weights.  grad_src is zeroed by the call.  The sums are 64-bit fixed point with one scale per (sample, channel)
 *   (csrc/fixed_accum.h): the result is bit-identical run to run (ABI 4; ABI 3 used float atomics).  workspace: caller-owned, 16-byte aligned,
 *   casmvs_homo_warp_backward_workspace_bytes(B, C, D, H, W) bytes (the fixed-point map + the channels' largest magnitudes); contents need not survive the call.
 * casmvs_softmax_regress_backward_f32: depth = sum_k softmax(cost)_k d_k (mvsnet.py:175-177): grad_cost (B,D,h,w) =
 *   grad_depth (B,h,w) * p_k (d_k - depth).  (The confidence is computed under no_grad in the reference.) */
size_t casmvs_homo_warp_backward_workspace_bytes(int B, int C, int D, int H, int W);
int casmvs_homo_warp_backward_f32(const float *grad_out, const float *proj, const float *depth, float *grad_src, void *workspace,
                                  int B, int C, int H, int W, int D, void *stream);
int casmvs_softmax_regress_backward_f32(const float *cost, const float *depth_values, const float *grad_depth,
                                        float *grad_cost, int B, int D, int h, int w, void *stream);

/* ---- (f-2) training kernels: weight / input gradients, train-mode ABN, FPN step, cost-volume gradient -----------
 * What `train.py` (train.py:99-127: forward in train mode, loss.backward()) needs beyond the inference engine.  The host
 * side (casmvsnet_pl_amd/training.py) wires them into torch.autograd.Function objects.
 *
 * casmvs_conv_wgrad_f32: gradient of any convolution kind of the model (CASMVS_CONV_S1 / S2 / T2, CASMVS_CONV2D_K3 / K5S2 /
 *   K1) with respect to its weight, on the matrix cores.  `in` is the layer's input (B,cin,D,H,W) (2D kinds: D = 1),
 *   `grad_out` the gradient of its output; grad_weight has the torch layout of the kind: Conv (cout,cin,taps),
 *   ConvTranspose3d (cin,cout,27).  `workspace` (device, casmvs_conv_wgrad_workspace_bytes) holds per-workgroup partial
 *   sums that are added in a fixed order: reproducible, no atomics.
 * casmvs_conv_dgrad_direct_f32: gradient with respect to the INPUT for Conv kinds, straight from the definition (one thread per
 *   input element).  Only for the layer shapes whose adjoint is not itself a forward layer of the MFMA engine (Conv2d
 *   k5 s2, 1x1 with 8 input channels): every other input gradient is casmvs_conv{2,3}d_forward_f32 with adjoint weights.
 *   `weight` (cout,cin,taps) on the device; (D,H,W) = the layer's input dims.
 * casmvs_channel_sums_f64: out (C, blocks, 2) doubles = per-workgroup partial (sum x, sum x^2) of x (N,C,n) per channel,
 *   blocks = casmvs_channel_sums_blocks(N, n): batch statistics of ABN / BatchNorm in train mode; bias gradients.
 * casmvs_abn_apply_f32: y = leaky_relu(x * scale[c] + shift[c]) (scale = gamma * rstd, shift = beta - mean * scale).
 * casmvs_abn_backward_sums_f64 / casmvs_abn_backward_apply_f32: with g = grad_y * leaky_relu'(y) and xhat = (x - mean) * rstd:
 *   sums (C, blocks, 2) = partial (sum g, sum g xhat) [= grad_beta, grad_gamma]; grad_x = scale * (g - m1 - xhat * m2) with
 *   m1 = sum g / M, m2 = sum g xhat / M (device vectors of C floats).
 * casmvs_pack_gather_f32: out[i] = [weight (n_weight), bias (n_bias), 0, 1][index[i]]: the packed layer image of
 *   casmvs_conv{2,3}d_pack_f32 as a gather on the device (index: n_out int32, derived once per layer shape by the caller).
 * casmvs_abn_train_finish_f32: one launch from the partial sums of casmvs_channel_sums_f64 to the layer's per-channel vectors (C
 *   floats each, device): mean, rstd = 1 / sqrt(biased var + eps), scale = gamma * rstd, shift = bias - mean * scale, and the
 *   in-place momentum update of running_mean / running_var (unbiased variance, like F.batch_norm; both NULL = skip).
 *   count = N * n elements per channel.  abs_eps >= 0: gamma = |weight| + abs_eps (InPlaceABN), otherwise gamma = weight.
 * casmvs_abn_backward_finish_f32: from the partial sums of casmvs_abn_backward_sums_f64 to grad_bias = sum g,
 *   grad_weight = sum g xhat (times sign(weight) when abs_eps >= 0), m1 = sum g / count, m2 = sum g xhat / count.
 * casmvs_upsample2x_add_f32 / casmvs_upsample2x_backward_f32: out (N,C,H,W) = lat + bilinear x2 (align_corners = True) of
 *   up (N,C,H/2,W/2) (mvsnet.py:36-38); grad_up = the transpose of the interpolation applied to grad_out (a gather).
 * casmvs_costvol_var_backward_f32: gradient of the variance volume (mvsnet.py:137-167) w.r.t. feats (B,V,C,h,w) given
 *   grad_vol (B,C,D,h,w): d var / d x_v = 2 x_v / V - 2 sum_v x_v / V^2 through the plane sweep's bilinear weights
 *   (reference view: no warp).  grad_feats is zeroed by the call; the hypotheses get no gradient (mvsnet.py:231).  C: a multiple of
 *   4 up to 64, V <= 64.  ORDER-INDEPENDENT (ABI 4): a workgroup accumulates its scatter in a 64-bit fixed-point LDS image (LDS float atomics retire
 *   lane by lane on gfx950) and adds the image to a 64-bit fixed-point gradient map with integer atomics; one scale per (sample, channel) from the
 *   channel's largest finite |grad_vol| and |feats| (a strict bound of every contribution: no range check, no fallback); a last pass rounds the sums to
 *   float32 once.  The result is bit-identical run to run (train.py:99-127 is reproducible); non-finite contributions poison exactly the elements a float
 *   accumulation would (csrc/fixed_accum.h, csrc/train.hip).  workspace: caller-owned, 16-byte aligned, casmvs_costvol_backward_workspace_bytes(B, V, C,
 *   G, D, h, w) bytes (G = 0 for the variance volume); contents need not survive the call. */
size_t casmvs_conv_wgrad_workspace_bytes(int kind, int B, int cin, int cout, int D, int H, int W);
int casmvs_conv_wgrad_f32(int kind, const float *in, const float *grad_out, float *grad_weight, void *workspace, int B,
                          int cin, int cout, int D, int H, int W, void *stream);

/* Weight gradient of the `prob` layer (Conv3d 8 -> 1, k3 s1 p1; models/mvsnet.py:89) on its own kernel: grad_weight (1, 8, 3, 3, 3)
 * = sum over (b, z, y, x) of grad_out (B, 1, D, H, W) * in (B, 8, D, H, W) shifted by the tap, zero padded.  The generic
 * casmvs_conv_wgrad_f32 pads the single output channel to a 16-row matrix tile; this one keeps 108 accumulators per thread on the
 * vector ALU (csrc/prob_wgrad.hip), deterministic (fixed-order reduction through `workspace`, casmvs_prob_wgrad_workspace_bytes).
 * Needs W % 4 == 0, 16-byte aligned tensors, 8 D H W < 2^29; casmvs_conv_wgrad_f32 routes the layer here when that holds. */
int casmvs_prob_wgrad_supported(int B, int D, int H, int W);
size_t casmvs_prob_wgrad_workspace_bytes(int B, int D, int H, int W);
int casmvs_prob_wgrad_f32(const float *in, const float *grad_out, float *grad_weight, void *workspace, int B, int D, int H, int W,
                          void *stream);
int casmvs_conv_dgrad_direct_f32(int kind, const float *weight, const float *grad_out, float *grad_in, int B, int cin,
                                 int cout, int D, int H, int W, void *stream);
int casmvs_channel_sums_blocks(int N, size_t n);
int casmvs_channel_sums_f64(const float *x, double *out, int N, int C, size_t n, void *stream);
int casmvs_abn_apply_f32(const float *x, const float *scale, const float *shift, float *y, int N, int C, size_t n, float slope,
                         void *stream);
int casmvs_abn_backward_sums_f64(const float *grad_y, const float *y, const float *x, const float *mean, const float *rstd,
                                 double *sums, int N, int C, size_t n, float slope, void *stream);
int casmvs_abn_backward_apply_f32(const float *grad_y, const float *y, const float *x, const float *scale, const float *mean,
                                  const float *rstd, const float *m1, const float *m2, float *grad_x, int N, int C, size_t n,
                                  float slope, void *stream);
int casmvs_pack_gather_f32(const float *weight, const float *bias, const int *index, float *out, int n_weight, int n_bias,
                           int n_out, void *stream);
/* The same gather for many layer images in ONE launch (training: every image of a step; casmvsnet_pl_amd/training.py's pack plan).  `segments`: DEVICE
 * array; segment s is gathered by the workgroups first_block .. first_block + ceil(n_out / 256) - 1 (ascending, no gaps; n_blocks = their total). */
typedef struct casmvs_pack_segment {
  const float *weight, *bias; /* bias may be NULL when n_bias == 0 */
  const int *index;
  float *out;
  int n_weight, n_bias, n_out, first_block;
} casmvs_pack_segment;
int casmvs_pack_gather_batch_f32(const casmvs_pack_segment *segments, int n_segments, int n_blocks, void *stream);
int casmvs_abn_train_finish_f32(const double *sums, int blocks, int C, double count, const float *weight, const float *bias,
                                float abs_eps, float eps, float momentum, float *running_mean, float *running_var, float *scale,
                                float *shift, float *mean, float *rstd, void *stream);
int casmvs_abn_backward_finish_f32(const double *sums, int blocks, int C, double count, const float *weight, float abs_eps,
                                   float *grad_weight, float *grad_bias, float *m1, float *m2, void *stream);
/* The two per-channel epilogues folded into the elementwise passes (what casmvsnet_pl_amd/training.py calls): every workgroup reduces the channel's
 * partial sums itself; casmvs_abn_train_apply_f32 = casmvs_abn_train_finish_f32 + casmvs_abn_apply_f32, casmvs_abn_backward_apply_fused_f32 =
 * casmvs_abn_backward_finish_f32 + casmvs_abn_backward_apply_f32 (grad_weight / grad_bias written, m1 / m2 internal): one launch less per layer and pass. */
int casmvs_abn_train_apply_f32(const float *x, const double *sums, int blocks, double count, const float *weight, const float *bias, float abs_eps,
                               float eps, float momentum, float *running_mean, float *running_var, float *scale, float *shift, float *mean,
                               float *rstd, float *y, int N, int C, size_t n, float slope, void *stream);
int casmvs_abn_backward_apply_fused_f32(const float *grad_y, const float *y, const float *x, const double *sums, int blocks, double count,
                                        const float *weight, float abs_eps, const float *scale, const float *mean, const float *rstd,
                                        float *grad_weight, float *grad_bias, float *grad_x, int N, int C, size_t n, float slope, void *stream);
int casmvs_upsample2x_add_f32(const float *lat, const float *up, float *out, int N, int C, int H, int W, void *stream);
int casmvs_upsample2x_backward_f32(const float *grad_out, float *grad_up, int N, int C, int H, int W, void *stream);
size_t casmvs_costvol_backward_workspace_bytes(int B, int V, int C, int G, int D, int h, int w);
int casmvs_costvol_var_backward_f32(const float *feats, const float *proj, const float *depth, const float *grad_vol,
                                    float *grad_feats, void *workspace, int B, int V, int C, int h, int w, int D, void *stream);
/* The same for the group-wise correlation volume (mvsnet.py:142-144,157-162,169-172): grad_vol (B,G,D,h,w), G divides C.  One launch instead of a
 * channel-expanded gradient volume, one warp backward and one recomputed warped volume per source view. */
int casmvs_costvol_gwc_backward_f32(const float *feats, const float *proj, const float *depth, const float *grad_vol, float *grad_feats, void *workspace,
                                    int B, int V, int C, int G, int h, int w, int D, void *stream);

/* ---- self test ------------------------------------------------------------------------------
 * Runs the MFMA lane-mapping probes the conv kernels rely on (v_mfma_f32_16x16x4_f32 operand /
 * result layout, and v_mfma_f32_4x4x1_16b_f32 with A-block broadcast).  Returns 0 when the hardware semantics match the kernels' assumptions.
 * Needs a gfx950 device.  `dump` (host, 64*4*4 floats, may be NULL) receives raw probe outputs.
 */
int casmvs_selftest_mfma(float *dump);

/* Issue-rate probe: `blocks` workgroups x 4 wavefronts x `iters` x 16 back-to-back MFMAs on
 * independent accumulators; writes the achieved TFLOP/s.
 * shape: 0 = v_mfma_f32_4x4x1_16b_f32 (cbsz 4), 1 = 16x16x4_f32, 2 = 32x32x2_f32, 3 = 16x16x1_4b_f32,
 * 4 = 16x16x4_f32 interleaved with one ds_read_b32 per MFMA (the conv inner loop in isolation). */
int casmvs_selftest_mfma_rate(int shape, int blocks, int iters, float *tflops);

#ifdef __cplusplus
}
#endif
#endif /* CASMVS_H */
