/*
 * ganet_hip.h -- C ABI of libganet_hip.so: GA-Net's guided-aggregation hot path
 * (SGA, LGA, GetCostVolume, DisparityRegression) as hand-written HIP kernels for
 * AMD Instinct MI355X (gfx950 / CDNA4).
 *
 * This is the drop-in boundary for the reference's native layer: every entry
 * point below names the reference interface it replaces (paths relative to the
 * reference repository root).  Plain pointers and sizes only -- no torch types.
 *
 * Conventions
 *   - All pointers are DEVICE pointers to contiguous fp32 (or uint8 where stated)
 *     buffers owned by the caller; the library never allocates tensor storage.
 *   - Volumes are [N,C,D,H,W]; guidance [N,C,5,H,W] (taps w0..w4, L1-normalised by
 *     the caller as in models/GANet_deep.py:264-268); LGA input [B,D,H,W] with
 *     filters [B,3*(2r+1)^2,H,W] (B = N for the 4-D ops, N*C for the "3d" ops).
 *   - dir: 0 down, 1 up, 2 right, 3 left (the reference's mask values,
 *     libs/GANet/src/GANet_kernel.cu:964-994).
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 *     Work is enqueued asynchronously; nothing here synchronises the device.
 *   - Return value: 0 on success, negative on error (GANET_E_*); a message is
 *     available from ganet_last_error() (thread-local).  The reference returns 1
 *     unconditionally and never checks the runtime (GANet_cuda.cpp:5-64).
 */
#ifndef GANET_HIP_H
#define GANET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANET_OK 0
#define GANET_E_INVALID (-1)      /* null pointer, non-positive size, bad dir      */
#define GANET_E_UNSUPPORTED (-2)  /* shape outside the compiled kernel set         */
#define GANET_E_RUNTIME (-3)      /* HIP runtime / launch error                    */

#define GANET_ABI_VERSION 11      /* v11 = v10 + ganet_residual_relu_forward / _backward (SGABlock's residual epilogue) */
int ganet_abi_version(void);
const char *ganet_last_error(void);
/* 1 if this build runs the lockstep CPU emulator (tests only), 0 for the gfx950 build */
int ganet_is_simulator(void);

/* ---------------------------------------------------------------- SGA ------------ */

/* One directional volume A_dir = scan_dir(x, g).
 * Replaces: memcpy + sga_{down,up,right,left}_forward<<<>>>
 *           (libs/GANet/src/GANet_kernel.cu:66-127, 285-346, 507-565, 720-778). */
int ganet_sga_scan_forward(const float *x, const float *g, float *A,
                           int N, int C, int D, int H, int W, int dir, void *stream);

/* Fast-path forward used by ganet_amd's SgaFunction: the four directional volumes
 * are written to A_ws ([4][N*C*D*H*W] floats, caller-allocated; kept for backward),
 * out = element-wise max in reference order, mask = winning direction (uint8), and
 * kp ([4][N*C*H*W] uint16) = first-argmax over d of every directional volume.
 * Replaces: sga_kernel_forward (GANet_kernel.cu:935-998) incl. the three `Max`
 * launches (:23-36) and five device memcpys; kp is what MaxDepth (:50-64) recomputes
 * four times in the reference's backward. */
int ganet_sga_forward(const float *x, const float *g0, const float *g1, const float *g2,
                      const float *g3, float *A_ws, float *out, uint8_t *mask, uint16_t *kp,
                      int N, int C, int D, int H, int W, void *stream);

/* The steps of ganet_sga_forward / ganet_sga_backward as they run inside those calls: direction `dir`'s scan into its place in
 * the op's PRIVATE workspace (A_ws / G_ws: [4][N*C*D*H*W] floats; kp: the whole [4][N*C*H*W] array).  A_ws holds the four
 * directional volumes in the API layout.  G_ws is not in the API layout in general: ganet_sga_workspace_layout() returns 1 if
 * the two vertical directions' adjoint volumes are tiled for these dimensions (element (s, d, h, w) of G_ws[dir < 2] at
 * ((((s * W/16 + w/16) * H/4 + h/4) * D + d) * 4 + h%4) * 16 + w%16), 0 if everything has the API layout.
 * ganet_sga_backward_scan (above) always writes the API layout.
 * Replaces: the same reference code as ganet_sga_scan_forward / ganet_sga_backward_scan. */
int ganet_sga_scan_forward_ws(const float *x, const float *g, float *A_ws,
                              int N, int C, int D, int H, int W, int dir, void *stream);
int ganet_sga_backward_scan_ws(const float *g, const uint8_t *mask, const uint16_t *kp, const float *grad_out, float *G_ws,
                               int N, int C, int D, int H, int W, int dir, void *stream);
int ganet_sga_workspace_layout(int N, int C, int D, int H, int W);

/* The last step of ganet_sga_forward on its own: out / mask / kp from the four directional volumes in A_ws (same buffers, same
 * layouts).  ganet_sga_forward == 4 x ganet_sga_scan_forward_ws + this; exported so that a caller (bench.py) can time it in place.  A_ws as ganet_sga_scan_forward_ws leaves it.
 * Replaces: the three `Max` launches (GANet_kernel.cu:23-36, 964-994) and the four MaxDepth launches of the backward (:50-64). */
int ganet_sga_merge(const float *A_ws, float *out, uint8_t *mask, uint16_t *kp,
                    int N, int C, int D, int H, int W, void *stream);

/* Inference-only forward: out = relu(bn_scale[c] * max_dir A_dir + bn_shift[c]) (bn_scale = bn_shift = NULL:
 * plain max).  No mask / arg-max is produced; A_ws ([4][N*C*D*H*W]) is scratch and stays untouched when the scans can take
 * the running maximum themselves (the usual case: W % 4 == 0, D <= 208, 16-byte aligned buffers).
 * Replaces: sga_kernel_forward (GANet_kernel.cu:935-998) followed by the eval-mode BatchNorm3d + ReLU of
 * SGABlock.forward (models/GANet_deep.py:269-271: `x = self.SGA(...); x = self.bn_relu(x)`), with
 * bn_scale = weight / sqrt(running_var + eps), bn_shift = bias - running_mean * bn_scale. */
int ganet_sga_forward_infer(const float *x, const float *g0, const float *g1, const float *g2,
                            const float *g3, float *A_ws, float *out, const float *bn_scale,
                            const float *bn_shift, int N, int C, int D, int H, int W, void *stream);
/* How many volume-sized ([N*C*D*H*W] floats) scratch buffers ganet_sga_forward_infer needs behind A_ws for these
 * arguments: 0 (the scans take the running maximum themselves; A_ws may then be NULL) or 4.  Negative on error. */
int ganet_sga_forward_infer_scratch(const float *x, const float *g0, const float *g1, const float *g2,
                                    const float *g3, const float *out, int N, int C, int D, int H, int W);

/* Reverse-scan adjoint of one direction: G = [mask == dir] * grad_out propagated through the
 * recurrence with first-argmax routing (kp_dir: [N*C*H*W] uint16 of that direction).
 * Replaces: cudaMemset + get_temp_grad (:38-48) + the top_diff part of
 *           sga_*_data_backward (:144-181 & mirrors). */
int ganet_sga_backward_scan(const float *g, const uint8_t *mask, const uint16_t *kp_dir,
                            const float *grad_out, float *G,
                            int N, int C, int D, int H, int W, int dir, void *stream);

/* One direction of the backward pass: adjoint scan into G_ws ([N*C*D*H*W] scratch) followed by
 * the per-pixel gradient kernel.  grad_x: written (accumulate = 0) or accumulated into
 * (accumulate = 1); gw ([N,C,5,H,W]) is written.
 * Replaces: cudaMemset + get_temp_grad + MaxDepth + sga_*_data_backward (:129-208 & mirrors)
 *           + sga_*_weight_backward (:210-281 & mirrors). */
int ganet_sga_backward_dir(const float *x, const float *g, const float *A, const uint8_t *mask,
                           const uint16_t *kp_dir, const float *grad_out, float *G_ws,
                           float *grad_x, float *gw,
                           int N, int C, int D, int H, int W, int dir, int accumulate,
                           void *stream);

/* Fast-path backward: four adjoint scans into G_ws ([4][N*C*D*H*W] scratch) + ONE per-pixel
 * kernel for all gradients, from the volumes / mask / kp saved by ganet_sga_forward.
 * grad_x and gw0..gw3 are fully overwritten.
 * Replaces: sga_kernel_backward (GANet_kernel.cu:1000-1129). */
int ganet_sga_backward(const float *x, const float *g0, const float *g1, const float *g2,
                       const float *g3, const float *A_ws, const uint8_t *mask, const uint16_t *kp,
                       const float *grad_out, float *G_ws, float *grad_x, float *gw0, float *gw1,
                       float *gw2, float *gw3, int N, int C, int D, int H, int W, void *stream);

/* The last step of ganet_sga_backward on its own: every gradient from the four adjoint volumes in G_ws (written by
 * ganet_sga_backward_scan_ws, in the layout ganet_sga_workspace_layout reports for these dimensions -- NOT by
 * ganet_sga_backward_scan, whose G has the API layout) and the forward volumes in A_ws.  On 16-byte aligned volumes
 * ganet_sga_backward == 4 x ganet_sga_backward_scan_ws + this; ganet_sga_backward itself also accepts a contiguous gradient /
 * guidance at a 4-byte aligned address (it then keeps the API layout internally), the _ws entries return GANET_E_UNSUPPORTED
 * for that on a tiled shape.
 * Replaces: the bottom_diff part of sga_*_data_backward (:182-207 & mirrors) + sga_*_weight_backward (:210-281 & mirrors). */
int ganet_sga_backward_point(const float *x, const float *g0, const float *g1, const float *g2, const float *g3,
                             const float *A_ws, const float *G_ws, float *grad_x, float *gw0, float *gw1,
                             float *gw2, float *gw3, int N, int C, int D, int H, int W, void *stream);

/* Reference-compatible buffer contract, for callers that keep the reference's
 * libs/GANet/functions/GANet.py unchanged:
 *   sga_cuda_forward(input, g0..g3, temp_out, output, mask)      GANet_cuda.cpp:39-48
 *   sga_cuda_backward(input, g0..g3, temp_out, mask, max_idx, gradOutput, temp_grad,
 *                     gradInput, grad0..grad3)                    GANet_cuda.cpp:50-64
 * mask is float-valued; temp_out ends the forward holding A_left and is reused as
 * scratch by the backward (which recomputes the other three volumes); temp_grad and
 * max_idx are scratch (adjoint volume / per-pixel arg-max, as in the reference);
 * gradInput is ACCUMULATED into (the caller zero-fills it, functions/GANet.py:33);
 * grad0..3 are written. */
int ganet_sga_forward_compat(const float *x, const float *g0, const float *g1, const float *g2,
                             const float *g3, float *temp_out, float *out, float *mask_f32,
                             int N, int C, int D, int H, int W, void *stream);
int ganet_sga_backward_compat(const float *x, const float *g0, const float *g1, const float *g2,
                              const float *g3, float *temp_out, const float *mask_f32,
                              float *max_idx, const float *grad_out, float *temp_grad,
                              float *grad_x, float *gw0, float *gw1, float *gw2, float *gw3,
                              int N, int C, int D, int H, int W, void *stream);

/* ---------------------------------------------------------------- LGA ------------ */

/* One LGA pass, y fully overwritten.  radius in {1,2,3}.
 * Replaces: lga_cuda_forward / lga3d_cuda_forward (GANet_cuda.cpp:14-37) ->
 *           lga_filtering_forward (GANet_kernel.cu:1131-1175).  The reference
 *           accumulates into a zero-filled output; here the zero-fill is not needed. */
int ganet_lga_forward(const float *x, const float *f, float *y,
                      int B, int D, int H, int W, int radius, void *stream);

/* One LGA pass that also reduces its output over the disparity axis: snorm [B,H,W] = sum_d |y|, sdy [B,H,W] = sum_d d * y,
 * so that F.normalize(y, p=1, dim=1) followed by DisparityRegression (the end of DispAgg.forward, models/GANet_deep.py:246-247)
 * is sdy / max(snorm, 1e-12) -- a per-pixel finish instead of two more passes over the volume.  y may be NULL (inference:
 * the volume itself is not needed); with y given it is written as by ganet_lga_forward (training keeps it for the backward,
 * ganet_norm_disparity_regression_backward takes it together with out and snorm).  GANET_E_UNSUPPORTED where the fused kernel
 * does not apply (radius 3, planes of 2^28 pixels or more): run the two separate entries instead. */
int ganet_lga_forward_regress(const float *x, const float *f, float *y, float *snorm, float *sdy,
                              int B, int D, int H, int W, int radius, void *stream);

/* ABI v7.  One LGA pass (transposed = 0: as ganet_lga_forward) or its data-backward (transposed = 1: x is the output gradient,
 * y the input gradient, as the second half of ganet_lga_backward) on volumes in the PAIR-INTERLEAVED layout
 * [B][ceil(D/2)][H][W][2] -- planes 2m and 2m+1 of a pixel adjacent; for odd D the odd half of the last pair is zero (written
 * so by this entry when y is interleaved; required of x when x is interleaved).  x_paired / y_paired select the layout of
 * either side; at most one of them (neither: the API-layout pass or data-backward on its own).  The layout is for volumes that never cross the operator API: the
 * intermediate between the two passes of an LGA2 (functions/GANet.py:176-187) and its gradient -- a consumer stages a plane
 * pair with two 16-byte copies per lane instead of seven 4-byte ones, a producer stores a pair with one 8-byte store.
 * radius 2 only, W even, 16-byte aligned volumes; GANET_E_UNSUPPORTED otherwise (use the API-layout entries). */
int ganet_lga_apply_paired(const float *x, const float *f, float *y, int B, int D, int H, int W, int radius,
                           int transposed, int x_paired, int y_paired, void *stream);
/* The same with the per-pixel EDGE SUMS of the filters as a side channel, edge: [B][3][H][W] floats (caller-allocated, private to
 * the op): the centre coefficient of the spatially replaced taps and the in-range sums of the two outer depth slabs, which
 * depend on f only.  transposed == 0 (a forward pass): WRITTEN as a by-product (the pass gathers those taps anyway);
 * transposed == 1 (the data-backward of the same filters): READ, instead of gathering 50 - 75 own taps per pixel for them.
 * Lga2Function's forward fills it once, its two data-backward launches use it. */
int ganet_lga_apply_paired_edges(const float *x, const float *f, float *y, float *edge, int B, int D, int H, int W, int radius,
                                 int transposed, int x_paired, int y_paired, void *stream);

/* ABI v7.  The filter gradient of one LGA pass (the first half of ganet_lga_backward: lga_filter_backward,
 * GANet_kernel.cu:1177-1216) with x OR gy in the pair-interleaved layout (see ganet_lga_apply_paired; at most one of
 * x_paired / gy_paired, neither: the API layout); gf keeps the API layout and is written (accumulate_gf = 0) or accumulated
 * into.  radius 2 only, W even, 16-byte aligned volumes. */
int ganet_lga_filter_grad_paired(const float *x, const float *gy, float *gf, int B, int D, int H, int W, int radius,
                                 int accumulate_gf, int x_paired, int gy_paired, void *stream);

/* One LGA pass backward: gx fully overwritten; gf written (accumulate_gf = 0) or
 * accumulated into (accumulate_gf = 1, what chained LGA2/LGA3 rely on,
 * functions/GANet.py:197-199).  gx may alias x (the reference's chained
 * backward does); it must not alias gy.
 * Replaces: lga_cuda_backward / lga3d_cuda_backward (GANet_cuda.cpp:5-28) ->
 *           lga_filter_backward (:1177-1216) + cudaMemset + lga_data_backward
 *           (:1218-1269). */
int ganet_lga_backward(const float *x, const float *f, const float *gy, float *gx, float *gf,
                       int B, int D, int H, int W, int radius, int accumulate_gf, void *stream);

/* ------------------------------------------- GetCostVolume / DisparityRegression -- */

/* cost [N,2C,Dn,H,W] from x,y [N,C,H,W]; Dn = maxdisp + 1.
 * Replaces: GetCostVolume.forward (libs/GANet/modules/GANet.py:119-134) and its
 * autograd-derived backward. */
int ganet_cost_volume_forward(const float *x, const float *y, float *cost,
                              int N, int C, int Dn, int H, int W, void *stream);
int ganet_cost_volume_backward(const float *grad_cost, float *grad_x, float *grad_y,
                               int N, int C, int Dn, int H, int W, void *stream);

/* out [N,H,W] = sum_d d * x[N,Dn,H,W].
 * Replaces: DisparityRegression.forward (libs/GANet/modules/GANet.py:142-148). */
int ganet_disparity_regression_forward(const float *x, float *out,
                                       int N, int Dn, int H, int W, void *stream);
int ganet_disparity_regression_backward(const float *grad_out, float *grad_x,
                                        int N, int Dn, int H, int W, void *stream);

/* ------------------------------------------------- callers' normalisations ---- */

/* L1 normalisation over a strided channel axis, F.normalize(x, p=1, dim) = x / max(sum|x|, 1e-12).
 * x [N][G][C][K][H][W]  ->  y_g [N][C][K][H][W] for g < G (G <= 4; unused outputs may be NULL).
 * G = 4, K = 5 replaces SGABlock's torch.split + view + 4 x F.normalize(p=1, dim=2) of the guidance
 * (models/GANet_deep.py:263-268; ~16 stock kernels); G = C = 1, K = 3(2r+1)^2 replaces
 * F.normalize(g, p=1, dim=1) of the LGA filters (models/GANet_deep.py:235).
 * backward: grad_x [N][G][C][K][H][W] from x and the G gradients wrt y_g (autograd of the same ops). */
int ganet_l1_normalize_forward(const float *x, float *y0, float *y1, float *y2, float *y3,
                               int N, int G, int C, int K, int H, int W, void *stream);
int ganet_l1_normalize_backward(const float *x, const float *gy0, const float *gy1,
                                const float *gy2, const float *gy3, float *grad_x,
                                int N, int G, int C, int K, int H, int W, void *stream);

/* out [N,H,W] = sum_d d * x[N,Dn,H,W] / s,  s = snorm [N,H,W] = max(sum_d |x|, 1e-12).
 * Replaces: F.normalize(x, p=1, dim=1) followed by DisparityRegression at the end of DispAgg.forward
 * (models/GANet_deep.py:246-247): one pass over the volume instead of five.  backward needs x, out, snorm. */
int ganet_norm_disparity_regression_forward(const float *x, float *out, float *snorm,
                                            int N, int Dn, int H, int W, void *stream);
int ganet_norm_disparity_regression_backward(const float *x, const float *out, const float *snorm,
                                             const float *grad_out, float *grad_x,
                                             int N, int Dn, int H, int W, void *stream);

/* y = softmax over the Dn axis of -x, x [N,Dn,H,W]  (nn.Softmin(dim=1) between the two LGA calls of DispAgg.forward,
 * models/GANet_deep.py:244); backward from y and grad_y. */
int ganet_softmin_forward(const float *x, float *y, int N, int Dn, int H, int W, void *stream);
int ganet_softmin_backward(const float *y, const float *grad_y, float *grad_x,
                           int N, int Dn, int H, int W, void *stream);

/* out [N,H,W] = sum_d d * softmin_d(x), x [N,Dn,H,W]: Softmin(dim=1) followed by DisparityRegression as in Disp.forward
 * (models/GANet_deep.py:217-219) without materialising the probabilities; mx, ssum [N,H,W] (running max of -x and
 * the sum of exponentials) are kept for the backward, which recomputes them from x. */
int ganet_softmin_regression_forward(const float *x, float *out, float *mx, float *ssum,
                                     int N, int Dn, int H, int W, void *stream);
int ganet_softmin_regression_backward(const float *x, const float *out, const float *mx, const float *ssum,
                                      const float *grad_out, float *grad_x,
                                      int N, int Dn, int H, int W, void *stream);

/* y [S,Do,Ho,Wo] = F.interpolate(x [S,Di,Hi,Wi], size=[Do,Ho,Wo], mode='trilinear', align_corners=False) per slice
 * (S = N * C), and its adjoint.  Replaces: the up-sampling of the cost volume in Disp.forward / DispAgg.forward
 * (models/GANet_deep.py:212, 240), i.e. ATen's upsample_trilinear3d and -- the point -- upsample_trilinear3d_backward,
 * which scatters with eight atomicAdds per output element; the backward here is a gather per input voxel. */
int ganet_trilinear_upsample_forward(const float *x, float *y, int S, int Di, int Hi, int Wi,
                                     int Do, int Ho, int Wo, void *stream);
int ganet_trilinear_upsample_backward(const float *grad_y, float *grad_x, int S, int Di, int Hi, int Wi,
                                      int Do, int Ho, int Wo, void *stream);

/* The end of SGABlock.forward (models/GANet_deep.py:270-277): behind `conv_refine` (Conv3d + BatchNorm3d, no ReLU) the
 * block adds its input back and applies ReLU,  `x += rem; return relu(x)`.  One pass:
 *   y[n,c,d,h,w] = relu(bn_scale[c] * t + bn_shift[c] + rem)     t = the convolution's output, (bn_scale, bn_shift) = the
 *                  BatchNorm3d folded from its running statistics (eval mode);
 *   bn_scale == bn_shift == NULL:  y = relu(t + rem)             t = bn(conv(..)) as the framework computed it (training).
 * y may alias t (the reference adds in place too).  relu as ATen's: [v <= 0] ? 0 : v (a NaN passes).
 * Backward: g = [y <= 0] ? 0 : grad_y;  grad_rem = g;  grad_t = bn_scale[c] * g (bn_scale NULL: = g; grad_t NULL with
 * bn_scale NULL: not written -- both inputs take grad_rem).
 * Replaces: batch_norm (eval) + add_ + relu_ and their backward nodes -- 7 volume passes forward in stock PyTorch, 3 here. */
int ganet_residual_relu_forward(const float *t, const float *rem, const float *bn_scale, const float *bn_shift,
                                float *y, int N, int C, int D, int H, int W, void *stream);
int ganet_residual_relu_backward(const float *y, const float *grad_y, const float *bn_scale, float *grad_t,
                                 float *grad_rem, int N, int C, int D, int H, int W, void *stream);

/* ---------------------------------------------------------------- diagnostics ---- */

/* Runs a 64-lane probe of every DPP pattern the kernels rely on and compares with
 * the documented lane maps.  scratch: device buffer of >= 8*64 ints.  host_out
 * (host memory, >= 8*64 ints) receives the raw lane values.  Synchronises `stream`.
 * Returns 0 if all patterns match. */
int ganet_selftest_dpp(int *scratch_dev, int *host_out, void *stream);
/* The same for the whole-wavefront patterns of the 64-lanes-per-scanline scans (wave_shl:1, wave_shr:1, the wave-wide
 * maximum and sum); scratch / host_out: >= 4*64 ints. */
int ganet_selftest_dpp_wave(int *scratch_dev, int *host_out, void *stream);

/* Knobs (also read from the environment at first use).  One default path per operator plus its general fallback; the knobs
 * exist so that tests can reach the fallbacks and the forced modes:
 *   GANET_SGA_ROWWAVE / GANET_SGA_COLBLOCK = 0|1  LDS-staged row-per-wave / column-block scans (default 1; 0: the 16-lane
 *                         segment kernels, which are also the fallback for W % 4 != 0 and D > 208)
 *   GANET_SGA_WIDE_SCAN=0|1|2  scans with the whole wavefront on one scanline: never | for inputs with few scanlines and for
 *                         D > 272 (default) | whenever D > 48 (tests).  D up to 1088
 *   GANET_SGA_WIDE_COL=0|1|2  vertical scans on LDS-staged column blocks with one wavefront per column (1,024-thread blocks,
 *                         D <= 192): never | for inputs with few column blocks and D >= 96 (default; measured on
 *                         [1,1,192,240,624]: forward 0.33 -> 0.21 ms, adjoint 0.52 -> 0.40) | whenever the kernel applies (tests)
 *   GANET_SGA_TILED = 0|1 ganet_sga_backward's private workspace: the vertical directions' adjoint volumes G_down / G_up in the API
 *                         layout | tiled [slice][W/16][H/4][D][4][16] where W % 16 == 0, H % 4 == 0 and the 16-column kernels run
 *                         (default; whole step -1.1 %; see ganet_sga_workspace_layout)
 *   GANET_LGA_WAVE=0|1|2  LGA kernel family (radius <= 2): 256-thread tiles (any radius; the general fallback) | wave-autonomous
 *                         plane-pair kernels, LDS-DMA staging, FMAs packed along plane pairs, one ring per wave on 32 x 2 pixel
 *                         tiles | the same with the forward / data-backward kernels on ONE ring per 256-thread workgroup (32 x 8
 *                         tiles: the halo'd tile staged once for four waves, 12 rows fetched for 8 instead of 24; a barrier per
 *                         plane pair).  Default 2 (whole step -1.4 ... -3.5 % in five A/Bs on four boxes against 1, profiles/r8*_ab_step*).
 *                         The filter gradient: under 2, where its x is pair-interleaved (an LGA2's second pass), the 75 taps of a 32 x 2
 *                         tile are split over a wave pair that shares the x and gy rings (five waves per SIMD instead of three: its launch
 *                         -7 %, profiles/r9k_*, r9l_*); with an API-layout x one ring per wave in both (the wave-pair form lost there,
 *                         the four-wave ring form of round 5 was no gain: removed)
 *   GANET_LGA_SEGS = n    depth segments per tile for the plane-pair forward / data-backward (0 = automatic)
 *   GANET_LGA_MIX=0|1|n   the same kernels with a MIXED item list: whole tiles first (a whole number per SIMD), the remaining
 *                         tiles cut into depth segments, at most one segment per SIMD -- the waves of a SIMD share one VALU, so
 *                         a pass lasts as long as the SIMD with the most tiles (3 of 2.34 on average at 240x624).  Default 1
 *                         (measured: forward pass 0.103 -> 0.0955 ms); n > 1: n SIMDs assumed (tests)
 * Read by ganet_amd.functions.GANet, not by this library: GANET_LGA_PAIRED=0 keeps the intermediate volume of a two-pass
 * chain in the API layout instead of pair-interleaved (ganet_lga_apply_paired; default on, measured -7 % on Lga2Function
 * fwd+bwd); GANET_SGA_SAVE=recompute selects the reference's memory profile.  GANET_TRACE_DISPATCH=1 prints which LGA kernel
 * a call took.  Unknown names -> GANET_E_INVALID. */
int ganet_set_option(const char *name, int value);
/* current value of a library option (>= 0), GANET_E_INVALID for an unknown name */
int ganet_get_option(const char *name);

#ifdef __cplusplus
}
#endif
#endif /* GANET_HIP_H */
