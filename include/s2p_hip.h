/* libs2p_hip.so -- C ABI of the MI355X (gfx950) S2P hot path.
 *
 * Drop-in boundary.  The reference checkout ships NO native code
 * and NO generator source, so "the reference interface each entry
 * replaces" is the PyTorch op the SPADE-lineage generator/discriminator would call from
 * Python (README.md:72-75 names the lineage; README.md:33,59 name the CLI that reaches
 * them).  Each entry point below cites that torch op; INTEGRATION.md shows the ctypes
 * stub a maintainer adds.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller
 *     (the library never allocates, frees or retains device memory);
 *   - every launch goes to the `stream` argument (a hipStream_t passed as void*), no
 *     implicit synchronisation, safe under hipGraph capture;
 *   - return 0 on success, negative on error; s2p_last_error() gives the message
 *     (thread-local); no C++ exception crosses the ABI;
 *   - activations are NHWC ("channels-last") in `dtype` (S2P_F32 or S2P_BF16); weights
 *     are "packed" [group][Cout][tap][Cin_pad] in the same dtype (s2p_pack_weights);
 *     master weights / gradients / optimizer state are fp32.
 */
#ifndef S2P_HIP_H
#define S2P_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#pragma GCC visibility push(default)

#define S2P_VERSION 124

enum { S2P_F32 = 0, S2P_BF16 = 1 };
enum { S2P_ACT_NONE = 0, S2P_ACT_RELU = 1, S2P_ACT_LRELU = 2, S2P_ACT_TANH = 3, S2P_ACT_SWISH = 4 };
/* epilogue modes of the conv family */
enum { S2P_EPI_STORE = 0,       /* y = act(conv + bias)                                   */
       S2P_EPI_ADD = 1,         /* y = act(conv + bias) + aux          (residual add)     */
       S2P_EPI_MUL_ACTGRAD = 2  /* y = conv * act'(aux)  (fused backward of the producer's
                                   ReLU / LeakyReLU; aux = that producer's OUTPUT)        */ };

/* One 2-D convolution problem, forward orientation.  torch equivalents:
 *   transposed == 0 : F.conv2d(x, w, b, stride, padding)            (pad mode zeros/reflect)
 *   transposed == 1 : F.conv_transpose2d(x, w, None, stride, padding, output_padding)
 * x : [N, H, W, x_pitch>=groups*Cin]   y : [N, Ho, Wo, y_pitch>=groups*Cout]
 * Cin must be a multiple of 16/sizeof(dtype) elements (pad thin inputs with zeros). */
typedef struct {
  int32_t dtype;
  int32_t N, H, W, Cin, x_pitch;
  int32_t Ho, Wo, Cout, y_pitch;
  int32_t KH, KW, stride, pad;
  int32_t transposed;
  int32_t reflect;
  int32_t groups;       /* >=1: batched independent convs; group g reads channels
                           [g*x_gstride, +Cin) and writes [g*y_gstride, +Cout).  A stride
                           >= the pitch makes that side group-major: [groups][N][H][W][pitch]
                           (whole tensors `stride` elements apart), e.g. the 12 gamma|beta
                           planes of the generator, one 1-KB-row tensor per norm            */
  int32_t x_gstride, y_gstride;
  int32_t cin_real;     /* un-padded input channels (0: unknown = Cin).  Lets the thin-input kernels skip zero
                           padding channels: a 7x7 conv with <= 4 real input channels (the generator's stem)
                           contracts 4 channels per tap instead of 8                                        */
} s2p_conv_desc;

int s2p_version(void);
const char* s2p_last_error(void);

/* ---- conv family (replaces torch.nn.functional.conv2d / conv_transpose2d and their
 *      autograd backward: cudnn_convolution_backward_input / _weight) ---------------- */
/* w_fwd : packed [groups][Cout][KH*KW][Cin]      (for transposed==1 too)               */
/* act == S2P_ACT_LRELU: slope must be <= 1 in every FORWARD entry point of this header (conv, fused conv + norm,
 * InstanceNorm forward / apply): the kernels evaluate  max(v, v * slope)  -- one multiply and one maximum; a larger slope is
 * rejected with an error (s2p_last_error).  The reference uses 0.2 everywhere (LeakyReLU(0.2): SPEC.md).                    */
int s2p_conv2d_fwd(const s2p_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                   const void* aux, void* y, int act, float slope, int epi, void* stream);
/* dx = d(loss)/dx.  w_bwd : packed [groups][Cin][KH*KW][Cout_pad] (the transpose of w_fwd).
 * dy has pitch y_pitch and channel count Cout (must itself satisfy the chunk multiple).
 * epi / aux as above (aux has the layout of dx); with S2P_EPI_MUL_ACTGRAD the result is
 * multiplied by aux_act'(aux) (aux = OUTPUT of the activation that produced x); an optional
 * aux2 (layout of dx) is added first: dx = (dgrad + aux2) * aux_act'(aux)  -- the gradient
 * arriving at x from a second consumer (e.g. a feature-matching / perceptual loss tap).
 * Reflect-padded convs: dx is produced on the PADDED grid [N,H+2p,W+2p,x_pitch]; fold it
 * with s2p_reflect_pad_bwd.                                                             */
int s2p_conv2d_dgrad(const s2p_conv_desc* d, const void* dy, const void* w_bwd,
                     const void* aux, const void* aux2, void* dx, int epi, int aux_act, float slope,
                     void* stream);
/* s2p_conv2d_fwd / s2p_conv2d_dgrad with a caller-owned device scratch of at least s2p_conv2d_{fwd,dgrad}_workspace(...)
 * bytes (0 for most shapes).  bf16 launches that cannot fill the chip (<= 160 workgroups: small maps with a long K, e.g.
 * the PatchGAN 256->512 layers on 7x7 / 12x12 maps) then split K over the idle CUs: every slice stores an fp32 partial
 * tile to the scratch and a second kernel applies bias / activation / epilogue to their fixed-order sum (no atomics;
 * bitwise reproducible).  A NULL / short workspace runs the unsplit kernels: same result up to fp32 summation order.   */
size_t s2p_conv2d_fwd_workspace(const s2p_conv_desc* d, int epi);
int s2p_conv2d_fwd_ws(const s2p_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                      const void* aux, void* y, int act, float slope, int epi, void* workspace,
                      size_t workspace_bytes, void* stream);
size_t s2p_conv2d_dgrad_workspace(const s2p_conv_desc* d);
int s2p_conv2d_dgrad_ws(const s2p_conv_desc* d, const void* dy, const void* w_bwd,
                        const void* aux, const void* aux2, void* dx, int epi, int aux_act, float slope,
                        void* workspace, size_t workspace_bytes, void* stream);
/* conv -> InstanceNorm -> MAT / SPADE modulation -> activation (replaces F.conv2d followed by F.instance_norm and the
 * `normalized * (1 + gamma) + beta` + LeakyReLU of the SPADE-lineage ResBlk: s2p_conv2d_fwd_ws + s2p_in_norm_fwd):
 *   y      = conv(x, w) + bias  [+ aux with epi == S2P_EPI_ADD]                 (kept: the backward needs it)
 *   y_mat  = act(IN(y) * (1 + g_img + g_st) + b_img + b_st),  stats = the statistics of y (s2p_in_stats format)
 * For bf16 3x3 stride-1 pad-1 convs on planes of 321..448 pixels (<= 21 x 21) with Cin, Cout multiples of 64 this is ONE
 * launch: the workgroup that owns an (image, 64-channel) plane of y normalises it in its epilogue (csrc/conv_plane.hip);
 * so is it, without gamma / beta maps (plain InstanceNorm: gb_img == NULL), for the PatchGAN 4x4 stride-1 layers on planes
 * of up to 192 pixels (csrc/conv_planeg.hip).  Other shapes run the two calls above back to back
 * (s2p_conv2d_mat_is_fused tells which).  groups must be 1; act: none / relu / lrelu.
 * y may be NULL where the launch is fused (s2p_conv2d_mat_is_fused): the conv output itself is then not written -- a forward
 * pass that keeps nothing for a backward only needs y_mat.                                                               */
int s2p_conv2d_fwd_mat(const s2p_conv_desc* d, const void* x, const void* w_fwd, const float* bias, const void* aux,
                       void* y, int epi, const void* gb_img, int gb_pitch, const float* gb_st, int gb_st_pitch,
                       int act, float slope, float eps, void* y_mat, int y_mat_pitch, float* stats, void* workspace,
                       size_t workspace_bytes, void* stream);
/* The backward counterpart: dgrad of a conv whose INPUT was the output of a MAT norm, followed by that norm's backward
 * (s2p_conv2d_dgrad_ws + s2p_in_norm_bwd_res).  d describes the conv (forward orientation); dy = dL/d(conv output);
 * xn / stats / gb_img / gb_st / act: the norm's input, statistics and modulation as in s2p_in_norm_bwd; outputs
 * dxn = dL/d(xn) (+ res), dgb_img, dgb_st as there.  aux (may be NULL; layout of d_mid) is a second gradient arriving at
 * the norm's OUTPUT (a feature-matching tap on the activation): it is added to the dgrad result before the norm backward.
 * Under the conditions of s2p_conv2d_fwd_mat (on the dgrad: Cout is the contraction, Cin the produced channels) this is ONE
 * launch and dL/d(norm output) never reaches HBM (d_mid and sums may then be NULL: s2p_conv2d_mat_is_fused); otherwise it
 * is written to d_mid ([N,H,W,x_pitch]) and the norm backward runs as its own launch(es) (needs `sums`:
 * s2p_in_bwd_sums_floats(N, H*W, Cin) floats).                                                                        */
int s2p_conv2d_dgrad_mat(const s2p_conv_desc* d, const void* dy, const void* w_bwd, void* d_mid, const void* aux, const void* xn,
                         int xn_pitch, const float* stats, const void* gb_img, int gb_pitch, const float* gb_st,
                         int gb_st_pitch, int act, float slope, float eps, float* sums, void* dxn, int dxn_pitch,
                         void* dgb_img, int dgb_pitch, float* dgb_st, int dgb_st_pitch, const void* res, int res_pitch,
                         void* workspace, size_t workspace_bytes, void* stream);
/* 1 when s2p_conv2d_fwd_mat (dgrad == 0) / s2p_conv2d_dgrad_mat (dgrad == 1) run this problem as ONE fused launch (no d_mid / sums
 * scratch needed), 0 when they fall back to two calls.  has_gb: gamma / beta image maps are passed.                    */
int s2p_conv2d_mat_is_fused(const s2p_conv_desc* d, int dgrad, int has_gb);
/* dw (fp32) [groups][Cout][KH*KW][Cin_real] for transposed==0,
 *           [groups][Cin][KH*KW][Cout_real] for transposed==1  (= channels-last physical
 * layout of the torch parameter).  dw is ACCUMULATED into (caller zeroes it);
 * dw_gstride = elements between groups.  cin_real/cout_real: un-padded channel counts.
 * db (may be NULL; transposed==0 only): fp32 [groups][Cout] bias gradient, ACCUMULATED into,
 * fused into the same pass over dy.                                                     */
int s2p_conv2d_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db,
                     int cin_real, int cout_real, int64_t dw_gstride, int splitk, void* stream);
/* The same with a caller-owned device scratch of at least s2p_conv2d_wgrad_workspace(...) bytes: the K-split units of the
 * bf16 kernels (and the workgroups of the thin tiled kernel, and the pixel blocks of a separate bias-gradient pass) then
 * store partial tiles / sums there and a second kernel adds them to dw / db in a fixed order -- no atomics, bitwise
 * reproducible (s2p_conv2d_wgrad itself, and a NULL / short workspace, accumulate with fp32 atomics).                    */
size_t s2p_conv2d_wgrad_workspace(const s2p_conv_desc* d, int cin_real, int cout_real);
int s2p_conv2d_wgrad_ws(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db,
                        int cin_real, int cout_real, int64_t dw_gstride, int splitk, void* workspace,
                        size_t workspace_bytes, void* stream);
/* Batched weight gradient: n_jobs (<= 16) convolutions of the SAME geometry `d` (groups must be 1; a grouped conv is
 * passed as one job per group with offset pointers and the tensors' pitches in d->x_pitch / d->y_pitch) in ONE launch.
 * `jobs` is a HOST array (copied into the kernel arguments: safe under hipGraph capture).  dw / db are ACCUMULATED into,
 * as in s2p_conv2d_wgrad.  For bf16 3x3 stride-1 pad-1 convs with Cin, Cout multiples of 64 this runs the slab kernel
 * (csrc/wgrad_slab.hip): no atomics -- K-split partial tiles go to `workspace` and are summed in a fixed order, so the
 * result is bitwise reproducible; other geometries run one s2p_conv2d_wgrad_ws launch per job on the same workspace.
 * workspace: caller-owned device scratch of at least s2p_conv2d_wgrad_batched_workspace(...) bytes (may be 0).       */
typedef struct { const void* x; const void* dy; float* dw; float* db; } s2p_wgrad_job;
size_t s2p_conv2d_wgrad_batched_workspace(const s2p_conv_desc* d, int n_jobs, int cin_real, int cout_real);
int s2p_conv2d_wgrad_batched(const s2p_conv_desc* d, const s2p_wgrad_job* jobs, int n_jobs, int cin_real,
                             int cout_real, void* workspace, size_t workspace_bytes, void* stream);
/* adjoint of F.pad(mode='reflect'): dx[N,H,W,C] = fold(dxp[N,H+2p,W+2p,C])              */
int s2p_reflect_pad_bwd(int dtype, const void* dxp, int N, int H, int W, int C, int pad, void* dx,
                        void* stream);
/* db[c] += sum over pixels of dy[p][c]  (bias gradient; fp32 accumulate into db)        */
int s2p_channel_sum(int dtype, const void* dy, int64_t pixels, int C, int pitch, float* db,
                    void* stream);

/* ---- instance norm + MAT/SPADE modulation (replaces F.instance_norm + the elementwise
 *      `normalized * (1 + gamma) + beta` + activation of the SPADE-lineage norm) -------- */
/* Per-(n,c) statistics of x over the HW pixels of each image.  `stats` is an OPAQUE fp32 buffer of
 * s2p_in_stats_floats(N,HW,C) elements, written (not accumulated: no zero-init needed) by s2p_in_stats and read by
 * the three consumers below with the SAME (N,HW,C): per-split partial moments {mean_b, M2_b} about a pivot, merged
 * by the consumers in a fixed order (no atomics: bitwise reproducible; no cancellation for |mean| >> std).
 * The buffer is self-describing (it starts with its split geometry): a consumer may be called on a batch PREFIX
 * (N' <= N images of the same x) with the same buffer.  Consumers use mean and rstd = 1/sqrt(biased var + eps). */
int64_t s2p_in_stats_floats(int N, int HW, int C);
int s2p_in_stats(int dtype, const void* x, int N, int HW, int C, int pitch, float eps,
                 float* stats, void* stream);
/* y = act(xhat*(1+g_img+g_st) + (b_img+b_st)).  gb_img: [N,HW,gb_pitch] with gamma at
 * channel offset 0 and beta at offset C (NULL -> 0); gb_st: fp32 [N][gb_st_pitch], gamma at
 * [0,C) beta at [C,2C) (NULL -> 0).                                                     */
int s2p_in_apply_fwd(int dtype, const void* x, int N, int HW, int C, int pitch,
                     const float* stats, const void* gb_img, int gb_pitch,
                     const float* gb_st, int gb_st_pitch, int act, float slope, float eps,
                     void* y, int y_pitch, void* stream);
/* s2p_in_stats + s2p_in_apply_fwd in one call: for planes of at most 512 (bf16) / 256 (fp32) pixels ONE fused launch
 * reads x once, computes exact two-pass statistics and applies the modulation; larger planes run the two kernels.
 * `stats` (s2p_in_stats_floats(N,HW,C) floats) is written in the same format either way, for the backward.        */
int s2p_in_norm_fwd(int dtype, const void* x, int N, int HW, int C, int pitch, const void* gb_img, int gb_pitch,
                    const float* gb_st, int gb_st_pitch, int act, float slope, float eps, void* y, int y_pitch,
                    float* stats, void* stream);
/* backward, given da = dL/dy (post-activation).  `sums` is an OPAQUE fp32 buffer of
 * s2p_in_bwd_sums_floats(N,HW,C) elements (per-split partial backward sums; written by
 * s2p_in_bwd_reduce, read by s2p_in_bwd_apply; no zero-init needed).
 * s2p_in_bwd_apply writes dx, d(gamma_img | beta_img) into dgb_img (layout of gb_img; may be NULL)
 * and d(gamma_st | beta_st) into dgb_st (fp32 [N][dgb_st_pitch], layout of gb_st; may be NULL). */
int64_t s2p_in_bwd_sums_floats(int N, int HW, int C);
int s2p_in_bwd_reduce(int dtype, const void* da, int da_pitch, const void* x, int N, int HW, int C,
                      int pitch, const float* stats, const void* gb_img, int gb_pitch,
                      const float* gb_st, int gb_st_pitch, int act, float slope, float eps,
                      float* sums, void* stream);
int s2p_in_bwd_apply(int dtype, const void* da, int da_pitch, const void* x, int N, int HW, int C,
                     int pitch, const float* stats, const void* gb_img, int gb_pitch,
                     const float* gb_st, int gb_st_pitch, int act, float slope, float eps,
                     const float* sums, void* dx, int dx_pitch, void* dgb_img, int dgb_pitch,
                     float* dgb_st, int dgb_st_pitch, void* stream);

/* ---- state path: positional encoding (nerf-pytorch embedder)  ------------------------ */
/* out[n][0:S]=s, then for k<L: sin(2^k s), cos(2^k s); columns >= S*(1+2L) up to out_pitch
 * are zero-filled.  fp32 in / fp32 out.                                                 */
int s2p_posenc_fwd(const float* state, int N, int S, int L, float* out, int out_pitch, void* stream);

/* s2p_in_bwd_reduce + s2p_in_bwd_apply in one call (one fused launch for planes of at most 512 (bf16) / 256 (fp32) pixels with
 * relu / lrelu / no activation; otherwise the two kernels, which need `sums`).                                             */
int s2p_in_norm_bwd(int dtype, const void* da, int da_pitch, const void* x, int N, int HW, int C, int pitch,
                    const float* stats, const void* gb_img, int gb_pitch, const float* gb_st, int gb_st_pitch,
                    int act, float slope, float eps, float* sums, void* dx, int dx_pitch, void* dgb_img,
                    int dgb_pitch, float* dgb_st, int dgb_st_pitch, void* stream);
/* The same with dx += res folded into the store (res: a tensor of dx's layout and dtype, e.g. the skip-connection gradient of
 * a residual block; fp32 add, one rounding; the two-kernel path adds it in a third pass).                                   */
int s2p_in_norm_bwd_res(int dtype, const void* da, int da_pitch, const void* x, int N, int HW, int C, int pitch,
                        const float* stats, const void* gb_img, int gb_pitch, const float* gb_st, int gb_st_pitch,
                        int act, float slope, float eps, float* sums, void* dx, int dx_pitch, void* dgb_img,
                        int dgb_pitch, float* dgb_st, int dgb_st_pitch, const void* res, int res_pitch, void* stream);

/* Small fp32 linear layers of the state path (replaces F.linear + LeakyReLU and their autograd backward for the
 * StateMapping MLP and the per-norm state affine; batch M of a few dozen rows: latency-bound, csrc/linear_small.hip).
 * y[M][y_pitch] = act(x[M][K] . w[N][w_row]^T + bias[N]); columns [N, n_store) of y are written as zeros.
 * K, pitches and w_row must be multiples of 4 floats.                                    */
int s2p_linear_fwd(const float* x, int M, int K, int x_pitch, const float* w, int w_row, const float* bias, int N,
                   int act, float slope, float* y, int y_pitch, int n_store, void* stream);
/* Backward given dy = dL/dy and the layer OUTPUT y (needed when act != NONE; dpre = dy * act'(y)):
 *   dw[N][dw_row] += dpre^T . x (columns < k_real), db[N] += sum_m dpre (db may be NULL),
 *   dx[M][dx_pitch] = dpre . w  (dx may be NULL; needs w_bwd [K][wb_row], the transpose of w).
 * No atomics: fixed summation order.  workspace: s2p_linear_bwd_workspace(M,K,N) bytes (0 for N < 2048). */
size_t s2p_linear_bwd_workspace(int M, int K, int N);
int s2p_linear_bwd(const float* x, int x_pitch, const float* dy, int dy_pitch, const float* y, int y_pitch, int M,
                   int K, int k_real, int N, const float* w_bwd, int wb_row, int act, float slope, float* dw,
                   int dw_row, float* db, float* dx, int dx_pitch, void* workspace, size_t workspace_bytes,
                   void* stream);

/* ---- pooling / resize / layout ---------------------------------------------------- */
/* F.avg_pool2d(k=3,s=2,p=1,count_include_pad=False) and its backward                   */
int s2p_avgpool3x3s2_fwd(int dtype, const void* x, int N, int H, int W, int C, void* y, void* stream);
int s2p_avgpool3x3s2_bwd(int dtype, const void* dy, int N, int H, int W, int C, void* dx,
                         int accumulate, void* stream);
/* F.max_pool2d(2,2) (floor) and backward fused with the producer's ReLU mask            */
int s2p_maxpool2x2_fwd(int dtype, const void* x, int N, int H, int W, int C, void* y, void* stream);
int s2p_maxpool2x2_bwd(int dtype, const void* dy, const void* x, int N, int H, int W, int C,
                       void* dx, void* stream);
/* F.interpolate(mode='nearest') on NHWC                                                 */
int s2p_resize_nearest(int dtype, const void* x, int N, int H, int W, int C, void* y, int Ho, int Wo,
                       void* stream);
/* fp32 NCHW [N,C,H,W]  ->  NHWC dtype with channel pitch (zero pad), written at channel
 * offset c_off; and back.                                                               */
int s2p_nchw_to_nhwc(int dtype, const float* x, int N, int C, int H, int W, void* y, int y_pitch,
                     int c_off, int zero_pad, void* stream);
int s2p_nhwc_to_nchw(int dtype, const void* x, int x_pitch, int c_off, int N, int C, int H, int W,
                     float* y, int accumulate, void* stream);
/* dataset frames: uint8 NHWC [pixels][C] (the layout rlkit/torch/slac/algo.py:189-190 reads) ->
 * NHWC dtype in [-1,1] with zero-padded pitch (v = u8/127.5 - 1), and back
 * (u8 = clamp(round((v+1)*127.5))); the round trip is exact on all 256 values.          */
int s2p_u8_to_nhwc(int dtype, const void* x, int64_t pixels, int C, void* y, int y_pitch, void* stream);
int s2p_nhwc_to_u8(int dtype, const void* x, int x_pitch, int64_t pixels, int C, void* y, void* stream);
/* generic cast copy between dtypes (n elements)                                         */
int s2p_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream);

/* ---- losses (forward value + gradient seed in one pass) ---------------------------- */
/* loss_out[0] += scale * sum|a-b| ; if grad_a: grad_a = (accumulate? grad_a:0) + scale*sign(a-b)
 * a,b: `count` elements each (dtype)                                                    */
int s2p_l1_loss(int dtype, const void* a, const void* b, int64_t count, float scale,
                float* loss_out, void* grad_a, int accumulate, void* stream);
/* n_jobs (<= S2P_L1_MAX_JOBS) such terms in one launch (the feature-matching maps of all PatchGAN scales, the VGG taps);
 * `jobs` is a HOST array (copied into the kernel arguments).  Pointers 16-byte aligned, counts a multiple of the 16-byte
 * chunk (8 bf16 / 4 fp32); grad_a is overwritten (no accumulate form).                                              */
#define S2P_L1_MAX_JOBS 16
typedef struct { const void* a; const void* b; void* grad_a; int64_t count; float scale; float* loss_out; } s2p_l1_job;
int s2p_l1_loss_multi(int dtype, const s2p_l1_job* jobs, int n_jobs, void* stream);
/* hinge terms on a D logit map x (count elements):
 *   mode 0: loss += scale*sum(relu(1+x)), grad = scale*(1+x>0)      (D on fake)
 *   mode 1: loss += scale*sum(relu(1-x)), grad = -scale*(1-x>0)     (D on real)
 *   mode 2: loss += -scale*sum(x),       grad = -scale              (G)                */
int s2p_hinge_loss(int dtype, const void* x, int64_t count, int mode, float scale,
                   float* loss_out, void* grad_x, void* stream);
/* the same on an NHWC map x [pixels][pitch] whose channel 0 is the logit (the discriminator heads' output layout);
 * grad_x (same layout, may be NULL): gradient in channel 0, zeros in the other channels                              */
int s2p_hinge_loss_strided(int dtype, const void* x, int64_t pixels, int pitch, int mode, float scale,
                           float* loss_out, void* grad_x, void* stream);

/* ---- ensemble state-dynamics head (reference gaussian_ensemble.py:83-96 and
 *      state_transition_rollout.py:192-204) ------------------------------------------------
 * raw   : fp32 [B][E*2*D] output of the last ensemble layer (member e at columns e*2D: mu | logstd)
 * xin   : fp32 [B][x_pitch] normalised (obs, action); obs_dim = D-1 leading columns
 * mean/std: fp32 [E][B][D] (may be NULL): mu (obs part + input obs, 'local' mode) and
 *         exp(soft_clamp(logstd, min_logstd, max_logstd))
 * pick  : int32 [B] member index per sample; next_obs [B][D-1] = mean[pick]*obs_std+obs_mean,
 *         reward [B] = mean[pick][D-1]*rew_std+rew_mean
 * disagreement[B] = max_e || mean_e[:D-1] - avg_e mean[:D-1] ||_2 ; aleatoric[B] = max_e || std_e ||_2 */
int s2p_ensemble_head(const float* raw, int raw_pitch, const float* xin, int x_pitch, int B, int E, int D,
                      const float* min_logstd, const float* max_logstd, float* mean, float* std,
                      const int32_t* pick, const float* obs_mean, const float* obs_std, float rew_mean,
                      float rew_std, float* next_obs, float* reward, float* disagreement, float* aleatoric,
                      void* stream);

/* ---- optimizer + weight packing ---------------------------------------------------- */
/* torch.optim.Adam step on flat fp32 buffers; g is multiplied by grad_scale first.      */
int s2p_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale, void* stream);
/* same update with the step counter in DEVICE memory (incremented on the device first), so the launch
 * can be captured in a hipGraph and replayed                                             */
int s2p_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                      float beta2, float eps, int* step_dev, float grad_scale, void* stream);
/* the same update of a RANGE of a flat buffer; tick != 0 increments the device step counter first.  One optimizer step
 * applied in several launches (the first with tick = 1, the others with tick = 0: they read the counter the first one
 * wrote, so order them behind it) -- e.g. the part of a gradient buffer that is final early, under the rest of the backward */
int s2p_adam_step_dev_part(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                           float beta2, float eps, int* step_dev, float grad_scale, int tick, void* stream);
/* one packing job: src fp32 [R][T][C] (channels-last master weight: R rows, T taps, C
 * channels) -> dst_fwd[r][t][c] (row length T*Cpad, zero pad c>=C)  and/or
 * dst_bwd[c][t][r_off + r] (row length T*Rrow; untouched elements must be pre-zeroed)    */
typedef struct {
  const float* src; void* dst_fwd; void* dst_bwd;
  int32_t R, T, C;
  int32_t Cpad;        /* fwd: padded channel count                                     */
  int32_t Rrow;        /* bwd: row length (in r) of the transposed matrix               */
  int32_t r_off;       /* bwd: column offset of this job inside a fused matrix          */
  int32_t dtype;
} s2p_pack_job;
/* jobs: DEVICE array of n_jobs descriptors (caller-owned)                               */
int s2p_pack_weights(const s2p_pack_job* jobs, int n_jobs, int max_elems, void* stream);

/* ---- small elementwise helpers ------------------------------------------------------ */
/* dx = dy * act'(y)   (y = activation OUTPUT)                                           */
int s2p_act_bwd(int dtype, const void* dy, const void* y, int64_t n, int act, float slope, void* dx,
                void* stream);
/* x *= *scale  (device fp32 scalar: applies an upstream grad_output without a host sync) */
int s2p_scale(int dtype, void* x, int64_t n, const float* scale, void* stream);
/* out = a + b (n elements; out may alias a or b)                                         */
int s2p_add(int dtype, const void* a, const void* b, void* out, int64_t n, void* stream);
/* dst[p][dst_off+c] (+)= src[p][src_off+c] for c<C, p<pixels  (torch.cat(dim=1) on NHWC and
 * its backward slice)                                                                   */
int s2p_copy_channels(int dtype, const void* src, int src_pitch, int src_off, void* dst,
                      int dst_pitch, int dst_off, int C, int64_t pixels, int accumulate, void* stream);

/* ---- image-fidelity metrics (the paper's PSNR / SSIM, rebuttal.md:50 -- no reference code) ----
 * a, b: fp32 NCHW [N,C,H,W].  sq_err_sum[n] += sum over the image of (a-b)^2;  ssim_sum[n] += sum over channels and
 * over the (H-10)x(W-10) fully covered positions of the 11x11 Gaussian-window (sigma 1.5) SSIM index with
 * C1 = (0.01 R)^2, C2 = (0.03 R)^2, R = data_range.  Both accumulators are caller-zeroed fp32 [N].
 * PSNR = 10 log10(R^2 C H W / sq_err_sum);  SSIM = ssim_sum / (C (H-10) (W-10)).  Requires H, W >= 11.            */
int s2p_image_metrics(const float* a, const float* b, int N, int C, int H, int W, float data_range,
                      float* sq_err_sum, float* ssim_sum, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
