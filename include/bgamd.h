/*
 * bgamd.h -- C ABI of libbgamd.so: the MI355X (gfx950) kernels behind the
 * Bias-GAN conv-GAN training step.
 *
 * The reference has no C plugin ABI for this path: its arithmetic is torch.nn
 * modules (src/deepCam/architecture/gpsro/deeplab.py, deeplab_gan.py,
 * utils/losses.py) and its own pattern for a native op is a
 * torch.autograd.Function around an extension's forward()/backward()
 * (deeplab.py:9-22, Conv2dLocalFunction).  Each entry point below replaces the
 * torch.nn op sequence named in its comment; the Python host mirror
 * (bias-gan_amd/ops.py) binds them with ctypes from autograd.Function wrappers
 * of exactly that shape.  INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All pointers are DEVICE
 *     pointers unless a comment says host.  The caller owns every buffer; the
 *     library keeps no pointer after a call returns and has no global state.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as
 *     void*); it never synchronises, allocates or frees.
 *   - return value: 0 on success, a negative BG_E_* code otherwise;
 *     bg_last_error() returns a thread-local message for the last failure.
 *   - activations are NHWC ("pixel-major"): element (n,h,w,c) of a tensor with
 *     pixel stride ld lives at ((n*H+h)*W+w)*ld + c.  ld >= C lets an op read
 *     or write a channel slice of a wider buffer (concat without a copy).
 *     C and ld must be multiples of 8 (bf16) / 4 (f32): 16-byte vectors.
 *   - dtype: BG_BF16 = activations/weights in bfloat16 with fp32 accumulation
 *     (the performance path), BG_F32 = everything fp32 (the parity path).
 *     Statistics, losses, gradients of parameters and optimiser state are
 *     always fp32 (statistic sums fp64).
 *   - conv weights: forward reads [Cout][KH][KW][Cin_k] ("KRSC") and bwd_data the
 *     transposed copy [Cin][KH][KW][Cout_k] ("CRSK"), where the innermost
 *     (reduction) dimension is zero-padded to a multiple of bg_conv_weight_kpad()
 *     elements (64 for bf16, 32 for f32) so the GEMM K loop needs no tail
 *     predicate; bg_pack_conv_weights builds both copies from the dense fp32 /
 *     bf16 master layout [Cout][KH][KW][Cin].  Depthwise weights are [KH][KW][C].
 */
#ifndef BGAMD_H
#define BGAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_BF16 0
#define BG_F32 1
#define BG_FP8 2 /* operand storage of the bg_*_fp8 entry points only (OCP e4m3 / e5m2, one byte per element) */
#define BG_FP8_E4M3 0
#define BG_FP8_E5M2 1

#define BG_OK 0
#define BG_E_ARG (-1)    /* bad argument / shape the kernels do not support */
#define BG_E_LAUNCH (-2) /* hipLaunch failed (message carries hipGetErrorString) */

#define BG_ABI_VERSION 4

int bg_abi_version(void);
const char* bg_last_error(void);
/* A dedicated non-blocking HIP stream on the current device (host-side call; *out receives the hipStream_t).  The host
 * mirror wraps these (torch.cuda.ExternalStream) for its side work instead of taking streams from PyTorch's shared pool
 * of 32 per device, in which two side streams may be the same HIP stream. */
int bg_stream_create(void** out);
int bg_stream_destroy(void* stream);

/* ---------------------------------------------------------------------------
 * Dense convolution as implicit GEMM on MFMA (nn.Conv2d with groups=1:
 * deeplab.py:178,182 entry convs; :58,81,95 pointwise/skip 1x1; :331 ASPP
 * dilated 3x3; :363-369 decoder; :622,626,630 1x1).
 * out = floor((in + 2*pad - dil*(k-1) - 1)/stride) + 1 must hold for (H,Ho),(W,Wo).
 * ------------------------------------------------------------------------- */
typedef struct bg_conv_desc {
    int32_t dtype;            /* BG_BF16 | BG_F32 */
    int32_t N, H, W, Cin;     /* input  [N,H,W,Cin],  pixel stride ldx */
    int32_t Ho, Wo, Cout;     /* output [N,Ho,Wo,Cout], pixel stride ldy */
    int32_t KH, KW;
    int32_t stride, pad, dil; /* symmetric, same for H and W */
    int32_t ldx, ldy;
} bg_conv_desc;

/* y = conv(x, w) (+ bias[Cout], fp32, may be NULL).  w: K-padded KRSC copy, dtype = d->dtype. */
int bg_conv2d_fwd(const bg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, void* stream);
/* Same as bg_conv2d_fwd without bias, and additionally sum[.][c] += sum_pixels y, sumsq[.][c] += sum_pixels y^2
 * of the outputs as stored (fp64, caller zeroes): the batch statistics of the BatchNorm that follows
 * the convolution, taken from the accumulators instead of re-reading y (replaces bg_norm_stats).
 * groups > 1: the output pixels form `groups` equal contiguous ranges (sub-batches that are normalised
 * separately, e.g. D(real) and D(fake) pushed through the network as one batch); sum/sumsq are
 * [groups][Cout].  A group's pixel count must then be a multiple of 128. */
int bg_conv2d_fwd_stats(const bg_conv_desc* d, const void* x, const void* w, void* y, double* sum, double* sumsq,
                        int32_t groups, void* stream);
/* dx = conv_transpose(dy, w).  wt: K-padded CRSK copy of the weights.  Overwrites dx. */
int bg_conv2d_bwd_data(const bg_conv_desc* d, const void* dy, const void* wt, void* dx, void* stream);
/* dw += x (*) dy, dw fp32 KRSC (accumulated with float atomics; caller zeroes it
 * once per backward pass).  dbias (fp32 [Cout], may be NULL) += sum over pixels of dy. */
/* Split-K forms for launches with few output tiles and a long reduction (3x3x3 convolutions on the 1x2x3 maps at the
 * bottom of the 3-D nets: one or two 128 x 128 tiles walking > 1000 K-steps): (tiles x splits) workgroups, each
 * stores its K range's partial tile into its own slice of ws, fp32 [splits][N*Ho*Wo][Cout] (resp.
 * [splits][N*H*W][Cin]); bg_splitk_reduce then sums the slices in index order -- no atomics, results independent of
 * scheduling -- and rounds to the activation dtype; for a BatchNorm that follows the caller takes bg_norm_stats of
 * the result.  `splits` must divide the K-steps (KH*KW*ceil(C/32) for bf16, /16 for fp32) into non-empty ranges
 * of ceil(K-steps / splits).  No bias. */
int bg_conv2d_fwd_splitk(const bg_conv_desc* d, const void* x, const void* w, float* ws, int32_t splits, void* stream);
int bg_conv2d_bwd_data_splitk(const bg_conv_desc* d, const void* dy, const void* wt, float* ws, int32_t splits, void* stream);
int bg_splitk_reduce(int32_t dtype, const float* ws, int32_t splits, int64_t rows, int32_t C, void* y, int32_t ldy, void* stream);
int bg_conv2d_bwd_weight(const bg_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, void* stream);
/* The same weight gradient WITHOUT float atomics: every pixel split stores its 128 x 128 partial tiles into `ws` (plain
 * stores) and a second launch adds them to dW in split order -- each dW element is written by one thread, so the result does
 * not depend on workgroup scheduling (bit-reproducible).  Measured against the atomics: equal on the MFMA-bound 3 x 3 layers,
 * 0.6-0.9x on HBM-bound 1 x 1 layers (the split tiles are written and read back), so the host mirror keeps it opt-in.
 * bg_conv2d_bwd_weight_ws_bytes: the workspace this descriptor needs (tiles x splits x 64 KiB; a few tens of MB). */
int bg_conv2d_bwd_weight_ws_bytes(const bg_conv_desc* d, int64_t* bytes);
int bg_conv2d_bwd_weight_ws(const bg_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, float* ws,
                            int64_t ws_bytes, void* stream);
/* Weight gradients of n_layers POINTWISE (1x1, stride 1) convolutions of one shape in one launch (bf16 operands):
 * dw_l[Cout][Cin] (fp32) += dy_l^T x_l over the M pixels, for l < n_layers.  tbl (HOST memory, read during the call):
 * n_layers rows of 4 int64 {x_l, dy_l, dw_l, 0} device addresses -- they travel in the kernel arguments.  The (layer, pixel) space is cut into equal ranges walked by gangs of one
 * workgroup per output tile; a dw tile receives at most two float-atomic adds when the group holds more layers
 * than ranges' worth of work (then the result is bit-reproducible), operands are read from HBM once.  No bias. */
int bg_conv2d_bwd_weight_grouped(int32_t dtype, const int64_t* tbl, int32_t n_layers, int64_t M, int32_t Cin, int32_t Cout,
                                 int32_t ldx, int32_t ldy, void* stream);
/* The same for n_layers k x k convolutions of ONE geometry d (bf16; stride 1, odd square kernel <= 5, pad = dil*(k-1)/2,
 * so Ho = H, Wo = W: the 3x3 convolutions of the decoder, deeplab.py:363-369, and of the ASPP, :331): a tap is a
 * pointwise weight gradient against x shifted by the tap's rows / columns with the border masked, so the nine taps
 * run as members of one gang (few tiles: dy and the shifted windows of x are fetched once per XCD) or as nine layers.
 * dw_l is [Cout][KH][KW][Cin] fp32, accumulated into; tbl as above. */
int bg_conv2d_bwd_weight_grouped_taps(const bg_conv_desc* d, const int64_t* tbl, int32_t n_layers, void* stream);

/* Reduction-dimension padding granule of the packed weight copies (elements). */
int bg_conv_weight_kpad(int32_t dtype);
/* Tuning / test hook for the forward and data-gradient GEMM launches of this process: -1 = tile family by the
 * library's heuristics (default), 0 = the 64 x 64-per-wave tiles only, 2 = the fat-tile kernel (one 384/256-row tile
 * per CU) wherever it is legal, small shapes included.  Results are the same up to summation order. */
int bg_conv_set_variant(int32_t variant);
/* Build the padded KRSC and CRSK copies for a batch of layers in one launch.  tbl
 * (device, n entries of 8 int64: src_off, krsc_off, crsk_off, K, RS, C, Cp, Kp in
 * elements; Cp/Kp = C/K rounded up to the granule) addresses each layer inside the
 * flat buffers; max_elems = largest K*RS*Cp + C*RS*Kp over the layers. */
int bg_pack_conv_weights(int32_t dtype, const void* src, void* dst_krsc, void* dst_crsk, const int64_t* tbl,
                         int32_t n_layers, int64_t max_elems, void* stream);

/* ---------------------------------------------------------------------------
 * fp8 operand path of the dense convolutions (BASELINE.json configs[4]: "fp8 MFMA conv path with bf16
 * accumulate"; no counterpart in the reference -- it replaces the same nn.Conv2d calls as bg_conv2d_fwd /
 * _bwd_data).  Operands are OCP fp8 bytes, q = round_to_nearest_even(clamp(v * 2^e, +-max)) with ONE power-of-two
 * exponent e per tensor kept in device memory; the MFMA is the block-scaled 16x16x128 f8f6f4 form (twice the bf16
 * rate, fp32 accumulation) whose hardware block scales carry 2^-e, so outputs are in real units.  Outputs, the
 * statistics epilogue, tiling and determinism are those of the bf16 fat-tile kernel (bf16 y / dx, fp64 sums).
 *
 * bg_quant_fp8: xq[r][c] = fp8(x[r][c] * 2^(*exp)) for c < C, zeros for C <= c < Cq (Cq = C rounded up to 16: the
 *   16-byte vector of one-byte elements); src_dtype BG_BF16 | BG_F32; fmt BG_FP8_E4M3 (max 448) | BG_FP8_E5M2
 *   (max 57344).  amax (may be NULL): *amax = max(*amax, max |x|) as the bits of a non-negative float (atomic max on
 *   uint32) -- the statistic bg_fp8_roll turns into the NEXT call's exponent (delayed scaling).
 * bg_fp8_roll: for i < n: exp[i] = floor(log2(max_of(fmt[i]) / amax[i])) - margin (0 where amax[i] == 0), amax[i] = 0.
 * bg_pack_conv_weights_fp8: bg_pack_conv_weights from the fp32 master layout into e4m3 KRSC / CRSK copies (reduction
 *   dimension zero-padded to 128), one exponent per layer derived from the layer's own max |w| in the same call
 *   (exps[n_layers], amax_ws[n_layers] scratch; tbl as bg_pack_conv_weights with Cp / Kp multiples of 128).
 * bg_conv2d_fwd_fp8: y (bf16) = conv(xq, wq) (+ bias); sum/sumsq as bg_conv2d_fwd_stats (NULL: none).  d->dtype must
 *   be BG_BF16 (the output type); d->Cin / d->ldx describe xq (bytes; multiples of 16).
 * bg_conv2d_bwd_data_fp8: dx (bf16) = conv_transpose(dyq, wtq); dy_fmt names dyq's format; d->Cout / d->ldy
 *   describe dyq, d->Cin / d->ldx the bf16 dx.
 * ------------------------------------------------------------------------- */
/* bg_norm_act_bwd_apply_stats that ALSO writes dx as e5m2 bytes for the data-gradient GEMM of the convolution that
 * produced x (bg_quant_fp8's contract with fmt = BG_FP8_E5M2 on the dx values as stored; dxq: [rows][lddxq] bytes, C
 * rounded up to 16 with zero lanes; *q_amax updated): one quantisation pass saved per layer.  bf16, training mode,
 * dx required (act may be 0: the apply pass behind a fused fork backward, whose gradient already carries act'). */
int bg_norm_act_bwd_apply_stats_q8(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                                   int32_t ldx, const double* s1, const double* s2, const float* gamma, const float* beta,
                                   const float* mean, const float* rstd, int32_t train, float* dgamma, float* dbeta, void* dx,
                                   int32_t lddx, void* dres, int32_t lddres, int64_t rows, int32_t C, int32_t groups,
                                   int32_t act, void* dxq, int32_t lddxq, const int32_t* q_exp, uint32_t* q_amax, void* stream);
int bg_quant_fp8(int32_t src_dtype, const void* x, int32_t ldx, int64_t rows, int32_t C, void* xq, int32_t ldq, int32_t Cq,
                 int32_t fmt, const int32_t* exp, uint32_t* amax, void* stream);
int bg_fp8_roll(int32_t* exp, uint32_t* amax, const int32_t* fmt, int32_t n, int32_t margin, void* stream);
int bg_pack_conv_weights_fp8(const float* src, void* dst_krsc, void* dst_crsk, const int64_t* tbl, int32_t n_layers,
                             int64_t max_elems, int32_t* exps, uint32_t* amax_ws, void* stream);
int bg_conv2d_fwd_fp8(const bg_conv_desc* d, const void* xq, const void* wq, const int32_t* exp_x, const int32_t* exp_w,
                      const float* bias, void* y, double* sum, double* sumsq, int32_t groups, void* stream);
int bg_conv2d_bwd_data_fp8(const bg_conv_desc* d, const void* dyq, int32_t dy_fmt, const void* wtq, const int32_t* exp_dy,
                           const int32_t* exp_w, void* dx, void* stream);

/* ---------------------------------------------------------------------------
 * Depthwise 3x3 of SeparableConv2d_same, with fixed_padding folded in
 * (deeplab.py:66-87: pad = dil on every side for k=3).  Ho = ceil(H/stride).
 * w: [3][3][C] dtype; dw: [3][3][C] fp32 accumulated.
 * ------------------------------------------------------------------------- */
typedef struct bg_dwconv_desc {
    int32_t dtype;
    int32_t N, H, W, C;
    int32_t Ho, Wo;
    int32_t stride, dil;
    int32_t ldx, ldy;
} bg_dwconv_desc;
int bg_dwconv3x3_fwd(const bg_dwconv_desc* d, const void* x, const void* w, void* y, void* stream);
int bg_dwconv3x3_bwd_data(const bg_dwconv_desc* d, const void* dy, const void* w, void* dx, void* stream);
int bg_dwconv3x3_bwd_weight(const bg_dwconv_desc* d, const void* x, const void* dy, float* dw, void* stream);
/* dx = bg_dwconv3x3_bwd_data(dy) + add in one pass: the gradient at a Block's input is the data gradient of its first
 * depthwise convolution plus the gradient arriving through the skip path (deeplab.py:134-141; autograd would form the
 * sum in a pass of its own).  add: [N, H, W, C] of pixel stride ldadd; dx may alias add.  Stride 1, dilation 1 or 2. */
int bg_dwconv3x3_bwd_data_add(const bg_dwconv_desc* d, const void* dy, const void* w, const void* add, int32_t ldadd,
                              void* dx, void* stream);
/* The same two kernels with the producer's normalisation + activation applied to every loaded input chunk:
 * the depthwise convolution (and its weight gradient) of act(x*scale[g,c] + shift[g,c]) where x is the RAW output
 * of the previous pointwise convolution -- the activated tensor of the reference's
 * [BatchNorm2d -> LeakyReLU -> SeparableConv2d_same] chain (deeplab.py:90-143) is never written to memory.
 * scale/shift: fp32 [groups, C] from bg_norm_finalize_affine (image n belongs to group n / (N/groups)); the "same"
 * zero padding applies to the activated tensor; values are rounded to the storage type exactly as
 * bg_norm_act_fwd would have stored them, so both pipelines give identical bits.  Stride 1, dilation 1 or 2. */
int bg_dwconv3x3_fwd_pre(const bg_dwconv_desc* d, const void* x, const float* scale, const float* shift,
                         int32_t groups, int32_t act, const void* w, void* y, void* stream);
int bg_dwconv3x3_bwd_weight_pre(const bg_dwconv_desc* d, const void* x, const float* scale, const float* shift,
                                int32_t groups, int32_t act, const void* dy, float* dw, void* stream);
/* bg_norm_finalize_affine + bg_dwconv3x3_fwd_pre in ONE launch: sum / sumsq are the fp64 [groups, C] statistics of x
 * (bg_conv2d_fwd_stats' epilogue or bg_norm_stats); every workgroup derives the affine of its image's group itself and
 * the first workgroup of each group writes mean / rstd / scale / shift (fp32 [groups, C], needed by the backward
 * kernels); running_* (may be NULL) get one momentum update per group, in group order.  Same bits as the two calls. */
int bg_dwconv3x3_fwd_pre_stats(const bg_dwconv_desc* d, const void* x, const double* sum, const double* sumsq,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* mean, float* rstd, float* scale, float* shift, int32_t groups,
                               int32_t act, const void* w, void* y, void* stream);

/* Backward of the fused unit above in ONE pass over its two inputs (bf16): from dy (gradient w.r.t. the depthwise
 * output, pixel stride d->ldy) and x (the raw convolution output the forward kernel normalised, pixel stride d->ldx):
 *   da = depthwise data gradient of dy (gradient w.r.t. the activated tensor; pixel stride ldda; what
 *        bg_dwconv3x3_bwd_data writes), dw += the depthwise weight gradient on the recomputed activation (what
 *        bg_dwconv3x3_bwd_weight_pre adds; NULL: skipped), s1 / s2 (fp64 [groups, C], caller zeroes) += the two
 *        statistics of the BatchNorm backward (what bg_norm_act_bwd_reduce adds with y = NULL: sum g, sum g * xhat with
 *        g = da * act'(x * scale + shift), xhat = (x - mean) * rstd, da as stored).
 * scale / shift / mean / rstd: fp32 [groups, C] of the forward pass.  Stride 1, dilation 1 or 2.  Replaces three launches
 * and six tensor passes by three passes; bg_norm_act_bwd_apply_stats follows as before. */
int bg_dwconv3x3_bwd_fused(const bg_dwconv_desc* d, const void* dy, const void* w, const void* x, const float* scale,
                           const float* shift, const float* mean, const float* rstd, int32_t groups, int32_t act, void* da,
                           int32_t ldda, float* dw, double* s1, double* s2, void* stream);

/* The fork at a Block's input, backward, in ONE pass (bf16; round 4).  Reference: the Block of
 * architecture/gpsro/deeplab.py:90-143 feeds its input `inp` to `rep` (first op: the depthwise convolution of
 * SeparableConv2d_same, :75-87) and to the skip path (:134-141); autograd then adds the two gradients and continues into
 * the producer of `inp`, the previous Block's [BatchNorm -> (+ residual) -> LeakyReLU].  This entry point does, from
 *   dy   gradient w.r.t. the depthwise output (pixel stride d->ldy),
 *   a0   the Block's activated input (pixel stride d->ldx),
 *   skip gradient w.r.t. the skip alias of a0 (pixel stride ldskip),
 *   z    the producer BatchNorm's input, mean / rstd its saved fp32 [groups, C] statistics (pixel stride ldz):
 *   t    = bf16(depthwise data gradient of dy + skip)          -- what bg_dwconv3x3_bwd_data_add stores
 *   gout = t * act'(a0)      (act 1: a0 > 0 ? 1 : 0.2, act 2: a0 > 0 ? 1 : 0, act 0: 1)   -- the gradient BEHIND the
 *                              producer's activation: its bg_norm_act_bwd_apply_stats then runs with act = 0 and its
 *                              residual gradient is gout itself
 *   dw  += depthwise weight gradient (a0 against dy; NULL: weights frozen)
 *   s1 / s2 (fp64 [groups, C], caller zeroes) += sum gout, sum gout * (z - mean) * rstd   -- what bg_norm_act_bwd_reduce
 *                              adds for the producer.
 * Replaces bg_dwconv3x3_bwd_data_add + bg_dwconv3x3_bwd_weight + bg_norm_act_bwd_reduce and the residual-gradient output
 * of the apply pass: 10 tensor passes -> 5.  Stride 1, dilation 1. */
int bg_dwconv3x3_bwd_fork(const bg_dwconv_desc* d, const void* dy, const void* w, const void* a0, const void* skip,
                          int32_t ldskip, const void* z, int32_t ldz, const float* mean, const float* rstd, int32_t groups,
                          int32_t act, void* gout, int32_t ldgout, float* dw, double* s1, double* s2, void* stream);

/* ---------------------------------------------------------------------------
 * 3-D DeepLab GAN path (architecture/gpsro/deeplab3d.py).  A volume [N,D,H,W,C] is the
 * NHWC tensor [N*D,H,W,C]; every entry point above applies to it as it stands.  The third dimension adds:
 *
 * bg_depth_unfold: y[n,od,p, kd*C + c] = x[n, od*stride - pad + kd*dil, p, c] (0 outside), p = pixel of the H*W
 *   plane.  nn.Conv3d(k, stride, padding, dilation) (deeplab3d.py:120,124,274,307-312) = this, then the 2-D
 *   convolution entry points over KD*C input channels with weights [K][KH][KW][KD][C].  KD = 1, stride 2, pad 0
 *   is the depth subsampling of the 1x1x1 stride-2 skip convolutions (deeplab3d.py:51).
 * bg_depth_fold: the adjoint (gradient w.r.t. x from the gradient w.r.t. y); overwrites dx.
 * ------------------------------------------------------------------------- */
int bg_depth_unfold(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t D, int32_t Do,
                    int32_t HW, int32_t C, int32_t KD, int32_t stride, int32_t pad, int32_t dil, void* stream);
int bg_depth_fold(int32_t dtype, const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t N, int32_t D, int32_t Do,
                  int32_t HW, int32_t C, int32_t KD, int32_t stride, int32_t pad, int32_t dil, void* stream);
/* Depthwise 3x3x3 of SeparableConv3d_same with fixed_padding folded in (deeplab3d.py:22-43: pad = dil on every
 * side for k = 3).  Do/Ho/Wo = ceil(./stride).  w: [3][3][3][C] dtype; dw: [3][3][3][C] fp32 accumulated. */
typedef struct bg_dwconv3d_desc {
    int32_t dtype;
    int32_t N, D, H, W, C;
    int32_t Do, Ho, Wo;
    int32_t stride, dil;
    int32_t ldx, ldy;
} bg_dwconv3d_desc;
int bg_dwconv3x3x3_fwd(const bg_dwconv3d_desc* d, const void* x, const void* w, void* y, void* stream);
int bg_dwconv3x3x3_bwd_data(const bg_dwconv3d_desc* d, const void* dy, const void* w, void* dx, void* stream);
int bg_dwconv3x3x3_bwd_weight(const bg_dwconv3d_desc* d, const void* x, const void* dy, float* dw, void* stream);
/* Linear interpolation along depth, align_corners=True: F.interpolate(mode='trilinear', align_corners=True)
 * (deeplab3d.py:315-320,545) = this followed by bg_resize_bilinear_fwd on the folded tensor.  x: [N][Di][HW][ldx],
 * y: [N][Do][HW][ldy]; C a multiple of 4.  _bwd is the adjoint (overwrites dx). */
int bg_depth_resize_fwd(int32_t in_dtype, int32_t out_dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N,
                        int32_t Di, int32_t Do, int32_t HW, int32_t C, void* stream);
int bg_depth_resize_bwd(int32_t dy_dtype, int32_t dx_dtype, const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t N,
                        int32_t Di, int32_t Do, int32_t HW, int32_t C, void* stream);
/* Depth half of nn.AvgPool3d(2, stride=1, padding=p) of the 3-D Deconv upsamplers (deeplab3d.py:349-386,436,527; the
 * in-plane half is bg_avgpool2x2): y[n,od] = 0.5 * (x[n,od+off] + x[n,od+off+1]), zeros outside (count_include_pad).
 * off = -p: forward (Do = Di + 2p - 1); off = p - 1 with x := dy, y := dx: the adjoint. */
int bg_depth_avg2(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t Di, int32_t Do, int32_t HW,
                  int32_t C, int32_t off, void* stream);

/* ---------------------------------------------------------------------------
 * Partial-convolution U-Net GAN (architecture/common/partialconv3d.py,
 * architecture/gpsro/infill3d.py, infill3d_gan.py).  Volumes and masks are folded NHWC tensors [N*D,H,W,C];
 * masks hold exact 0/1 values.
 *
 * bg_mask_window: the mask half of PartialConv3d(multi_channel=True) (partialconv3d.py:49-75): s = sum of the mask
 *   over all C input channels and the k^3 window ("conv with all-ones weights"; the same for every output
 *   channel), update_mask = clamp(s, 0, 1), ratio = C*k^3 / (s + eps) * update_mask; both fp32 [N*Do*Ho*Wo].
 *   C is the real channel count of `mask` (pad lanes up to the next vector multiple are ignored).  The window
 *   sum runs over up to three segments of input channels: `mask` (a per-channel tensor, may be NULL) and rows0 /
 *   rows1 (fp32 [N*D*H*W], may be NULL): per-pixel masks standing for c0 / c1 identical channels -- what every
 *   update_mask is (its Cout channels are equal) -- so past the first layer no Cout-channel mask tensor exists.
 *   planar != 0: PartialConv2d (architecture/common/partialconv2d.py:49-75) on N*D independent images -- the
 *   window is k x k, the depth axis is not windowed (Do == D), winsize = C*k^2.
 * bg_resize_nearest3d_rows: the nearest resize of such a per-pixel mask.
 * bg_mul_rows:   y = x * m (element-wise; input * mask_in, partialconv3d.py:77).
 * bg_scale_rows: y[r,c] = (x ? x[r,c] : 1) * s[r] + (bias ? bias[c] * t[r] : 0): raw_out * mask_ratio, the bias form
 *   ((raw - b) * ratio + b) * update_mask = conv_nobias * ratio + b * update_mask (ratio already carries the
 *   update_mask factor), and -- with x == NULL -- the broadcast of update_mask to the Cout mask channels.
 *   Both are their own adjoints w.r.t. x.
 * bg_resize_nearest3d_*: F.interpolate(size=..., mode='nearest') (infill3d.py:217-222), source index
 *   floor(dst * in / out); _bwd is the adjoint (overwrites dx).
 * bg_resize_trilinear3d_*: F.interpolate(size=..., mode='trilinear') as PConvUNet3d(upsampling_mode='trilinear') calls it
 *   (infill3d.py:217-220; align_corners unset = False): source coordinate max(in/out * (o + 0.5) - 0.5, 0) per axis,
 *   identity where in == out; _bwd gathers per input voxel (the exact adjoint, no atomics; overwrites dx).
 * bg_pc_dropout: PCDropout3d.forward in training mode (infill3d.py:115-135) with the nn.Dropout3d draw handed in as
 *   keep[n][c] in {0,1} (fp32, row stride ldk): mask_out = mask * keep, y = x * (1 - (mask - mask_out)) / scale with
 *   scale = 1 - p.  The mask is EITHER `rows` (fp32 per pixel: an update_mask, whose channels are equal) OR `mask` (a
 *   per-channel tensor); mask_out may be NULL (the adjoint w.r.t. x is the same call on the gradient).
 * bg_tv_loss_*: total_variation_loss (utils/losses.py:40-44) as it acts on a contiguous fp32 5-D tensor viewed as
 *   [A][D][H][W]: mean |shift along H| + mean |shift along D|; loss accumulated (caller zeroes), dx = coef[0] * d loss/dx.
 * ------------------------------------------------------------------------- */
int bg_mask_window(int32_t dtype, const void* mask, int32_t ld, int32_t C, const float* rows0, int32_t c0, const float* rows1,
                   int32_t c1, int32_t N, int32_t D, int32_t H, int32_t W, int32_t Do, int32_t Ho, int32_t Wo, int32_t k,
                   int32_t stride, int32_t pad, int32_t planar, float eps, float* update_mask, float* ratio, void* stream);
int bg_resize_nearest3d_rows(const float* x, float* y, int32_t N, int32_t Di, int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho,
                             int32_t Wo, void* stream);
int bg_mul_rows(int32_t dtype, const void* x, int32_t ldx, const void* m, int32_t ldm, void* y, int32_t ldy, int64_t rows,
                int32_t C, void* stream);
int bg_scale_rows(int32_t dtype, const void* x, int32_t ldx, const float* s, const float* bias, const float* t, void* y,
                  int32_t ldy, int64_t rows, int32_t C, void* stream);
int bg_resize_nearest3d_fwd(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t Di, int32_t Hi,
                            int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, int32_t C, void* stream);
int bg_resize_nearest3d_bwd(int32_t dtype, const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t N, int32_t Di,
                            int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, int32_t C, void* stream);
int bg_resize_trilinear3d_fwd(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t Di, int32_t Hi,
                              int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, int32_t C, void* stream);
int bg_resize_trilinear3d_bwd(int32_t dtype, const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t N, int32_t Di,
                              int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, int32_t C, void* stream);
int bg_pc_dropout(int32_t dtype, const void* x, int32_t ldx, const float* rows, const void* mask, int32_t ldm, const float* keep,
                  int32_t ldk, void* y, int32_t ldy, void* mask_out, int32_t ldmo, int64_t nrows, int64_t rows_per_sample,
                  int32_t C, float scale, void* stream);
/* y = m*a + (1-m)*b on flat fp32 arrays (output_comp, utils/losses.py:71); a == NULL: y = (1-m)*b (its adjoint w.r.t. b). */
int bg_blend_f32(const float* m, const float* a, const float* b, float* y, int64_t n, void* stream);
int bg_tv_loss_fwd(const float* x, int64_t A, int32_t D, int32_t H, int32_t W, float* loss, void* stream);
int bg_tv_loss_bwd(const float* x, int64_t A, int32_t D, int32_t H, int32_t W, const float* coef, float* dx, void* stream);

/* ---------------------------------------------------------------------------
 * Normalisation (BatchNorm2d train/eval, InstanceNorm2d, Identity) fused with
 * the residual add (Block: x += skip, deeplab.py:141) and LeakyReLU(0.2)
 * (deeplab.py:100,180,334,365).
 *
 * Rows are pixels: R = N*H*W rows of C channels.  `groups` = 1 for batch
 * statistics, N for instance statistics; rows_per_group = R / groups.
 * ------------------------------------------------------------------------- */
/* sum[g,c] += sum_x, sumsq[g,c] += sum_x^2 (fp64, caller zeroes). */
int bg_norm_stats(int32_t dtype, const void* x, int64_t rows, int32_t C, int32_t ldx, int32_t groups, double* sum,
                  double* sumsq, void* stream);
/* From the sums: mean/rstd (saved for backward, fp32 [groups,C]), the affine
 * (scale, shift) the apply kernel uses, and the BatchNorm running-stat update
 * (momentum, unbiased variance; running_* may be NULL).  gamma/beta may be NULL
 * (InstanceNorm defaults: no affine).  A group of ONE element normalises to
 * exactly 0 (mean = x, var = 0), the arithmetic the reference's torch 1.8 had. */
int bg_norm_finalize(const double* sum, const double* sumsq, int64_t rows_per_group, int32_t groups, int32_t C,
                     const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                     float* running_var, float* mean, float* rstd, float* scale, float* shift, void* stream);
/* bg_norm_finalize with the arithmetic of bg_norm_act_fwd_stats' fused finalize (fp32 rsqrt, the affine formed the
 * way the backward kernels re-form it) and one running-statistics update per statistic group, in group order:
 * for consumers that apply the affine themselves (bg_dwconv3x3_fwd_pre). */
int bg_norm_finalize_affine(const double* sum, const double* sumsq, int64_t rows_per_group, int32_t groups, int32_t C,
                            const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, float* mean, float* rstd, float* scale, float* shift, void* stream);
/* eval mode: scale/shift from running statistics. */
int bg_norm_eval_affine(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* scale, float* shift, void* stream);
/* y = act( x*scale[g,c] + shift[g,c] + res ), scale/shift NULL = identity, res
 * NULL = none, act: 0 none, 1 LeakyReLU(0.2), 2 ReLU (every act argument below).  y may alias x or res. */
int bg_norm_act_fwd(int32_t dtype, const void* x, int32_t ldx, const float* scale, const float* shift,
                    const void* res, int32_t ldres, void* y, int32_t ldy, int64_t rows, int32_t C, int32_t groups,
                    int32_t act, void* stream);
/* bg_norm_finalize + bg_norm_act_fwd in ONE launch (training-mode statistics): every thread derives
 * the affine of its channels from the fp64 sums; the first row-block of each group also writes
 * mean/rstd (fp32 [groups,C]).  running_* (BatchNorm) get one momentum update per group, in group
 * order, from group 0's first row-block: groups > 1 with running statistics is BatchNorm over
 * sub-batches that the reference pushes through the layer in separate calls. */
int bg_norm_act_fwd_stats(int32_t dtype, const void* x, int32_t ldx, const double* sum, const double* sumsq,
                          const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, float* mean, float* rstd, const void* res, int32_t ldres, void* y,
                          int32_t ldy, int64_t rows, int32_t C, int32_t groups, int32_t act, void* stream);
/* Backward pass 1: g = dy * act'(y);  s1[g,c] += sum g, s2[g,c] += sum g*xhat
 * (fp64, caller zeroes) with xhat = (x-mean)*rstd.  y == NULL with act != 0 (layers WITHOUT a
 * residual): the LeakyReLU branch is taken from the recomputed pre-activation
 * x*(gamma*rstd) + (beta - mean*gamma*rstd) -- the forward kernel's own arithmetic -- so the
 * activated tensor is not read again (gamma/beta NULL = 1/0). */
int bg_norm_act_bwd_reduce(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                           int32_t ldx, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           int64_t rows, int32_t C, int32_t groups, int32_t act, double* s1, double* s2, void* stream);
/* Coefficients of pass 2 (dx = A*g + B*x + Cc) and dgamma += sum_g s2, dbeta += sum_g s1
 * (dgamma/dbeta/gamma may be NULL).  train != 0: batch/instance statistics
 * (full formula); train == 0: dx = scale*g. */
int bg_norm_bwd_finalize(const double* s1, const double* s2, int64_t rows_per_group, int32_t groups, int32_t C,
                         const float* gamma, const float* mean, const float* rstd, int32_t train, float* A, float* B,
                         float* Cc, float* dgamma, float* dbeta, void* stream);
/* Backward pass 2: g = dy*act'(y); dx = A*g + B*x + Cc (A NULL: dx = g; dx NULL:
 * skipped); dres = g (NULL: skipped).  dx may alias dy. */
int bg_norm_act_bwd_apply(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                          int32_t ldx, const float* A, const float* B, const float* Cc, void* dx, int32_t lddx,
                          void* dres, int32_t lddres, int64_t rows, int32_t C, int32_t groups, int32_t act,
                          void* stream);

/* bg_norm_bwd_finalize + bg_norm_act_bwd_apply in ONE launch: coefficients from s1/s2 per thread,
 * dgamma += sum_g s2 and dbeta += sum_g s1 by the first row-block (both may be NULL).
 * y == NULL with act != 0: as in bg_norm_act_bwd_reduce. */
int bg_norm_act_bwd_apply_stats(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                                int32_t ldx, const double* s1, const double* s2, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, int32_t train, float* dgamma, float* dbeta,
                                void* dx, int32_t lddx, void* dres, int32_t lddres, int64_t rows, int32_t C,
                                int32_t groups, int32_t act, void* stream);

/* ---------------------------------------------------------------------------
 * Resampling / pooling / layout (deeplab.py:375,379,663 bilinear
 * align_corners=True; :621 AdaptiveAvgPool2d((1,1)); torch.cat by ld slices).
 * ------------------------------------------------------------------------- */
int bg_resize_bilinear_fwd(int32_t in_dtype, int32_t out_dtype, const void* x, int32_t ldx, void* y, int32_t ldy,
                           int32_t N, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C, void* stream);
/* dx (overwritten) = adjoint of the above applied to dy. */
int bg_resize_bilinear_bwd(int32_t dy_dtype, int32_t dx_dtype, const void* dy, int32_t lddy, void* dx, int32_t lddx,
                           int32_t N, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C, void* stream);
/* out[g,c] (fp32, caller zeroes) += scale * sum over the group's rows of x. */
int bg_colsum(int32_t dtype, const void* x, int32_t ldx, int64_t rows, int32_t C, int32_t groups, float scale,
              float* out, void* stream);
/* y[row,c] = scale * v[g(row),c]  (v fp32 [groups,C]). */
int bg_broadcast_rows(int32_t dtype, const float* v, float scale, void* y, int32_t ldy, int64_t rows, int32_t C,
                      int32_t groups, void* stream);
/* Strided 2-D copy with dtype conversion: dst[row*ldd + c] = src[row*lds + c], c < C. */
int bg_cast_rows(int32_t src_dtype, int32_t dst_dtype, const void* src, int32_t lds, void* dst, int32_t ldd,
                 int64_t rows, int32_t C, void* stream);
/* NCHW fp32 (contiguous) -> NHWC dtype with channels [C, Cp) zero-filled, and back
 * (adjoint direction used for gradients: NHWC -> NCHW fp32, first C channels). */
int bg_nchw_to_nhwc(int32_t dst_dtype, const float* src, void* dst, int32_t N, int32_t C, int32_t HW, int32_t Cp,
                    int32_t ldd, void* stream);
int bg_nhwc_to_nchw(int32_t src_dtype, const void* src, int32_t lds, float* dst, int32_t N, int32_t C, int32_t HW,
                    void* stream);
int bg_fill_f32(float* p, float v, int64_t n, void* stream);
/* nn.AvgPool2d(2, stride=1, padding=p) of the Deconv upsamplers (deeplab.py:402-429,469-471,
 * 649-651; count_include_pad): y[n,h,w,:] = 0.25 * sum_{i,j in {0,1}} x[n,h+off+i,w+off+j,:], zero
 * outside x.  Forward: off = -p, Ho = Hi+2p-1.  Backward: the same call with x := dy, y := dx,
 * off = p-1 and the sizes exchanged. */
int bg_avgpool2x2(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int32_t N, int32_t Hi, int32_t Wi,
                  int32_t Ho, int32_t Wo, int32_t C, int32_t off, void* stream);
int bg_axpy_rows(int32_t dtype, const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t C,
                 void* stream); /* y += x */

/* ---------------------------------------------------------------------------
 * Discriminator head nn.Linear(2048*h*w, 1) on the NCHW-flattened feature map
 * (deeplab_gan.py:21,32-35).  x: NHWC features [N,HW,C]; wlin: fp32 in the
 * reference's layout [C*HW] (index c*HW + p), so checkpoints interchange.
 * ------------------------------------------------------------------------- */
int bg_linear_head_fwd(int32_t dtype, const void* x, int32_t ldx, const float* wlin, const float* blin, float* logits,
                       int32_t N, int32_t HW, int32_t C, void* stream);
/* dx = dlogits[n] * w ; dw += sum_n dlogits[n]*x ; db += sum_n dlogits[n] (dw/db may be NULL). */
int bg_linear_head_bwd(int32_t dtype, const void* x, int32_t ldx, const float* wlin, const float* dlogits, void* dx,
                       int32_t lddx, float* dw, float* db, int32_t N, int32_t HW, int32_t C, void* stream);

/* ---------------------------------------------------------------------------
 * Losses (utils/losses.py:129-172, train_gan.py:142-152, deeplab_gan.py:112).
 * Each writes the scalar loss to *loss (device fp32) and, where given, the
 * gradient for upstream weight 1 (callers scale).
 * ------------------------------------------------------------------------- */
/* mean_n( max(x,0) - x*y + log1p(exp(-|x|)) ); dx[n] = (sigmoid(x)-y)/N. */
int bg_bce_logits(const float* x, const float* y, int32_t n, float* loss, float* dx, void* stream);
/* Generator output and targets are NCHW fp32 at the module boundary (as in the
 * reference), so the pixel losses run on flat fp32 arrays.
 * loss += inv_norm * sum(|p-t|*w)  (w NULL = 1; caller zeroes loss). */
int bg_l1_loss_fwd(const float* p, const float* t, const float* w, int64_t n, float inv_norm, float* loss,
                   void* stream);
/* dp = sign(p-t) * w * coef[0] * inv_norm   (coef: device scalar = upstream gradient). */
int bg_l1_loss_bwd(const float* p, const float* t, const float* w, int64_t n, float inv_norm, const float* coef,
                   float* dp, void* stream);
/* The other regression criteria of train_gan.py:142-152 on the same flat arrays.  kind: 0 = |d|
 * (= bg_l1_loss_*), 1 = nn.SmoothL1Loss (beta = 1), 2 = nn.MSELoss; d = p - t; weights as above
 * (losses.py:101-128: L1LossWeighted(smooth=...), L2LossWeighted). */
int bg_pixel_loss_fwd(int32_t kind, const float* p, const float* t, const float* w, int64_t n, float inv_norm,
                      float* loss, void* stream);
int bg_pixel_loss_bwd(int32_t kind, const float* p, const float* t, const float* w, int64_t n, float inv_norm,
                      const float* coef, float* dp, void* stream);
/* loss += inv_norm * sum over (n,pixel) of (||g[n,:,pixel]||_2 - 1)^2, g NCHW fp32 (caller zeroes loss). */
int bg_gp_penalty(const float* g, int32_t N, int32_t C, int32_t HW, float inv_norm, float* loss, void* stream);

/* ---------------------------------------------------------------------------
 * Optimiser over a flat fp32 parameter arena (optim.Adam / AdamW,
 * utils/parsing_helpers.py:8-12) with the low-precision weight copy fused in.
 * g is multiplied by grad_scale first (1/world_size after a SUM all-reduce).
 * ------------------------------------------------------------------------- */
/* beta1 / beta2 are doubles: torch evaluates 1 - beta in double before rounding it to the tensors' fp32 (0.001f, not
 * 1.f - 0.999f, which is 1.3e-5 off), and checkpoints interchange the moments. */
int bg_adam_step(float* p, const float* g, float* m, float* v, void* p_lp /* bf16 copy or NULL */, int64_t n, float lr,
                 double beta1, double beta2, float eps, float weight_decay, int32_t decoupled, float bias_corr1,
                 float bias_corr2, float grad_scale, void* stream);
/* The same update with its step-dependent scalars in device memory: hyper = {lr, bias_corr1, bias_corr2, grad_scale}
 * (4 floats).  For a training step captured into a hipGraph: the host rewrites hyper before each replay. */
/* dst[0..n) = v0..v3 (n <= 4), values in the launch arguments: how the host hands bg_adam_step_dev its scalars without a
 * pinned staging buffer. */
int bg_set_floats(float* dst, int32_t n, float v0, float v1, float v2, float v3, void* stream);
int bg_adam_step_dev(float* p, const float* g, float* m, float* v, void* p_lp, int64_t n, const float* hyper, double beta1,
                     double beta2, float eps, float weight_decay, int32_t decoupled, void* stream);
/* LAMB as the reference selects it (utils/parsing_helpers.py:13-14: apex.optimizers.FusedLAMB(lr, eps, weight_decay), a
 * dependency that is not in the reference tree -- NVIDIA apex as shipped in nvcr.io/nvidia/pytorch:20.12-py3, docker/Dockerfile:1;
 * restated from its published two-stage form, csrc/multi_tensor_lamb.cu) over a flat arena cut into parameter tensors
 * [seg[t], seg[t+1]) (nseg + 1 offsets in device memory):
 *   bg_sumsq_f32:   *out += sum (scale * x[i])^2                              -- the global gradient norm (caller zeroes out)
 *   bg_lamb_stage1: s = grad_scale * g / max(1, ||g|| / max_grad_norm); adam_w_mode 0: s += wd p;
 *                   m = b1 m + b3 s (b3 = 1 - b1 with grad_averaging, else 1); v = b2 v + (1 - b2) s^2;
 *                   u = (m / bc1) / (sqrt(v / bc2) + eps) (+ wd p in adam_w_mode 1), written over g;
 *                   param_sumsq[t] += sum p^2, update_sumsq[t] += sum u^2       (caller zeroes both)
 *   bg_lamb_stage2: p -= ratio_t * u, ratio_t = lr * ||p_t|| / ||u_t|| where (weight_decay != 0 or use_nvlamb) and both
 *                   norms are non-zero, else lr; refreshes the bf16 copy p_lp (may be NULL). */
int bg_sumsq_f32(const float* x, int64_t n, float scale, double* out, void* stream);
int bg_lamb_stage1(const float* p, float* g, float* m, float* v, const int64_t* seg, int32_t nseg, const double* grad_sumsq,
                   float max_grad_norm, double beta1, double beta2, int32_t grad_averaging, float eps, float weight_decay,
                   int32_t adam_w_mode, float bias_corr1, float bias_corr2, float grad_scale, float* param_sumsq, float* update_sumsq,
                   void* stream);
int bg_lamb_stage2(float* p, const float* update, void* p_lp, const int64_t* seg, int32_t nseg, const float* param_sumsq,
                   const float* update_sumsq, float lr, float weight_decay, int32_t use_nvlamb, void* stream);
int bg_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);

/* ---------------------------------------------------------------------------
 * Data path: .npy header parser and the pinned-host -> HBM asynchronous staging
 * ring that replaces the reference's synchronous numpy_reader
 * (src/numpy_reader/cpp/numpy_reader.cpp:60-170 ParseFile, :403-431 readChunk,
 * :434-497 getSample/getBatch; pybind surface :501-536).
 * ------------------------------------------------------------------------- */
#define BG_E_IO (-3) /* file could not be opened / parsed / read completely */

typedef struct bg_npy_info {
    int32_t dtype_code;    /* 0 '<f4', 1 '<f8', 2 '<i4', 3 '<i8' (the reference's type map, numpy_reader.h:72-77) */
    int32_t typesize;
    int32_t fortran_order;
    int32_t ndim;          /* 0 for a scalar array */
    int64_t shape[8];
    int64_t data_offset;   /* first payload byte (format 1.0, 2.0 and 3.0 headers) */
    int64_t file_size;
} bg_npy_info;
/* HOST call.  Errors carry the reference's message fragments ("failed to open file", "not a numpy
 * file", "ill formatted or corrupt", "unsupported datatype", "big endian"). */
int bg_npy_parse(const char* path, bg_npy_info* out);

typedef struct bg_ring bg_ring;
/* device >= 0: n_slots pinned host buffers + n_slots device buffers of slot_bytes, one copy stream,
 * n_threads reader threads.  device = -1: host-only ring (plain page-aligned memory). */
int bg_ring_create(int32_t device, int32_t n_slots, int64_t slot_bytes, int32_t n_threads, bg_ring** out);
int bg_ring_destroy(bg_ring* r);
/* Start reading nbytes at file offset `offset` of `path` into a free slot, split into n_chunks
 * parallel preads; returns immediately with a ticket.  BG_E_ARG if every slot is in flight. */
int bg_ring_submit(bg_ring* r, const char* path, int64_t offset, int64_t nbytes, int32_t n_chunks, int64_t* ticket);
/* Wait for the read (host) and make consumer_stream wait for the H2D copy (device, no host block);
 * *ptr = device (or host) address of the payload, valid until bg_ring_release.  A short file gives
 * BG_E_IO with "file corruption" in the message. */
int bg_ring_acquire(bg_ring* r, int64_t ticket, void* consumer_stream, void** ptr);
/* acquire + asynchronous copy of the payload to dst (device memory on consumer_stream / host memory). */
int bg_ring_copy_out(bg_ring* r, int64_t ticket, void* dst, int64_t nbytes, void* consumer_stream);
/* Give the slot back; the next copy into it is ordered after consumer_stream's work so far. */
int bg_ring_release(bg_ring* r, int64_t ticket, void* consumer_stream);

#ifdef __cplusplus
}
#endif
#endif /* BGAMD_H */
