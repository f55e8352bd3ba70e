/* diffuvolume_hip.h -- C ABI of libdiffuvolume_hip.so (MI355X / gfx950 only).
 *
 * The reference (iSEE-Laboratory/DiffuVolume) has no FFI layer: its hot path is
 * Python calling ATen.  This header is the boundary a maintainer binds instead
 * (ctypes stub in INTEGRATION.md).  Each entry point names the reference code it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to a dense, contiguous row-major tensor in
 *    the reference's own layout (NCHW / NCDHW); outputs are caller-allocated;
 *  - nothing here allocates, frees, synchronises or keeps global mutable state;
 *    calls are re-entrant and are enqueued on `stream` of the current device;
 *  - return value: 0 = ok, <0 = DV_ERR_* (bad argument), >0 = hipError_t.
 */
#ifndef DIFFUVOLUME_HIP_H
#define DIFFUVOLUME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dv_stream_t; /* hipStream_t */

#define DV_OK 0
#define DV_ERR_NULL (-1)        /* required pointer is NULL */
#define DV_ERR_SHAPE (-2)       /* dimension <= 0 or inconsistent */
#define DV_ERR_UNSUPPORTED (-3) /* shape/option outside what the kernels implement */
#define DV_ERR_ALIGN (-4)       /* pointer not 16-byte aligned */

#define DV_ACT_NONE 0
#define DV_ACT_RELU 1      /* SceneFlow / ACVNet */
#define DV_ACT_MISH 2      /* KITTI12 / PCWNet: x*tanh(softplus(x)) */
#define DV_ACT_LEAKY 3     /* KITTI15 / IGEV: LeakyReLU(0.01) */
#define DV_ACT_SIGMOID 4   /* ConvGRU gates (KITTI15/core/update.py:36-37); dv_conv2d_* only */
#define DV_ACT_TANH 5      /* ConvGRU candidate state (update.py:38); dv_conv2d_* only */

int dv_version(void);
const char* dv_error_string(int code);

/* ---- cost-volume builders ------------------------------------------------
 * build_gwc_volume: SceneFlow/models/submodule.py:228-238 (+ groupwise_correlation
 * :209-215); identical in KITTI12 :109-119 and KITTI15 :159-169.
 * ref,tgt [B,C,H,W] -> out [B,G,D,H,W];  out[b,g,d,y,x] = mean_c ref*tgt(x-d), 0 for x<d. */
int dv_gwc_volume_f32(const float* ref, const float* tgt, float* out,
                      int B, int C, int H, int W, int D, int G, dv_stream_t stream);

/* build_concat_volume: SceneFlow/models/submodule.py:180-191 (zero_left=0, = KITTI15
 * :206-217) and KITTI12/models/submodule.py:86-97 (zero_left=1).
 * ref,tgt [B,C,H,W] -> out [B,2C,D,H,W]. */
int dv_concat_volume_f32(const float* ref, const float* tgt, float* out,
                         int B, int C, int H, int W, int D, int zero_left, dv_stream_t stream);

/* F.softmax(att_weights, dim=2) * build_concat_volume(...): SceneFlow/models/acv_ddim.py:388-390
 * fused.  att [B,1,D,H,W] logits -> out [B,2C,D,H,W]. */
int dv_concat_attn_volume_f32(const float* ref, const float* tgt, const float* att, float* out,
                              int B, int C, int H, int W, int D, dv_stream_t stream);
/* The same product from the softmax already taken (p [B,D,H,W] of dv_softmax_d_f32; identical bits): what
 * AttentionConcatVolume.tensor() runs when something other than the rank-1 layer wants the tensor of acv_ddim.py:390. */
int dv_concat_prob_volume_f32(const float* ref, const float* tgt, const float* p, float* out,
                              int B, int C, int H, int W, int D, dv_stream_t stream);

/* ---- the first aggregation layer of a DiffuVolume step on its factored input --------------------------------
 * SceneFlow/models/acv_ddim.py:254-262 feeds `volume * noise` to dres0[0] (convbn_3d(64,32,3,1,1) + ReLU, :200-203)
 * with volume = softmax(att, dim=2) * build_concat_volume(L, R) (:388-390) and noise a per-voxel scalar: the input is
 * s(b,d,y,x) * [L(y,x) ; R(y,x-d)], so the 3x3x3 convolution is sum_tap s(.) * (GL[tap] + GR[tap](x-d)) with GL / GR
 * the 1x1 convolutions of L / R with the layer's weights, built once per stereo pair (csrc/rank1_filter.hip).
 * dv_softmax_d_f32: att [B,D,HW] logits -> p [B,D,HW] = softmax over D (the arithmetic of dv_concat_attn_volume_f32).
 * dv_mul_f32: out = x * y elementwise (s = p * n01).
 * dv_conv3d_rank1_filter_f32: s [B,D,H,W], gl / gr [B,27*Cout,H,W] (channel = tap*Cout + co, tap = (kd*3+ky)*3+kx)
 *   -> out [B,Cout,D,H,W] = act(scale * conv + bias); D <= 48. */
/* The tables themselves: a 1x1 convolution from few input channels (Cin <= 32: the concat features) to many output
 * channels (27 * Cout of the layer), no BatchNorm / activation: in [B,Cin,HW] -> out [B,Cout,HW] = W [Cout,Cin] . in.
 * A pure HBM write stream (csrc/pointwise_expand.hip).  Weights are packed once (dv_pointwise_expand_packed_floats floats). */
size_t dv_pointwise_expand_packed_floats(int Cin, int Cout);
int dv_pointwise_expand_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout, dv_stream_t stream);
int dv_pointwise_expand_f32(const float* in, const float* wpacked, float* out, int B, int Cin, int HW, int Cout,
                            dv_stream_t stream);
int dv_softmax_d_f32(const float* att, float* p, int B, int D, int HW, dv_stream_t stream);
int dv_mul_f32(const float* x, const float* y, float* out, size_t n, dv_stream_t stream);
int dv_conv3d_rank1_filter_f32(const float* s, const float* gl, const float* gr, const float* ch_scale,
                               const float* ch_bias, float* out, int B, int D, int H, int W, int Cout, int act,
                               dv_stream_t stream);

/* ---- time-shifted noise -> [0,1] volume filter ----------------------------
 * DynamicHead add (SceneFlow/models/head.py:74-77) + clamp + rescale
 * (acv_ddim.py:256-258).  x_t [B,C,HW], shift [B,C] (fp32, the MLP output),
 * n01 = (clamp(x_t+shift,-1,1)+1)/2 in the state dtype; n01_f32 = (float)n01
 * is what multiplies the volume (acv_ddim.py:260).  The f32 flavour is the
 * first DDIM step (state still fp32), f64 the later ones. */
int dv_noise_prepare_f32(const float* x_t, const float* shift, float* n01,
                         int B, int C, int HW, dv_stream_t stream);
int dv_noise_prepare_f64(const double* x_t, const float* shift, double* n01, float* n01_f32,
                         int B, int C, int HW, dv_stream_t stream);

/* ---- 3-D aggregation convolutions ------------------------------------------
 * convbn_3d (+ReLU/Mish/LeakyReLU): SceneFlow/models/submodule.py:94-97 and its
 * uses acv_ddim.py:60-70,:82-83,:200-222.  Cubic kernel k in {1,3}, pad (k-1)/2,
 * stride in {1,2}; implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32).
 *   y = act( conv(in * in_scale) * ch_scale[co] + ch_bias[co] + residual )
 * in [B,Cin,D,H,W]; in_scale [B,D,H,W] or NULL (the `volume * noise.unsqueeze(1)`
 * prologue of acv_ddim.py:260); ch_scale/ch_bias [Cout] or NULL (eval-mode BN
 * folded by the caller); residual [B,Cout,Do,Ho,Wo] or NULL; out same shape.
 * `wpacked` comes from dv_conv3d_pack_weights_f32. */
size_t dv_conv3d_packed_floats(int Cin, int Cout, int k);
int dv_conv3d_pack_weights_f32(const float* w /*[Cout,Cin,k,k,k]*/, float* wpacked,
                               int Cin, int Cout, int k, dv_stream_t stream);
int dv_conv3d_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                  const float* in_scale, const float* residual, float* out,
                  int B, int Cin, int D, int H, int W, int Cout, int k, int stride, int act,
                  dv_stream_t stream);

/* Test hook: pins the tiling of the DIRECT stride-2 kernel (0 = the launcher's own choice from the size of one batch item,
 * 1 = 2 x 4 x 32 output tiles, 2 = 2 x 2 x 32); both tilings give the same bits.  Process-wide; returns DV_OK. */
int dv_conv3d_set_s2_tile(int mode);
/* Test hook, process-wide: the z-marching form of the single-output-channel head (conv3d_c1z_kernel) takes volumes of at
 * least `min_tiles` 16 x 64 tiles per batch item (0 = the default, 64); `segment` pins the planes per block (0 = the
 * launcher's choice from the block count; 3, 6 or 12 -- every segment length gives the same bits). */
int dv_conv3d_set_c1z(int min_tiles, int segment);

/* The same 3x3x3 stride-1 layer (Cout <= 32) on the fp16 matrix instruction with every fp32 operand
 * carried as hi+lo fp16 pairs: x*w ~= hi*hi' + hi*lo' + lo*hi', fp32 accumulate (csrc/conv3d_f16x3.hip).
 * Split error 2^-22 per operand, below the fp32 accumulation rounding; opt-in (DV_CONV_PRECISION=f16x3).
 * Activations must stay below 2.6e5 in magnitude (fp16 range after the 2^-2 pre-scale): if one does not,
 * or is NaN, *overflow_flag (device int, may be NULL) is set to 1 and the output is not to be trusted. */
size_t dv_conv3d_f16x3_packed_bytes(int Cin, int Cout);
int dv_conv3d_f16x3_pack_weights(const float* w /*[Cout,Cin,3,3,3]*/, void* wpacked, int Cin, int Cout,
                                 dv_stream_t stream);
int dv_conv3d_f16x3_f32(const float* in, const void* wpacked, const float* ch_scale, const float* ch_bias,
                        const float* in_scale, const float* residual, float* out, int* overflow_flag,
                        int B, int Cin, int D, int H, int W, int Cout, int act, dv_stream_t stream);

/* The same 3x3x3 stride-1 layer with the two in-plane taps computed by the Winograd minimal-filtering transform
 * F(2x2,3x3) (depth taps direct): 2.25x fewer multiplies, still on v_mfma_f32_16x16x4_f32 with fp32 operands and
 * accumulation (csrc/conv3d_wino.hip).  The transforms only add / subtract (the 1/2 factors are folded into the
 * packed weights), so results differ from dv_conv3d_f32 by fp32 rounding of a few extra additions per product.
 * Same arguments and epilogue as dv_conv3d_f32 (k = 3, stride = 1); `wpacked` from dv_conv3d_wino_pack_weights_f32. */
size_t dv_conv3d_wino_packed_floats(int Cin, int Cout);
int dv_conv3d_wino_pack_weights_f32(const float* w /*[Cout,Cin,3,3,3]*/, float* wpacked, int Cin, int Cout,
                                    dv_stream_t stream);
int dv_conv3d_wino_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                       const float* in_scale, const float* residual, float* out,
                       int B, int Cin, int D, int H, int W, int Cout, int act, dv_stream_t stream);

/* The same layer with ALL THREE taps by minimal filtering, F(2x2x2,3x3x3) (csrc/conv3d_wino3.hip): 8 instead of 12
 * multiplies per output and input channel, the depth positions of the transform on the four waves of a block (they meet in
 * LDS after the channel loop).  fp32 operands and accumulation on v_mfma_f32_16x16x4_f32; the transforms only add /
 * subtract, measured error no larger than the in-plane form's (tools/probes/wino_f222_numerics.py).  No filter prologue
 * (a call with `in_scale` takes dv_conv3d_wino_f32).  dv_conv3d_wino3_supported: 1 when four channel volumes of the
 * input fit 31-bit byte offsets. */
size_t dv_conv3d_wino3_packed_floats(int Cin, int Cout);
int dv_conv3d_wino3_supported(int Cin, int Cout, int D, int H, int W);
int dv_conv3d_wino3_pack_weights_f32(const float* w /*[Cout,Cin,3,3,3]*/, float* wpacked, int Cin, int Cout,
                                     dv_stream_t stream);
int dv_conv3d_wino3_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                        const float* residual, float* out, int B, int Cin, int D, int H, int W, int Cout, int act,
                        dv_stream_t stream);

/* The 3x3x3 STRIDE-2 layer (hourglass conv1 / conv3: acv_ddim.py:60,:66; pwcnet_ddim.py:137-147) in its polyphase
 * minimal-filtering form (csrc/conv3d_s2pp.hip): per in-plane axis the odd input phase sees a 2-tap filter, done as
 * F(2,2), the even phase one tap -- 25 instead of 36 multiplies per 2x2 outputs, depth taps direct, still on
 * v_mfma_f32_16x16x4_f32 with fp32 operands and accumulation.  The transforms only subtract (weights are summed once
 * by the pack), so results differ from dv_conv3d_f32(stride = 2) by fp32 rounding of one extra subtraction per operand.
 *   y = act( conv_s2(in) * ch_scale[co] + ch_bias[co] + residual )
 * 64 output channels per block and rows that travel as 16-byte quads: dv_conv3d_s2pp_supported says whether a layer
 * qualifies (Cout a multiple of 64, W a multiple of 4, D*H*W*4 <= 2^30; `in` 16-byte aligned), dv_conv3d_s2pp_f32 returns
 * DV_ERR_UNSUPPORTED otherwise.  Persistent launch: one persistent block per CU, two tiles at a time.
 * `wpacked` from dv_conv3d_s2pp_pack_weights_f32. */
int dv_conv3d_s2pp_supported(int Cin, int Cout, int D, int H, int W);
size_t dv_conv3d_s2pp_packed_floats(int Cin, int Cout);
int dv_conv3d_s2pp_pack_weights_f32(const float* w /*[Cout,Cin,3,3,3]*/, float* wpacked, int Cin, int Cout,
                                    dv_stream_t stream);
int dv_conv3d_s2pp_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                       const float* residual, float* out,
                       int B, int Cin, int D, int H, int W, int Cout, int act, dv_stream_t stream);

/* nn.ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1, bias=False) + BN
 * + skip add + activation: acv_ddim.py:74-80 and :91-92.  w [Cin,Cout,3,3,3].
 * in [B,Cin,D,H,W] -> out [B,Cout,2D,2H,2W]. */
size_t dv_deconv3d_packed_floats(int Cin, int Cout);
int dv_deconv3d_pack_weights_f32(const float* w /*[Cin,Cout,3,3,3]*/, float* wpacked,
                                 int Cin, int Cout, dv_stream_t stream);
int dv_deconv3d_k3s2_f32(const float* in, const float* wpacked, const float* ch_scale,
                         const float* ch_bias, const float* residual, float* out,
                         int B, int Cin, int D, int H, int W, int Cout, int act, dv_stream_t stream);

/* The k3 flavour has two launch shapes behind dv_deconv3d_k3s2_f32 / dv_deconv3d_k3s2_redir_f32 (same packed weights, same
 * arithmetic per output up to the order in which the fused skip channels join the sum): one-tile blocks
 * (csrc/deconv3d.hip) and one persistent block per CU with loader waves (csrc/deconv3d_pl.hip).
 * dv_deconv3d_pl_supported: 1 if the persistent kernel takes the shape (Cskip = 0: no fused skip convolution) -- whole
 * 8-channel input chunks, 32-channel output blocks, W % 4 == 0, Cskip % 4 == 0 and Cskip / 4 <= Cin / 8, 32-bit byte offsets
 * inside a batch item; the entry points also want 16-byte aligned tensors for it.  The choice never looks at the batch size.
 * dv_deconv3d_set_impl (test hook, process-wide): 0 = the launcher picks, 1 = one-tile blocks, 2 = persistent where supported. */
int dv_deconv3d_pl_supported(int Cin, int Cout, int D, int H, int W, int Cskip);
int dv_deconv3d_set_impl(int mode);
/* Test hook, process-wide: the persistent kernel launches at most n blocks (0 = one per CU), so that small volumes exercise
 * long per-block tile lists; results do not depend on it. */
int dv_deconv3d_pl_set_max_blocks(int n);

/* nn.ConvTranspose3d(kernel 4, stride 2, padding 1, bias=False) [+BN +LeakyReLU]: the IGEV hourglass's
 * conv3_up / conv2_up / conv1_up (KITTI15/core/igev_stereo_ddim.py:44-51, BasicConv deconv core/submodule.py:9-35).
 * w [Cin,Cout,4,4,4]; in [B,Cin,D,H,W] -> out [B,Cout,2D,2H,2W]; same fused epilogue as the k3 flavour. */
size_t dv_deconv3d_k4_packed_floats(int Cin, int Cout);
int dv_deconv3d_k4_pack_weights_f32(const float* w /*[Cin,Cout,4,4,4]*/, float* wpacked,
                                    int Cin, int Cout, dv_stream_t stream);
int dv_deconv3d_k4s2_f32(const float* in, const float* wpacked, const float* ch_scale,
                         const float* ch_bias, const float* residual, float* out,
                         int B, int Cin, int D, int H, int W, int Cout, int act, dv_stream_t stream);

/* ---- 2-D refinement stack (KITTI12 per-step disparity refinement) ------------------------------------
 * convbn (KITTI12/models/submodule.py:21-24: Conv2d(k, stride 1, padding = dilation, bias=False) + BatchNorm2d)
 * [+ Mish] and BasicBlock's `out += x` (:192-215), as used by refinenet_version3 (pwcnet_ddim.py:251-306):
 *   out = act( conv2d(in, w; dilation) * ch_scale + ch_bias + residual ),  k = 3 (dilation 1..16) or 1.
 * in [B,Cin,H,W], w [Cout,Cin,k,k], out / residual [B,Cout,H,W].  Weights are packed once per layer. */
size_t dv_conv2d_packed_floats(int Cin, int Cout, int k, int dilation);
int dv_conv2d_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout, int k, int dilation,
                               dv_stream_t stream);
int dv_conv2d_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                  const float* residual, float* out, int B, int Cin, int H, int W, int Cout, int k,
                  int dilation, int act, dv_stream_t stream);

/* The same convolution with the ConvGRU gate arithmetic of KITTI15/core/update.py:26-40 in its epilogue:
 *   v = act( conv2d(in, w) * ch_scale + ch_bias + residual );
 *   if (mul)      v = v * mul;                                   r * h            (update.py:38, the `r*h` operand)
 *   if (blend_z)  v = blend_h + blend_z * (v - blend_h);         (1-z)*h + z*q    (update.py:39)
 * mul / blend_z / blend_h are [B,Cout,H,W] or NULL; act may also be DV_ACT_SIGMOID / DV_ACT_TANH. */
int dv_conv2d_gated_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                        const float* residual, const float* mul, const float* blend_z, const float* blend_h,
                        float* out, int B, int Cin, int H, int W, int Cout, int k, int dilation, int act,
                        dv_stream_t stream);

/* The gated convolution over a VIRTUAL channel concatenation: `conv(torch.cat(inputs, dim=1))` without the copy
 * (ConvGRU.forward, KITTI15/core/update.py:33-38: hx = cat[h, x...], cat[r*h, x...]).  `inputs` / `channels` are HOST
 * arrays of n_inputs (1..4) device pointers [B,channels[i],H,W] and their channel counts; the weights are those
 * of the full convolution (Cin = sum of channels), packed as usual. */
int dv_conv2d_cat_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                      const float* ch_scale, const float* ch_bias, const float* residual, const float* mul,
                      const float* blend_z, const float* blend_h, float* out, int B, int H, int W, int Cout,
                      int k, int dilation, int act, dv_stream_t stream);

/* K-split form of dv_conv2d_cat_f32 for launches too small to fill the chip (a single IGEV pair at 1/8 and 1/16
 * resolution: KITTI15/core/update.py:33-40 at 48x156 / 24x78 pixels): `kslices` blocks share an output tile, each
 * sums a contiguous range of the input-channel chunks into scratch[kslices][B,Cout,H,W]; a second small kernel adds
 * the slices in a fixed order (deterministic) and applies the same fused epilogue.  kslices must be the value
 * dv_conv2d_auto_kslices returns for this shape (1 = use dv_conv2d_cat_f32); scratch is caller-allocated.  The factor is a
 * function of ONE batch item (`B` is ignored since round 5): it fixes the summation order, and a shard of a batch must
 * reproduce the batch's bits. */
int dv_conv2d_auto_kslices(int B, int Cin, int H, int W, int Cout, int k, int dilation);
int dv_conv2d_cat_ksplit_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                             const float* ch_scale, const float* ch_bias, const float* residual, const float* mul,
                             const float* blend_z, const float* blend_h, float* out, float* scratch, int kslices,
                             int B, int H, int W, int Cout, int k, int dilation, int act, dv_stream_t stream);

/* Stride-2 flavour (the down-sampling layers of the 2-D feature CNNs: SceneFlow/models/acv_ddim.py:19-21, :28 --
 * convbn(k 3, stride 2, pad 1) and the 1x1 stride-2 `downsample`): out [B,Cout,(H-1)/2+1,(W-1)/2+1], dilation 1,
 * residual (if any) has the output's shape.  Same packed weights as dv_conv2d_f32 with dilation 1. */
int dv_conv2d_s2_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                     const float* residual, float* out, int B, int Cin, int H, int W, int Cout, int k, int act,
                     dv_stream_t stream);

/* The 3x3, dilation-1, stride-1 case of dv_conv2d_cat_f32 in the Winograd F(2x2,3x3) form (csrc/conv2d_wino.hip): 2.25x
 * fewer multiplies on the same fp32 MFMA instruction, same epilogue (scale/bias, residual, activation, `mul`, GRU
 * blend) and the same virtual channel concatenation of up to four inputs; results equal dv_conv2d_cat_f32 up to fp32
 * rounding.  `wpacked` from dv_conv2d_wino_pack_weights_f32 (its own layout). */
size_t dv_conv2d_wino_packed_floats(int Cin, int Cout);
int dv_conv2d_wino_pack_weights_f32(const float* w /*[Cout,Cin,3,3]*/, float* wpacked, int Cin, int Cout,
                                    dv_stream_t stream);
int dv_conv2d_wino_cat_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                           const float* ch_scale, const float* ch_bias, const float* residual, const float* mul,
                           const float* blend_z, const float* blend_h, float* out, int B, int H, int W, int Cout,
                           int act, dv_stream_t stream);
/* Two 3x3 convolutions of the SAME input in one launch: ConvGRU's z and r gates (KITTI15/core/update.py:33-35,
 * `convz(hx)` and `convr(hx)`).  `wpacked` / `ch_scale` / `ch_bias` hold the Cout1 + Cout2 output channels back to back
 * (pack the concatenated weight; Cout1 % 32 == 0); channels < Cout1 go to out1 [B,Cout1,H,W] with residual1 / mul1,
 * the others to out2 [B,Cout2,H,W] with residual2 / mul2:  out_g = act(conv_g * scale + bias + residual_g) * mul_g. */
int dv_conv2d_wino_cat_pair_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                                const float* ch_scale, const float* ch_bias, const float* residual1, const float* mul1,
                                float* out1, const float* residual2, const float* mul2, float* out2, int B, int H,
                                int W, int Cout1, int Cout2, int act, dv_stream_t stream);
/* The same with dilation 1..16 (the refinement network's dilated layers, KITTI12/models/submodule.py:251-306, and the
 * dilated blocks of the 2-D feature CNNs): a dilation-d 3x3 convolution is d*d independent dilation-1 convolutions on the
 * sub-sampled images, so each block runs the Winograd kernel on a 16x16 tile of one sub-image.  Same packed weights. */
int dv_conv2d_wino_dil_cat_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                               const float* ch_scale, const float* ch_bias, const float* residual, const float* mul,
                               const float* blend_z, const float* blend_h, float* out, int B, int H, int W, int Cout,
                               int dilation, int act, dv_stream_t stream);

/* Input assembly of that refinement (KITTI12/models/pwcnet_ddim.py:486-502), fused:
 *   frw = warp(right, disp)  (models/submodule.py:137-176, incl. its align_corners mismatch and >= 0.999 mask),
 *   cv  = build_corrleation_volume(left, frw, maxshift, 1)  (:121-135, incl. its negative-shift slicing),
 *   out = cat(left - frw, left, Mish(du_a * disp + du_b), disp, cv)   [B, 3C + 1 + 2*maxshift + 1, H, W].
 * left/right [B,C,H,W] (C <= 32), disp [B,H,W]; du_a/du_b [C] = `dispupsample` (1x1 conv + BN) folded;
 * maxshift must be 24. */
int dv_refine_inputs_f32(const float* left, const float* right, const float* disp, const float* du_a,
                         const float* du_b, float* out, int B, int C, int H, int W, int maxshift,
                         dv_stream_t stream);

/* K-split forms of the two Winograd entries above, for launches that leave most of the chip empty (IGEV's ConvGRU at 1/16
 * resolution: 80 / 40 blocks per pair): `kslices` = dv_conv2d_wino_auto_kslices(Cin, H, W, Cout, dilation) blocks share an
 * output tile, each sums a contiguous range of the input-channel chunks into scratch [kslices][B,Cout,H,W] (Cout = Cout1 + Cout2
 * for the pair), and a second kernel adds the slices in slice order (deterministic) and applies the epilogue.  The factor
 * depends on ONE batch item only, so a shard of a batch reproduces the batch's bits.  kslices == 1: use the entries above. */
int dv_conv2d_wino_auto_kslices(int Cin, int H, int W, int Cout, int dilation);
int dv_conv2d_wino_cat_ksplit_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                                  const float* ch_scale, const float* ch_bias, const float* residual, const float* mul,
                                  const float* blend_z, const float* blend_h, float* out, float* scratch, int kslices, int B,
                                  int H, int W, int Cout, int dilation, int act, dv_stream_t stream);
int dv_conv2d_wino_cat_pair_ksplit_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                                       const float* ch_scale, const float* ch_bias, const float* residual1, const float* mul1,
                                       float* out1, const float* residual2, const float* mul2, float* out2, float* scratch,
                                       int kslices, int B, int H, int W, int Cout1, int Cout2, int act, dv_stream_t stream);

/* Space-to-batch for the dilated layers of refinenet_version3 (KITTI12/models/pwcnet_ddim.py:251-306), csrc/refine_inputs.hip.
 * dv_space_to_batch2_f32: [N,C,h,w] -> [4N,C,h/2,w/2]; sub-image (y & 1, x & 1) of item n becomes item 4n + 2(y & 1) + (x & 1)
 *   (h, w even; `in` 8-byte aligned).  A 3x3 convolution with dilation 2d on the input is one with dilation d on the output.
 * dv_batch_to_space_f32: the inverse of `levels` such steps at once: in [B * 4^levels, C, H >> levels, W >> levels] -> out [B,C,H,W]. */
int dv_space_to_batch2_f32(const float* in, float* out, int N, int C, int h, int w, dv_stream_t stream);
/* dv_conv2d_wino_cat_f32 (dilation 1, no residual / mul / blend) with its output stored through dv_space_to_batch2_f32's
 * mapping: out [4B,Cout,H/2,W/2] (H, W even).  The layer in front of a dilation doubling writes what the next layer reads. */
int dv_conv2d_wino_s2b_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                           const float* ch_scale, const float* ch_bias, float* out, int B, int H, int W, int Cout,
                           int act, dv_stream_t stream);
int dv_batch_to_space_f32(const float* in, float* out, int B, int C, int H, int W, int levels, dv_stream_t stream);

/* FeatureAtt.forward's broadcast product (KITTI15/core/submodule.py:234-239):
 * out[b,c,d,y,x] = sigmoid(logit[b,c,y,x]) * cv[b,c,d,y,x]; out may alias cv. */
int dv_feature_gate_f32(const float* cv /*[B,C,D,H,W]*/, const float* logit /*[B,C,H,W]*/, float* out,
                        int B, int C, int D, int H, int W, dv_stream_t stream);

/* The hourglass tail `F.relu(conv6(x) + redir1(skip))` (SceneFlow/models/acv_ddim.py:81-86, :91-92; KITTI12
 * pwcnet_ddim.py:236-248 with Mish) in ONE launch: the 1x1x1 `redir` convolution of the skip tensor is folded into
 * the transposed convolution as extra K-steps.
 *   out = act( deconv3d_k3s2(in, w) + redir_w . skip + ch_bias )
 * Both BatchNorm scales must already be folded into `w` (before dv_deconv3d_pack_weights_f32) and `redir_w`
 * ([Cout][Cskip] row-major, device), their shifts summed into ch_bias.  skip [B,Cskip,2D,2H,2W].
 * Needs W % 2 == 0 and ceil(Cskip/8) <= ceil(Cin/8); otherwise DV_ERR_UNSUPPORTED (run the two layers apart). */
int dv_deconv3d_k3s2_redir_f32(const float* in, const float* wpacked, const float* ch_bias, const float* skip,
                               const float* redir_w, float* out, int B, int Cin, int D, int H, int W, int Cout,
                               int Cskip, int act, dv_stream_t stream);

/* The attention branch's depth-wise stencils on the gwc volume (SceneFlow/models/acv_ddim.py:181-188, :377-381):
 * out = cat_g( Conv3d_(1,3,3),dil_g( Conv3d_(1,3,3)(gwc) ) ), both per channel (groups = channels), zero padded.
 * gwc/out [B,G,D,H,W]; w1 = `patch` weights [G][9]; w2 = the matching `patch_l1/l2/l3` weights [G][9];
 * dilation [G] (int32, 1..3).  All pointers on the device. */
int dv_patch_volume_f32(const float* gwc, const float* w1, const float* w2, const int* dilation, float* out,
                        int B, int G, int D, int H, int W, dv_stream_t stream);
/* The same pass with the dilation given as host-side runs of consecutive groups (run r: groups run_g0[r] ..
 * run_g0[r] + run_ng[r] - 1 at dilation run_dil[r]; the runs tile 0..G-1 in order): the 16-byte fast path needs the
 * dilation at launch time.  `dilation_dev` (device, [G]) serves the element-wise kernel when W % 4 != 0. */
int dv_patch_volume_runs_f32(const float* gwc, const float* w1, const float* w2, const int* dilation_dev, float* out,
                             int B, int G, int D, int H, int W, int nruns, const int* run_g0, const int* run_ng,
                             const int* run_dil, dv_stream_t stream);

/* attention_block.forward: SceneFlow/models/submodule.py:398-429 -- 4x4x4 window
 * multi-head self-attention (heads x C/heads), qkv Linear(C,3C)+bias, softmax,
 * final 1x1x1 Conv3d(C,C)+bias.  x [B,C,D,H,W] -> out same shape.  D must be a
 * multiple of 4; H,W not multiples of 4 are zero-padded with the reference's
 * -1000 mask rule.  C == 128, heads == 16 in every reference use. */
int dv_window_attn3d_f32(const float* x, const float* qkv_w /*[3C,C]*/, const float* qkv_b /*[3C]*/,
                         const float* proj_w /*[C,C]*/, const float* proj_b /*[C]*/, float* out,
                         int B, int C, int D, int H, int W, int heads, dv_stream_t stream);

/* ---- regression tail ---------------------------------------------------------
 * F.upsample(trilinear, x4) + F.softmax(dim=1) + disparity_regression
 * (acv_ddim.py:267-270, submodule.py:173-177) and the uncertainty
 * sum_k |disp-k| p_k (acv_ddim.py:325-329), fused: the [B,4D,4h,4w] volumes are
 * never materialised.  cost [B,D,h,w] -> disp [B,4h,4w], unc [B,4h,4w] (or NULL).
 * align_corners: 0 = SceneFlow, 1 = KITTI12 (pwcnet_ddim.py:480). */
int dv_upsample_softmax_regress_f32(const float* cost, float* disp, float* unc,
                                    int B, int D, int h, int w, int align_corners, dv_stream_t stream);

/* Uncertainty about an externally supplied disparity (KITTI12: the 2-D-refined `disp_finetune`,
 * pwcnet_ddim.py:548-552): unc = sum_k |disp - k| * softmax(trilinear(cost))_k.
 * cost [B,D,h,w], disp [B,4h,4w] (input) -> unc [B,4h,4w]. */
int dv_upsample_softmax_uncertainty_f32(const float* cost, const float* disp, float* unc,
                                        int B, int D, int h, int w, int align_corners, dv_stream_t stream);

/* disparity_regression on a materialised probability volume (submodule.py:173-177):
 * prob [B,D,H,W] -> disp [B,H,W]. */
int dv_disparity_regression_f32(const float* prob, float* disp, int B, int D, int H, int W,
                                dv_stream_t stream);

/* F.softmax(dim=1) + disparity_regression at the volume's own resolution (IGEV init_disp,
 * KITTI15/core/igev_stereo_ddim.py:382-383): cost [B,D,H,W] -> disp [B,H,W]. */
int dv_softmax_regress_f32(const float* cost, float* disp, int B, int D, int H, int W, dv_stream_t stream);

/* two-hot encoding of a quarter-resolution disparity (acv_ddim.py:403-419):
 * disp_q [B,h*w] -> x [B,nbins,h*w] = 2*twohot-1. */
int dv_encode_two_hot_f32(const float* disp_q, float* x, int B, int nbins, int hw, dv_stream_t stream);

/* One DDIM state update, everything after the regression of one step
 * (acv_ddim.py:272-294 and :318-362):
 *   x_start  = 2*twohot(bilinear_down4(clamp(disp,0,4*nbins-1))/4) - 1          (fp32 out)
 *   pred_eps = (sqrt_recip*n01 - x_start) / sqrt_recipm1                          (fp64)
 *   keep     = bilinear_down4( |disp-used|<dif_thr & unc<unc_thr );  mask = clamp(mask+keep,0,1)
 *   x_next   = last ? x_start : (mask==0 ? fill : x_start*sqrt_alpha_next + c*pred_eps + sigma*eps)
 *   ens     += cof * disp
 * n01_f32 / n01_f64: exactly one is non-NULL (state dtype of this step).
 * eps_f32 / eps_f64: exactly one non-NULL unless last (randn_like(img) follows img's dtype).
 * disp,unc,used,ens [B,4h,4w]; mask [B,h,w] in/out; fill,x_next,pred_eps fp64 [B,nbins,h,w];
 * pred_eps and ens may be NULL (not wanted). */
typedef struct dv_ddim_coef {
  double sqrt_recip_alpha;   /* sqrt(1/abar_t)     acv_ddim.py:156 */
  double sqrt_recipm1_alpha; /* sqrt(1/abar_t - 1) acv_ddim.py:157 */
  double sqrt_alpha_next;    /* sqrt(abar_next)    acv_ddim.py:356 */
  double c;                  /* sqrt(1-abar_next-sigma^2) acv_ddim.py:352 */
  double sigma;              /* acv_ddim.py:351 */
  float dif_thr;             /* 1  (acv_ddim.py:323) */
  float unc_thr;             /* 3  (acv_ddim.py:330) */
  float cof;                 /* ensemble weight of this step's disparity (acv_ddim.py:367) */
  int last;                  /* time_next < 0 (acv_ddim.py:344-346) */
  float clamp_max;           /* clamp of disp before the /4 downsample: 4*nbins-1 (ACV/PCW), nbins-1 (IGEV :265) */
  float ens_dif_thr;         /* >0: ensemble takes |disp-used|<thr ? disp : used (IGEV :323-327); 0: disp */
} dv_ddim_coef;

/* unc may be NULL (IGEV renewal mask tests the disparity only); coords0 [B,h,w] is NULL except for IGEV,
 * where the two-hot position is clamp(coords0 + disp_q, 0, nbins-1) (igev_stereo_ddim.py:268-272). */
int dv_ddim_step(const float* disp, const float* unc, const float* used, const float* coords0,
                 const float* n01_f32, const double* n01_f64,
                 const float* eps_f32, const double* eps_f64, const double* fill,
                 float* mask, float* x_start, double* pred_eps, double* x_next, float* ens,
                 int B, int nbins, int h, int w, const dv_ddim_coef* coef, dv_stream_t stream);

/* ---- IGEV: convex upsampling of the quarter-resolution disparity ------------------------
 * IGEVStereo_ddim.upsample_disp (KITTI15/core/igev_stereo_ddim.py:209-217) = F.softmax(spx_pred, 1) followed by
 * context_upsample (core/submodule.py:241-253):
 *   out[b,Y,X] = sum_{k=3ky+kx} p_k[b,Y,X] * scale * disp_low[b, (Y>>2)+ky-1, (X>>2)+kx-1]   (zeros outside)
 * disp_low [B,h,w]; weights [B,9,4h,4w] (logits when apply_softmax != 0, else probabilities); out [B,4h,4w];
 * scale = 4 in the reference (`disp*4.`).  weights / out must be 16-byte aligned. */
int dv_context_upsample_f32(const float* disp_low, const float* weights, float* out, int B, int h, int w,
                            float scale, int apply_softmax, dv_stream_t stream);

/* ---- IGEV: the small per-iteration operators around the ConvGRUs (KITTI15/core/update.py) ----
 * dv_conv2d_1in_f32: nn.Conv2d(1, Cout, k, padding=k/2) + bias + activation on a single-channel image (the motion
 *   encoder's 7x7 `convd1` on the disparity, update.py:86,:92).  in [B,1,H,W]; w [Cout,1,k,k] (k = 3, 5, 7); out [B,Cout,H,W].
 * dv_resize_bilinear_ac_f32: F.interpolate(x, (H,W), mode='bilinear', align_corners=True) (`interp`, update.py:100-102),
 *   PyTorch's source-index arithmetic.  in [BC,h,w] -> out [BC,H,W].
 * dv_avg_pool3s2_f32: F.avg_pool2d(x, 3, stride=2, padding=1) with the padded zeros counted (`pool2x`, update.py:96-97).
 *   in [BC,H,W] -> out [BC,(H-1)/2+1,(W-1)/2+1]. */
int dv_conv2d_1in_f32(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, int Cout,
                      int k, int act, dv_stream_t stream);
int dv_resize_bilinear_ac_f32(const float* in, float* out, int BC, int h, int w, int H, int W, dv_stream_t stream);
int dv_avg_pool3s2_f32(const float* in, float* out, int BC, int H, int W, dv_stream_t stream);

/* IGEV's once-per-pair 2-D front without MIOpen (csrc/igev_front.hip).
 * dv_conv2d_fewin_f32: nn.Conv2d(Cin <= 4, Cout, k in {3,5,7}, stride in {1,2}, padding=k/2) [+ bias] [+ folded eval
 *   BatchNorm: y * ch_scale + ch_shift, both or neither] + activation -- the 7x7 stride-2 stem of the context encoder
 *   (KITTI15/core/extractor.py:197) and the RGB stems; w [Cout,Cin,k,k]; out [B,Cout,(H-1)/stride+1,(W-1)/stride+1].
 * dv_instance_norm_act_f32: nn.InstanceNorm2d (affine=False) + activation over BC planes of HW floats
 *   (core/submodule.py:79-107, igev_stereo_ddim.py:100-117); in-place allowed (out == in). */
int dv_conv2d_fewin_f32(const float* in, const float* w, const float* bias, const float* ch_scale, const float* ch_shift,
                        float* out, int B, int Cin, int H, int W, int Cout, int k, int stride, int act,
                        dv_stream_t stream);
int dv_instance_norm_act_f32(const float* in, float* out, int BC, int HW, float eps, int act, dv_stream_t stream);

/* ---- IGEV: all-pairs correlation along the epipolar line + its level-1 pooling --------
 * Combined_Geo_Encoding_Volume.corr (KITTI15/core/geometry_ddim.py:72-80: einsum 'aijk,aijh->ajkh') and the
 * avg_pool2d([1,2]) of the pyramid (:28-30), once per stereo pair:
 *   corr0[b,y,x1,x2] = sum_c fmap1[b,c,y,x1] * fmap2[b,c,y,x2];   corr1[b,y,x1,x] = (corr0[..,2x] + corr0[..,2x+1]) / 2
 * fmap1 [B,C,H,W1], fmap2 [B,C,H,W2] (C <= 256); corr0 [B,H,W1,W2]; corr1 [B,H,W1,W2/2] (floor). */
int dv_allpairs_corr_f32(const float* fmap1, const float* fmap2, float* corr0, float* corr1, int B, int C, int H,
                         int W1, int W2, dv_stream_t stream);

/* ---- IGEV: geometry-encoding-volume lookup with the noise filter -----------------------
 * Combined_Geo_Encoding_Volume.__call__ (KITTI15/core/geometry_ddim.py:33-69), 2 pyramid levels,
 * radius 4: per pixel, (geo[c,:] * noise[:]) linearly sampled at disp/2^i + {-4..4} plus the all-pairs
 * correlation row sampled at coords/2^i - disp/2^i + {-4..4}.
 * geo [B,C,D,h,w]; corr0 [B,h,w,W2] (= corr() :72-80), corr1 [B,h,w,W2/2] (its avg_pool, :28-30);
 * disp, coords [B,h,w]; noisy: the [B,D,h,w] filter tensor, read as B*h*w rows of D floats (the
 * reference's raw reshape, :37); out [B, 2*(9C+9), h, w], channel order (geo0, corr0, geo1, corr1). */
int dv_geo_filter_lookup_f32(const float* geo, const float* corr0, const float* corr1,
                             const float* disp, const float* coords, const float* noisy, float* out,
                             int B, int C, int D, int h, int w, int W2, int radius, dv_stream_t stream);
/* The same lookup FUSED with the 1x1 convolution that consumes it in IGEV's update block (BasicMotionEncoder.convc1 +
 * bias + ReLU, KITTI15/core/update.py:79,:89): out[b, co, y, x] = act(bias[co] + sum_ch w[co, ch] * lookup[b, ch, y, x]),
 * out [B,64,h,w]; the [B,2*(C*9+9),h,w] lookup tensor is never written.  `wpacked` = dv_geo_lookup_conv1x1_pack_weights_f32 of
 * the nn.Conv2d weight [64, 2*(C*9+9)] (dv_geo_lookup_conv1x1_packed_floats(C) floats); Cout must be 64, radius 4. */
size_t dv_geo_lookup_conv1x1_packed_floats(int C);
int dv_geo_lookup_conv1x1_pack_weights_f32(const float* w, float* wpacked, int C, dv_stream_t stream);
int dv_geo_filter_lookup_conv1x1_f32(const float* geo, const float* corr0, const float* corr1, const float* disp,
                                     const float* coords, const float* noisy, const float* wpacked, const float* bias,
                                     float* out, int B, int C, int D, int h, int w, int W2, int radius, int Cout, int act,
                                     dv_stream_t stream);

/* ---- metrics (SceneFlow/utils/metrics.py:22-65) -------------------------------
 * Per-image sums over pixels with mask!=0: sums[b] = { n_mask, n_gt_pos, sum|gt-est|,
 * n_D1 (err>3 & err/|gt|>0.05), n_err>1, n_err>2, n_err>3, 0 } as fp64 [B,8].
 * The <0.1 mask-ratio skip and the means are taken by the caller. */
int dv_masked_metrics_f32(const float* est, const float* gt, const uint8_t* mask, double* sums,
                          int B, int HW, dv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFUVOLUME_HIP_H */
