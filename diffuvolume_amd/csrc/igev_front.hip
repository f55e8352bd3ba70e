// K13: the two operators IGEV's once-per-pair 2-D front needs beside the convolution kernels of conv2d*.hip, so that it
// runs without MIOpen (KITTI15/core/extractor.py:190-295 `MultiBasicEncoder`, core/submodule.py:79-107 `BasicConv_IN`,
// igev_stereo_ddim.py:100-117 stems / spx heads):
//   dv_conv2d_fewin_f32      nn.Conv2d(Cin <= 4, Cout, k in {3,5,7}, stride in {1,2}, padding=k/2) [+ bias]
//                            [+ per-channel scale / shift = folded eval BatchNorm] + activation: the 7x7 stride-2 stem of the
//                            context encoder (extractor.py:197) and the RGB stems.  VALU: three input channels would
//                            waste 1/4 .. 3/4 of every MFMA k-step, and the layer runs once per pair (0.02 TFLOP at 384x1248).
//   dv_instance_norm_act_f32 nn.InstanceNorm2d(C) (affine=False, eps) + activation, one block per (b, c) plane: mean and
//                            biased variance in two passes over the plane (fp32 sums in a fixed order: bit-reproducible and
//                            independent of the batch size, which MIOpen's batch-norm kernels are not bound to be).
#include "dv_common.h"

namespace {

// Block = 16 x 16 output pixels; the haloed input tile of ONE input channel at a time and the whole weight set sit in LDS;
// a thread keeps its K*K window in registers and walks the output channels (accumulators in registers, NCO at a time) with
// broadcast reads of the weights.
template <int K, int S, int NCO>
__global__ __launch_bounds__(256) void conv2d_fewin_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ ch_scale,
                                                           const float* __restrict__ ch_shift, float* __restrict__ out,
                                                           int Cin, int H, int W, int Cout, int Ho, int Wo, int act,
                                                           int ntx, int nty) {
  constexpr int P = K / 2, T = 16, IT = (T - 1) * S + K;
  extern __shared__ float sm[];
  float* in_s = sm;                 // [IT][IT]
  float* w_s = sm + IT * IT;        // [Cout][Cin][K*K]
  const int tid = threadIdx.x;
  unsigned t = blockIdx.x;
  const int tx = t % ntx; t /= ntx;
  const int ty = t % nty;
  const int b = t / nty;
  const int x0 = tx * T, y0 = ty * T;
  for (int i = tid; i < Cout * Cin * K * K; i += 256) w_s[i] = w[i];
  const int ly = tid >> 4, lx = tid & 15;
  const int y = y0 + ly, x = x0 + lx;
  const bool live = y < Ho && x < Wo;
  for (int c0 = 0; c0 < Cout; c0 += NCO) {
    float acc[NCO];
#pragma unroll
    for (int n = 0; n < NCO; ++n) acc[n] = 0.f;
    for (int c = 0; c < Cin; ++c) {
      const float* ib = in + ((size_t)b * Cin + c) * H * W;
      __syncthreads();              // the previous channel's window reads are done (and w_s is complete the first time)
      for (int i = tid; i < IT * IT; i += 256) {
        const int yy = i / IT, xx = i - yy * IT;
        const int yi = y0 * S - P + yy, xi = x0 * S - P + xx;
        in_s[i] = ((unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W) ? ib[(size_t)yi * W + xi] : 0.f;
      }
      __syncthreads();
      float win[K * K];
#pragma unroll
      for (int dy = 0; dy < K; ++dy)
#pragma unroll
        for (int dx = 0; dx < K; ++dx) win[dy * K + dx] = in_s[(ly * S + dy) * IT + lx * S + dx];
#pragma unroll
      for (int n = 0; n < NCO; ++n) {
        if (c0 + n < Cout) {
          const float* wc = w_s + ((size_t)(c0 + n) * Cin + c) * K * K;
#pragma unroll
          for (int i = 0; i < K * K; ++i) acc[n] = fmaf(win[i], wc[i], acc[n]);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int n = 0; n < NCO; ++n) {
        const int co = c0 + n;
        if (co < Cout) {
          float v = acc[n] + (bias ? bias[co] : 0.f);
          if (ch_scale) v = fmaf(v, ch_scale[co], ch_shift[co]);
          out[(((size_t)b * Cout + co) * Ho + y) * Wo + x] = dv_act(v, act);
        }
      }
    }
  }
}

// One block of 1024 threads per (b, c) plane.  Pass 1: sum -> mean; pass 2: sum of squared deviations -> biased variance
// (the two-pass form: no cancellation); pass 3: normalise + activation.  Per-thread partial sums run over a fixed stride
// and are combined by a fixed tree, so the result depends on nothing but the plane.
// (`in` and `out` may be the same tensor -- the wrapper's default is in place --, so neither is __restrict__)
__global__ __launch_bounds__(1024) void instance_norm_act_kernel(const float* in, float* out, int HW, float eps, int act) {
  __shared__ float red[16];
  __shared__ float stat;
  const size_t base = (size_t)blockIdx.x * HW;
  const float* p = in + base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  auto block_sum = [&](float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();                                 // `red` / `stat` of the previous reduction have been read
    if (lane == 0) red[wv] = v;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int i = 0; i < 16; ++i) s += red[i];
      stat = s;
    }
    __syncthreads();
    return stat;
  };
  float s = 0.f;
  for (int i = tid; i < HW; i += 1024) s += p[i];
  const float mean = block_sum(s) / (float)HW;
  float q = 0.f;
  for (int i = tid; i < HW; i += 1024) {
    const float d = p[i] - mean;
    q = fmaf(d, d, q);
  }
  const float rstd = 1.0f / sqrtf(block_sum(q) / (float)HW + eps);
  for (int i = tid; i < HW; i += 1024) out[base + i] = dv_act((p[i] - mean) * rstd, act);
}

}  // namespace

extern "C" int dv_conv2d_fewin_f32(const float* in, const float* w, const float* bias, const float* ch_scale,
                                   const float* ch_shift, float* out, int B, int Cin, int H, int W, int Cout, int k,
                                   int stride, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && Cin > 0 && Cin <= 4 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(k == 3 || k == 5 || k == 7, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(stride == 1 || stride == 2, DV_ERR_UNSUPPORTED);
  DV_REQUIRE((ch_scale == nullptr) == (ch_shift == nullptr), DV_ERR_NULL);
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_TANH, DV_ERR_UNSUPPORTED);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;          // padding k/2
  const int it = 15 * stride + k;
  const size_t lds = ((size_t)it * it + (size_t)Cout * Cin * k * k) * sizeof(float);
  DV_REQUIRE(lds <= 64 * 1024, DV_ERR_UNSUPPORTED);
  const int ntx = (Wo + 15) / 16, nty = (Ho + 15) / 16;
  const long long blocks = (long long)B * nty * ntx;
  DV_REQUIRE(blocks <= 0x7fffffffLL, DV_ERR_SHAPE);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)blocks), block(256);
#define DV_FEWIN(K, S)                                                                                                   \
  hipLaunchKernelGGL((conv2d_fewin_kernel<K, S, 16>), grid, block, lds, s, in, w, bias, ch_scale, ch_shift, out, Cin, H, \
                     W, Cout, Ho, Wo, act, ntx, nty)
  if (k == 3 && stride == 1) DV_FEWIN(3, 1);
  else if (k == 3) DV_FEWIN(3, 2);
  else if (k == 5 && stride == 1) DV_FEWIN(5, 1);
  else if (k == 5) DV_FEWIN(5, 2);
  else if (stride == 1) DV_FEWIN(7, 1);
  else DV_FEWIN(7, 2);
#undef DV_FEWIN
  return dv_launch_status();
}

extern "C" int dv_instance_norm_act_f32(const float* in, float* out, int BC, int HW, float eps, int act,
                                        dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(BC > 0 && HW > 0, DV_ERR_SHAPE);
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_TANH, DV_ERR_UNSUPPORTED);
  hipLaunchKernelGGL(instance_norm_act_kernel, dim3((unsigned)BC), dim3(1024), 0, (hipStream_t)stream, in, out, HW, eps, act);
  return dv_launch_status();
}
