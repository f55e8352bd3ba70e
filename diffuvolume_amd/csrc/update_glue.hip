// K12: the small per-iteration operators around IGEV's ConvGRUs (KITTI15/core/update.py) that are not convolutions
// of the MFMA kind.  They run 32 x steps times per pair; on PyTorch they were a generic bilinear kernel (0.2 ms per
// call at batch 4) and a naive MIOpen solver for the single-input-channel 7x7 (0.7 ms): 23 % of an iteration.
//   dv_conv2d_1in_f32         nn.Conv2d(1, Cout, K, padding=K/2) + bias + activation   (BasicMotionEncoder.convd1, :86,:92)
//   dv_resize_bilinear_ac_f32 F.interpolate(x, size, mode='bilinear', align_corners=True)   (`interp`, update.py:100-102)
//   dv_avg_pool3s2_f32        F.avg_pool2d(x, 3, stride=2, padding=1)  (count_include_pad)    (`pool2x`, update.py:96-97)
#include "dv_common.h"

namespace {

// ---- one input channel, K x K taps (K odd <= 7), Cout channels: VALU.  Block = 16 x 16 pixels; the haloed input
// tile and the whole weight set sit in LDS; a thread keeps its K*K window in registers and walks the output channels
// with broadcast reads of the weights (every lane reads the same address) ----
template <int K>
__global__ __launch_bounds__(256) void conv2d_1in_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int H, int W, int Cout, int act, int ntx, int nty) {
  constexpr int P = K / 2, T = 16, IT = T + 2 * P;
  extern __shared__ float sm[];
  float* in_s = sm;                 // [IT][IT]
  float* w_s = sm + IT * IT;        // [Cout][K*K]
  const int tid = threadIdx.x;
  unsigned t = blockIdx.x;
  const int tx = t % ntx; t /= ntx;
  const int ty = t % nty;
  const int b = t / nty;
  const int x0 = tx * T, y0 = ty * T;
  const float* ib = in + (size_t)b * H * W;
  for (int i = tid; i < IT * IT; i += 256) {
    const int yy = i / IT, xx = i - yy * IT;
    const int y = y0 - P + yy, x = x0 - P + xx;
    in_s[i] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? ib[(size_t)y * W + x] : 0.f;
  }
  for (int i = tid; i < Cout * K * K; i += 256) w_s[i] = w[i];
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  float win[K * K];
#pragma unroll
  for (int dy = 0; dy < K; ++dy)
#pragma unroll
    for (int dx = 0; dx < K; ++dx) win[dy * K + dx] = in_s[(ly + dy) * IT + lx + dx];
  const int y = y0 + ly, x = x0 + lx;
  if (y >= H || x >= W) return;
  float* ob = out + ((size_t)b * Cout * H + y) * W + x;
  for (int co = 0; co < Cout; ++co) {
    const float* wc = w_s + co * K * K;
    float acc = bias ? bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc = fmaf(win[i], wc[i], acc);
    ob[(size_t)co * H * W] = dv_act(acc, act);
  }
}

// PyTorch's align_corners=True source index: src = dst * (in - 1) / (out - 1)  (0 when out == 1)
__device__ __forceinline__ float resize_ac_at(const float* __restrict__ p, int h, int w, float sy, float sx, int Y, int X) {
  const float fy = sy * (float)Y, fx = sx * (float)X;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  return hy * (hx * p[(size_t)y0 * w + x0] + lx * p[(size_t)y0 * w + x1]) +
         ly * (hx * p[(size_t)y1 * w + x0] + lx * p[(size_t)y1 * w + x1]);
}

__global__ void resize_bilinear_ac_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, int H,
                                          int W, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int X = (int)(i % W);
  const int Y = (int)((i / W) % H);
  const size_t bc = i / ((size_t)W * H);
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  out[i] = resize_ac_at(in + bc * (size_t)h * w, h, w, sy, sx, Y, X);
}

// the same, four consecutive outputs of a row per thread (W % 4 == 0, 16-byte aligned rows): the kernel is its stores --
// `interp` writes the 1/4-resolution hidden state once per GRU iteration (61 MB at batch 4)
__global__ void resize_bilinear_ac_x4_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, int H,
                                             int W, size_t total4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int W4 = W >> 2;
  const int X = (int)(i % W4) * 4;
  const int Y = (int)((i / W4) % H);
  const size_t bc = i / ((size_t)W4 * H);
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const float* p = in + bc * (size_t)h * w;
  float4 v;
  v.x = resize_ac_at(p, h, w, sy, sx, Y, X);
  v.y = resize_ac_at(p, h, w, sy, sx, Y, X + 1);
  v.z = resize_ac_at(p, h, w, sy, sx, Y, X + 2);
  v.w = resize_ac_at(p, h, w, sy, sx, Y, X + 3);
  *reinterpret_cast<float4*>(out + (bc * H + Y) * (size_t)W + X) = v;
}

// 3x3 average, stride 2, padding 1, padded zeros counted (divisor 9): out = floor((in + 2 - 3) / 2) + 1
__global__ void avg_pool3s2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int Ho, int Wo,
                                   size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int X = (int)(i % Wo);
  const int Y = (int)((i / Wo) % Ho);
  const size_t bc = i / ((size_t)Wo * Ho);
  const float* p = in + bc * (size_t)H * W;
  float s = 0.f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int y = 2 * Y - 1 + dy;
    if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int x = 2 * X - 1 + dx;
      if ((unsigned)x < (unsigned)W) s += p[(size_t)y * W + x];
    }
  }
  out[i] = s / 9.0f;
}

}  // namespace

extern "C" int dv_conv2d_1in_f32(const float* in, const float* w, const float* bias, float* out, int B, int H, int W,
                                 int Cout, int k, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(k == 3 || k == 5 || k == 7, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_TANH, DV_ERR_UNSUPPORTED);
  const int it = 16 + 2 * (k / 2);
  const size_t lds = ((size_t)it * it + (size_t)Cout * k * k) * sizeof(float);
  DV_REQUIRE(lds <= 64 * 1024, DV_ERR_UNSUPPORTED);
  const int ntx = (W + 15) / 16, nty = (H + 15) / 16;
  const long long blocks = (long long)B * nty * ntx;
  DV_REQUIRE(blocks <= 0x7fffffffLL, DV_ERR_SHAPE);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)blocks), block(256);
  if (k == 3) hipLaunchKernelGGL(conv2d_1in_kernel<3>, grid, block, lds, s, in, w, bias, out, H, W, Cout, act, ntx, nty);
  else if (k == 5) hipLaunchKernelGGL(conv2d_1in_kernel<5>, grid, block, lds, s, in, w, bias, out, H, W, Cout, act, ntx, nty);
  else hipLaunchKernelGGL(conv2d_1in_kernel<7>, grid, block, lds, s, in, w, bias, out, H, W, Cout, act, ntx, nty);
  return dv_launch_status();
}

extern "C" int dv_resize_bilinear_ac_f32(const float* in, float* out, int BC, int h, int w, int H, int W,
                                         dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(BC > 0 && h > 0 && w > 0 && H > 0 && W > 0, DV_ERR_SHAPE);
  const size_t total = (size_t)BC * H * W;
  DV_REQUIRE((total + 255) / 256 <= 0x7fffffffull, DV_ERR_SHAPE);
  if (W % 4 == 0 && dv_aligned16(out)) {
    const size_t total4 = total / 4;
    hipLaunchKernelGGL(resize_bilinear_ac_x4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, in, out, h, w, H, W, total4);
    return dv_launch_status();
  }
  hipLaunchKernelGGL(resize_bilinear_ac_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     in, out, h, w, H, W, total);
  return dv_launch_status();
}

extern "C" int dv_avg_pool3s2_f32(const float* in, float* out, int BC, int H, int W, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(BC > 0 && H > 0 && W > 0, DV_ERR_SHAPE);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t total = (size_t)BC * Ho * Wo;
  DV_REQUIRE((total + 255) / 256 <= 0x7fffffffull, DV_ERR_SHAPE);
  hipLaunchKernelGGL(avg_pool3s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, H, W, Ho, Wo, total);
  return dv_launch_status();
}
