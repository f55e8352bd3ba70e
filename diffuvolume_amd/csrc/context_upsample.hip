// IGEV's convex upsampling of the quarter-resolution disparity (KITTI15/core/igev_stereo_ddim.py:209-217
// `upsample_disp`, core/submodule.py:241-253 `context_upsample`):
//   p      = softmax over the 9 taps of the superpixel logits [B,9,4h,4w]          (F.softmax(spx_pred, 1))
//   out[Y,X] = sum_k p_k[Y,X] * scale * disp[(Y>>2) + ky - 1, (X>>2) + kx - 1]      (unfold 3x3 pad 1, nearest x4)
// with tap k = 3*ky + kx (F.unfold's channel order) and zeros outside the image.  The reference materialises the
// unfolded [B,9,h,w] tensor, its nearest-neighbour x4 copy [B,9,4h,4w], the softmax and the product; here each
// thread owns 4 consecutive X of one output row (they share one low-resolution cell): 9 x 16-byte logit loads, one
// 3x3 neighbourhood of the disparity, one 16-byte store.  HBM-bound: 40 B/output pixel.
#include "dv_common.h"

namespace {

template <bool SOFTMAX>
__global__ __launch_bounds__(256) void context_upsample_kernel(const float* __restrict__ disp,
                                                               const float* __restrict__ w9,
                                                               float* __restrict__ out, float scale, int h, int w,
                                                               size_t cells) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one (b, Y, x) cell = 4 output pixels
  if (i >= cells) return;
  const int x = (int)(i % w);
  const int Y = (int)((i / w) % (4 * h));
  const size_t b = i / ((size_t)w * 4 * h);
  const int y = Y >> 2;
  const float* dp = disp + b * (size_t)h * w;
  float nb[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xx = x + kx - 1;
      nb[ky * 3 + kx] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? dp[(size_t)yy * w + xx] * scale : 0.f;
    }
  const size_t plane = (size_t)16 * h * w;
  const size_t o = (size_t)Y * 4 * w + 4 * x;
  const float* lp = w9 + b * 9 * plane + o;
  float4 l[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) l[k] = *reinterpret_cast<const float4*>(lp + k * plane);
  float r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = j == 0 ? l[k].x : j == 1 ? l[k].y : j == 2 ? l[k].z : l[k].w;
    if (SOFTMAX) {
      float m = v[0];
#pragma unroll
      for (int k = 1; k < 9; ++k) m = fmaxf(m, v[k]);
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        v[k] = expf(v[k] - m);
        s += v[k];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = v[k] / s;
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc += nb[k] * v[k];
    r[j] = acc;
  }
  *reinterpret_cast<float4*>(out + b * plane + o) = make_float4(r[0], r[1], r[2], r[3]);
}

}  // namespace

extern "C" int dv_context_upsample_f32(const float* disp_low, const float* weights, float* out, int B, int h, int w,
                                       float scale, int apply_softmax, dv_stream_t stream) {
  DV_REQUIRE_PTR(disp_low);
  DV_REQUIRE_PTR(weights);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && h > 0 && w > 0, DV_ERR_SHAPE);
  DV_REQUIRE(dv_aligned16(weights) && dv_aligned16(out), DV_ERR_ALIGN);      // rows are 4*w floats: always 16-B multiples
  const size_t cells = (size_t)B * 4 * h * w;
  const unsigned nblk = (unsigned)((cells + 255) / 256);
  if (apply_softmax)
    hipLaunchKernelGGL((context_upsample_kernel<true>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, disp_low,
                       weights, out, scale, h, w, cells);
  else
    hipLaunchKernelGGL((context_upsample_kernel<false>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, disp_low,
                       weights, out, scale, h, w, cells);
  return dv_launch_status();
}
