// K7: trilinear x4 upsample -> softmax over disparity -> soft-argmax (+ uncertainty), fused.
// Replaces F.upsample(..., 'trilinear') + F.softmax(dim=1) + disparity_regression
// (SceneFlow/models/acv_ddim.py:267-270, submodule.py:173-177) and the uncertainty block
// acv_ddim.py:325-329.  The reference materialises three [B,192,H,W] tensors per call
// (377 MB each per pair); here each output pixel keeps its D in-plane interpolated
// costs in registers and walks the 4D bins four times (max, sum, expectation, |.| moment),
// so HBM traffic is the [B,D,h,w] cost in and two [B,4h,4w] maps out.
#include "dv_common.h"

namespace {

// PyTorch's linear source index (area_pixel_compute_source_index, non-cubic).
template <bool ALIGN>
__device__ __forceinline__ void lin_src(int dst, int in_size, int out_size, int& i0, int& i1, float& l0,
                                        float& l1) {
  float src;
  if (ALIGN) {
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    src = scale * (float)dst;
  } else {
    const float scale = (float)in_size / (float)out_size;
    src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
  }
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

template <int D, bool ALIGN>
__global__ __launch_bounds__(256) void upsample_softmax_regress_kernel(const float* __restrict__ cost,
                                                                       const float* __restrict__ disp_in,
                                                                       float* __restrict__ disp,
                                                                       float* __restrict__ unc, int h,
                                                                       int w, size_t total, unsigned cost_bytes) {
  const int H = 4 * h, W = 4 * w;
  constexpr int K = 4 * D;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int X = (int)(i % W);
  const int Y = (int)((i / W) % H);
  const int b = (int)(i / ((size_t)W * H));
  int y0, y1, x0, x1;
  float hy0, hy1, wx0, wx1;
  lin_src<ALIGN>(Y, h, H, y0, y1, hy0, hy1);
  lin_src<ALIGN>(X, w, W, x0, x1, wx0, wx1);
  const size_t plane = (size_t)h * w;
  // the 4 x D neighbour loads as buffer loads: four per-lane 32-bit byte offsets (slice d = 0 of batch item b) plus a
  // SCALAR offset d * plane -- no per-load 64-bit address arithmetic on the vector unit (it was 370 of the kernel's
  // 3 300 vector instructions per pixel).  The host guarantees the cost tensor is below 2^31 bytes.
  const int plane_bytes = __builtin_amdgcn_readfirstlane((int)(plane * sizeof(float)));
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cost), 0, (int)cost_bytes, 0x00020000);
  const unsigned ob = (unsigned)b * D * (unsigned)plane;
  const int o00 = (int)((ob + y0 * w + x0) * 4u), o01 = (int)((ob + y0 * w + x1) * 4u);
  const int o10 = (int)((ob + y1 * w + x0) * 4u), o11 = (int)((ob + y1 * w + x1) * 4u);
  float c[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int so = d * plane_bytes;
    const float v00 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o00, so, 0));
    const float v01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o01, so, 0));
    const float v10 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o10, so, 0));
    const float v11 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, o11, so, 0));
    c[d] = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
  }
  // v_k = l0*c[i0] + l1*c[i1]; k is a compile-time constant after unrolling, so the
  // index / weight computation folds away and c[] stays in registers.
#define DV_VK(k, out)                                  \
  {                                                    \
    int i0_, i1_;                                      \
    float l0_, l1_;                                    \
    lin_src<ALIGN>((k), D, K, i0_, i1_, l0_, l1_);     \
    (out) = l0_ * c[i0_] + l1_ * c[i1_];               \
  }
  // Softmax shift = max_k v_k.  Between two neighbouring slices the four bins are a linear ramp, so the maximum sits on
  // the outer bins of a segment (align_corners=False: k = 4d + 1 and 4d + 2; k = 0 and 4D - 1 are the clamped ends):
  // half the bins are enough.
  // (In floating point an inner bin can exceed that by an ulp when the two slices are nearly equal: its exponential is
  // then 1 + O(ulp), harmless.  The maximum of the 48 SLICES is not a valid shift: bins never reach a slice value, and
  // with steep costs every exponential would underflow.)
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (ALIGN || (k & 3) == 1 || (k & 3) == 2 || k == 0 || k == K - 1) {     // (align_corners=True: bins are not periodic)
      float v;
      DV_VK(k, v);
      m = fmaxf(m, v);
    }
  }
  // exp(v_k - m) is evaluated once and kept (4D values per pixel: the kernel runs one wave per SIMD on the
  // whole 512-entry register file); the later passes only divide and accumulate, in the reference's order
  // (softmax first, then the expectation / the absolute moment).
  float e[K];
  float s = 0.f, num = 0.f;               // sum_k e_k and sum_k e_k k in the same pass
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float v;
    DV_VK(k, v);
    e[k] = dv_exp_le0(v - m);
    s += e[k];
    num += e[k] * (float)k;
  }
  // p_k = e_k * (1/s): one correctly rounded division per pixel instead of 4D (each is ~10 instructions);
  // differs from e_k / s by at most one ulp of p_k
  // The normalisation is applied to the two sums, not to the 4D exponentials (sum_k (e_k / s) k = (sum_k e_k k) / s up
  // to rounding): one reciprocal and two multiplications per pixel instead of 4D.
  const float rs = 1.f / s;
  float dsp = 0.f;
  if (disp_in) {           // uncertainty about an externally refined disparity (pwcnet_ddim.py:548-552)
    dsp = disp_in[i];
  } else {
    dsp = num * rs;
    disp[i] = dsp;
  }
  if (unc) {
    float u = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) u += fabsf(dsp - (float)k) * e[k];
    unc[i] = u * rs;
  }
#undef DV_VK
}

// Any D: the D in-plane costs of a pixel live in LDS (one column per thread).
template <bool ALIGN>
__global__ void upsample_softmax_regress_generic(const float* __restrict__ cost,
                                                 const float* __restrict__ disp_in, float* __restrict__ disp,
                                                 float* __restrict__ unc, int D, int h, int w,
                                                 size_t total) {
  extern __shared__ float cs[];  // [D][blockDim.x]
  const int H = 4 * h, W = 4 * w, K = 4 * D;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < total;
  const size_t ii = live ? i : 0;
  const int X = (int)(ii % W);
  const int Y = (int)((ii / W) % H);
  const int b = (int)(ii / ((size_t)W * H));
  int y0, y1, x0, x1;
  float hy0, hy1, wx0, wx1;
  lin_src<ALIGN>(Y, h, H, y0, y1, hy0, hy1);
  lin_src<ALIGN>(X, w, W, x0, x1, wx0, wx1);
  const size_t plane = (size_t)h * w;
  const float* cb = cost + (size_t)b * D * plane;
  float* c = cs + threadIdx.x;
  const int st = blockDim.x;
  for (int d = 0; d < D; ++d) {
    const float* p = cb + (size_t)d * plane;
    c[d * st] = hy0 * (wx0 * p[(size_t)y0 * w + x0] + wx1 * p[(size_t)y0 * w + x1]) +
                hy1 * (wx0 * p[(size_t)y1 * w + x0] + wx1 * p[(size_t)y1 * w + x1]);
  }
  auto vk = [&](int k) {
    int i0, i1;
    float l0, l1;
    lin_src<ALIGN>(k, D, K, i0, i1, l0, l1);
    return l0 * c[i0 * st] + l1 * c[i1 * st];
  };
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmaxf(m, vk(k));
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += expf(vk(k) - m);
  float dsp = 0.f;
  if (!disp_in)
    for (int k = 0; k < K; ++k) dsp += (expf(vk(k) - m) / s) * (float)k;
  if (!live) return;
  if (disp_in) dsp = disp_in[i];
  else disp[i] = dsp;
  if (unc) {
    float u = 0.f;
    for (int k = 0; k < K; ++k) u += fabsf(dsp - (float)k) * (expf(vk(k) - m) / s);
    unc[i] = u;
  }
}

__global__ void disparity_regression_kernel(const float* __restrict__ prob, float* __restrict__ disp,
                                            int D, size_t plane, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t b = i / plane, p = i - b * plane;
  const float* src = prob + b * D * plane + p;
  float acc = 0.f;
  for (int d = 0; d < D; ++d) acc += src[(size_t)d * plane] * (float)d;
  disp[i] = acc;
}

// F.softmax(cost, dim=1) + disparity_regression with no upsampling (IGEV init_disp,
// igev_stereo_ddim.py:382-383): one thread per pixel, three passes over its D logits (L2-resident).
__global__ void softmax_regress_kernel(const float* __restrict__ cost, float* __restrict__ disp, int D,
                                       size_t plane, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t b = i / plane, p = i - b * plane;
  const float* src = cost + b * D * plane + p;
  float mx = src[0];
  for (int d = 1; d < D; ++d) mx = fmaxf(mx, src[(size_t)d * plane]);
  float sum = 0.f;
  for (int d = 0; d < D; ++d) sum += expf(src[(size_t)d * plane] - mx);
  float acc = 0.f;
  for (int d = 0; d < D; ++d) acc += (expf(src[(size_t)d * plane] - mx) / sum) * (float)d;
  disp[i] = acc;
}

}  // namespace

extern "C" int dv_softmax_regress_f32(const float* cost, float* disp, int B, int D, int H, int W,
                                      dv_stream_t stream) {
  DV_REQUIRE_PTR(cost);
  DV_REQUIRE_PTR(disp);
  DV_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, DV_ERR_SHAPE);
  const size_t plane = (size_t)H * W, total = plane * B;
  hipLaunchKernelGGL(softmax_regress_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, cost, disp, D, plane, total);
  return dv_launch_status();
}

static int launch_tail(const float* cost, const float* disp_in, float* disp, float* unc, int B, int D, int h,
                       int w, int align_corners, hipStream_t s) {
  const size_t total = (size_t)B * 16 * h * w;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  const size_t cost_bytes = (size_t)B * D * h * w * sizeof(float);
  if (D == 48 && cost_bytes < 0x80000000ull) {          // 32-bit buffer offsets; larger volumes take the generic kernel
    if (align_corners)
      hipLaunchKernelGGL((upsample_softmax_regress_kernel<48, true>), dim3(blocks), dim3(256), 0, s, cost,
                         disp_in, disp, unc, h, w, total, (unsigned)cost_bytes);
    else
      hipLaunchKernelGGL((upsample_softmax_regress_kernel<48, false>), dim3(blocks), dim3(256), 0, s, cost,
                         disp_in, disp, unc, h, w, total, (unsigned)cost_bytes);
    return dv_launch_status();
  }
  const int threads = 64;
  const size_t lds = (size_t)D * threads * sizeof(float);
  DV_REQUIRE(lds <= 64 * 1024, DV_ERR_UNSUPPORTED);
  const unsigned gblocks = (unsigned)((total + threads - 1) / threads);
  if (align_corners)
    hipLaunchKernelGGL((upsample_softmax_regress_generic<true>), dim3(gblocks), dim3(threads), lds, s, cost,
                       disp_in, disp, unc, D, h, w, total);
  else
    hipLaunchKernelGGL((upsample_softmax_regress_generic<false>), dim3(gblocks), dim3(threads), lds, s, cost,
                       disp_in, disp, unc, D, h, w, total);
  return dv_launch_status();
}

extern "C" int dv_upsample_softmax_regress_f32(const float* cost, float* disp, float* unc, int B, int D,
                                               int h, int w, int align_corners, dv_stream_t stream) {
  DV_REQUIRE_PTR(cost);
  DV_REQUIRE_PTR(disp);
  DV_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, DV_ERR_SHAPE);
  return launch_tail(cost, nullptr, disp, unc, B, D, h, w, align_corners, (hipStream_t)stream);
}

extern "C" int dv_upsample_softmax_uncertainty_f32(const float* cost, const float* disp, float* unc, int B,
                                                   int D, int h, int w, int align_corners,
                                                   dv_stream_t stream) {
  DV_REQUIRE_PTR(cost);
  DV_REQUIRE_PTR(disp);
  DV_REQUIRE_PTR(unc);
  DV_REQUIRE(B > 0 && D > 0 && h > 0 && w > 0, DV_ERR_SHAPE);
  return launch_tail(cost, disp, nullptr, unc, B, D, h, w, align_corners, (hipStream_t)stream);
}

extern "C" int dv_disparity_regression_f32(const float* prob, float* disp, int B, int D, int H, int W,
                                           dv_stream_t stream) {
  DV_REQUIRE_PTR(prob);
  DV_REQUIRE_PTR(disp);
  DV_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, DV_ERR_SHAPE);
  const size_t plane = (size_t)H * W, total = (size_t)B * plane;
  hipLaunchKernelGGL(disparity_regression_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, prob, disp, D, plane, total);
  return dv_launch_status();
}
