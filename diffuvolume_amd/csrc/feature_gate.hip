// FeatureAtt's volume gate (KITTI15/core/submodule.py:234-239): cv[b,c,d,y,x] *= sigmoid(logit[b,c,y,x]).
// The 2-D logits come from the image-feature branch (two 1x1 Conv2d, PyTorch side); the broadcast over
// the disparity axis is the HBM-bound part: one read + one write of the volume, the [C,H,W] logit plane
// stays in L2 across the D slices a block walks.
#include "dv_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// grid: (plane chunks, D split, B*C); each thread owns VEC consecutive x of one (y,x) position and walks d.
template <int VEC>
__global__ __launch_bounds__(256) void feature_gate_kernel(const float* __restrict__ cv,
                                                           const float* __restrict__ logit,
                                                           float* __restrict__ out, int D, int HW) {
  const int bc = blockIdx.z;
  const int i = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (i >= HW) return;
  const float* lp = logit + (size_t)bc * HW + i;
  float g[VEC];
  if (VEC == 4) {
    const float4 l = *reinterpret_cast<const float4*>(lp);
    g[0] = sigmoidf_(l.x); g[1 % VEC] = sigmoidf_(l.y); g[2 % VEC] = sigmoidf_(l.z); g[3 % VEC] = sigmoidf_(l.w);
  } else {
    g[0] = sigmoidf_(lp[0]);
  }
  const size_t base = (size_t)bc * D * HW + i;
  for (int d = blockIdx.y; d < D; d += gridDim.y) {
    const size_t o = base + (size_t)d * HW;
    if (VEC == 4) {
      float4 v = *reinterpret_cast<const float4*>(cv + o);
      v.x *= g[0]; v.y *= g[1 % VEC]; v.z *= g[2 % VEC]; v.w *= g[3 % VEC];
      *reinterpret_cast<float4*>(out + o) = v;
    } else {
      out[o] = cv[o] * g[0];
    }
  }
}

}  // namespace

extern "C" int dv_feature_gate_f32(const float* cv, const float* logit, float* out, int B, int C, int D,
                                   int H, int W, dv_stream_t stream) {
  DV_REQUIRE_PTR(cv);
  DV_REQUIRE_PTR(logit);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, DV_ERR_SHAPE);
  DV_REQUIRE((long long)B * C <= 65535, DV_ERR_SHAPE);
  const int HW = H * W;
  const bool vec = (HW % 4 == 0) && dv_aligned16(cv) && dv_aligned16(out) && dv_aligned16(logit);
  const int per = vec ? 1024 : 256;
  const int gx = (HW + per - 1) / per;
  int gy = 1;                                   // split D only when the plane alone cannot fill the chip
  while (gy < D && (long long)gx * gy * B * C < 2048) gy *= 2;
  if (gy > D) gy = D;
  const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)(B * C));
  if (vec)
    hipLaunchKernelGGL((feature_gate_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, cv, logit, out, D, HW);
  else
    hipLaunchKernelGGL((feature_gate_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, cv, logit, out, D, HW);
  return dv_launch_status();
}
