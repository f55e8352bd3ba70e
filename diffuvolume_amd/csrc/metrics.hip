// K9: masked per-image error sums for EPE / D1 / Thres{1,2,3}
// (SceneFlow/utils/metrics.py:22-65).  The reference gathers with boolean indexing and
// syncs the device once per metric per image (.item()); here one pass produces the seven
// sums per image and the caller (and the RCCL all-reduce across ranks) works on 64 bytes.
#include "dv_common.h"

namespace {

constexpr int kNS = 8;

__global__ __launch_bounds__(256) void masked_metrics_kernel(const float* __restrict__ est,
                                                             const float* __restrict__ gt,
                                                             const uint8_t* __restrict__ mask,
                                                             double* __restrict__ sums, int HW) {
  __shared__ double red[4][kNS];
  const int b = blockIdx.y;
  const size_t base = (size_t)b * HW;
  double s[kNS] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    const float g = gt[base + i];
    if (g > 0.f) s[1] += 1.0;
    if (mask[base + i]) {
      const float e = fabsf(g - est[base + i]);
      s[0] += 1.0;
      s[2] += (double)e;
      if (e > 3.f && e / fabsf(g) > 0.05f) s[3] += 1.0;
      if (e > 1.f) s[4] += 1.0;
      if (e > 2.f) s[5] += 1.0;
      if (e > 3.f) s[6] += 1.0;
    }
  }
#pragma unroll
  for (int k = 0; k < kNS; ++k)
    for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off, DV_WAVE);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int k = 0; k < kNS; ++k) red[wave][k] = s[k];
  __syncthreads();
  if (threadIdx.x < kNS - 1) {
    const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (v != 0.0) atomicAdd(&sums[(size_t)b * kNS + threadIdx.x], v);
  }
}

}  // namespace

extern "C" int dv_masked_metrics_f32(const float* est, const float* gt, const uint8_t* mask, double* sums,
                                     int B, int HW, dv_stream_t stream) {
  DV_REQUIRE_PTR(est);
  DV_REQUIRE_PTR(gt);
  DV_REQUIRE_PTR(mask);
  DV_REQUIRE_PTR(sums);
  DV_REQUIRE(B > 0 && HW > 0, DV_ERR_SHAPE);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(sums, 0, (size_t)B * kNS * sizeof(double), s);
  if (e != hipSuccess) return (int)e;
  int bx = (HW + 256 * 8 - 1) / (256 * 8);
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(masked_metrics_kernel, dim3(bx, B), dim3(256), 0, s, est, gt, mask, sums, HW);
  return dv_launch_status();
}
