// K3t: the per-pair tables of the rank-1 first layer (rank1_filter.hip):  GL[tap * Cout + co](y, x) = sum_c W[co, c, tap] L[c, y, x]
// -- a 1x1 convolution from FEW input channels (32 concat features, SceneFlow/models/acv_ddim.py:388) to MANY output
// channels (27 * 32 = 864), no BatchNorm, no activation.  It writes 0.85 GB per launch at batch 8 and multiplies almost
// nothing (13.6 GFLOP), so it is a pure HBM write stream; the generic 2-D convolution kernel spent 2.15 ms on it
// (0.4 TB/s: 27 blocks per pixel tile, each staging the same 32 channels through LDS for one 8-step K loop).
//
// Here a block owns 64 consecutive pixels of one batch item and ALL output channels: the A fragments of its four
// 16-pixel M-tiles (8 k-steps x 4 tiles = 32 registers per lane) are loaded once, straight from global memory; each
// wave then walks its share of the 16-channel N-tiles -- two 16-byte loads of packed weights (L1 / L2 resident: 110 KB
// for the whole layer), 32 MFMAs (v_mfma_f32_16x16x4_f32), four 16-byte stores -- with no LDS and no barrier.
#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PW_KS = 8;          // k-steps of 4 input channels: Cin <= 32
constexpr int PW_MT = 4;          // 16-pixel M-tiles per block

// packed weights: [ntile][kq 4][j 16][ks 8]  =  W[ntile * 16 + j][ks * 4 + kq]  (zero beyond Cout / Cin)
__global__ void pw_pack_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout, int ntiles) {
  const int total = ntiles * 4 * 16 * PW_KS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int r = i;
    const int ks = r % PW_KS; r /= PW_KS;
    const int j = r % 16; r /= 16;
    const int kq = r % 4;
    const int nt = r / 4;
    const int co = nt * 16 + j, ci = ks * 4 + kq;
    wpk[i] = (co < Cout && ci < Cin) ? w[(size_t)co * Cin + ci] : 0.f;
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void pw_expand_kernel(const float* __restrict__ in, const float* __restrict__ wpk,
                                                        float* __restrict__ out, int Cin, int Cout, int HW, int ntiles) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * (16 * PW_MT);
  const float* inb = in + (size_t)b * Cin * HW;
  float a[PW_MT][PW_KS];
#pragma unroll
  for (int m = 0; m < PW_MT; ++m)
#pragma unroll
    for (int ks = 0; ks < PW_KS; ++ks) {
      const int c = ks * 4 + kq, p = p0 + 16 * m + j;
      a[m][ks] = (c < Cin && p < HW) ? inb[(size_t)c * HW + p] : 0.f;
    }
  float* outb = out + (size_t)b * Cout * HW;
  const f32x4* wl = reinterpret_cast<const f32x4*>(wpk) + (size_t)lane * 2;      // this lane's 8 floats of a tile
  f32x4 bq[2][2];
  int nt = wave;
  if (nt < ntiles) { bq[0][0] = wl[(size_t)nt * 128]; bq[0][1] = wl[(size_t)nt * 128 + 1]; }
  for (int it = 0; nt < ntiles; nt += 4, it ^= 1) {
    if (nt + 4 < ntiles) { bq[it ^ 1][0] = wl[(size_t)(nt + 4) * 128]; bq[it ^ 1][1] = wl[(size_t)(nt + 4) * 128 + 1]; }
    f32x4 acc[PW_MT];
#pragma unroll
    for (int m = 0; m < PW_MT; ++m) {
      acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < PW_KS; ++ks)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][ks], bq[it][ks >> 2][ks & 3], acc[m], 0, 0, 0);
    }
    const int co = nt * 16 + j;
    if (co < Cout) {
      float* orow = outb + (size_t)co * HW;
#pragma unroll
      for (int m = 0; m < PW_MT; ++m) {
        const int p = p0 + 16 * m + 4 * kq;        // accumulator rows 4 kq .. 4 kq + 3 = four consecutive pixels
        if (VEC) {
          if (p < HW) *reinterpret_cast<f32x4*>(orow + p) = acc[m];      // HW % 4 == 0: the quad is all in or all out
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (p + r < HW) orow[p + r] = acc[m][r];
        }
      }
    }
  }
}

}  // namespace

extern "C" size_t dv_pointwise_expand_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cin > 4 * PW_KS || Cout <= 0) return 0;
  return (size_t)((Cout + 15) / 16) * 4 * 16 * PW_KS;
}

extern "C" int dv_pointwise_expand_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout, dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cin <= 4 * PW_KS && Cout > 0, DV_ERR_SHAPE);
  const int ntiles = (Cout + 15) / 16;
  const int total = ntiles * 4 * 16 * PW_KS;
  hipLaunchKernelGGL(pw_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wpacked, Cin, Cout, ntiles);
  return dv_launch_status();
}

extern "C" int dv_pointwise_expand_f32(const float* in, const float* wpacked, float* out, int B, int Cin, int HW, int Cout,
                                       dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && B <= 65535 && Cin > 0 && Cin <= 4 * PW_KS && HW > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(dv_aligned16(wpacked), DV_ERR_ALIGN);
  const int ntiles = (Cout + 15) / 16;
  const dim3 grid((unsigned)((HW + 16 * PW_MT - 1) / (16 * PW_MT)), (unsigned)B);
  if (HW % 4 == 0 && dv_aligned16(out))
    hipLaunchKernelGGL(pw_expand_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, wpacked, out, Cin, Cout, HW, ntiles);
  else
    hipLaunchKernelGGL(pw_expand_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, wpacked, out, Cin, Cout, HW, ntiles);
  return dv_launch_status();
}
