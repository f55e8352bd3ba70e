// K13: the two depth-wise (1,3,3) convolutions the attention branch applies to the group-wise correlation
// volume before aggregating it (SceneFlow/models/acv_ddim.py:181-188, :377-381; acv.py likewise):
//   gwc  = patch(gwc)                                         Conv3d(40,40,(1,3,3), groups=40, padding (0,1,1))
//   out  = cat(patch_l1(gwc[:, :8]), patch_l2(gwc[:, 8:24]), patch_l3(gwc[:, 24:40]))      dilation 1 / 2 / 3
// Both are per-channel 3x3 stencils inside one (b, g, d) plane, so they fuse into one pass: the input tile
// (+halo 1+dil) goes to LDS, the first stencil is evaluated on the tile + dil halo into a second LDS buffer
// (zero outside the image: the second convolution zero-pads the FIRST convolution's output), the second
// stencil writes the result.  One read + one write of the 1.9 GB volume instead of four PyTorch passes
// through MIOpen's grouped-conv path (26 ms at batch 8).
#include "dv_common.h"

namespace {

constexpr int TY = 16, TXB = 128, HMAX = 4;            // output tile; halo = 1 + dilation <= 4
constexpr int IW = TXB + 2 * HMAX, IH = TY + 2 * HMAX; // staged input
constexpr int MW = TXB + 2 * 3, MH = TY + 2 * 3;       // first-stencil output (halo = dilation <= 3)

__global__ __launch_bounds__(256) void patch_volume_kernel(const float* __restrict__ in, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, const int* __restrict__ dil,
                                                           float* __restrict__ out, long long plane0, int G, int D,
                                                           int H, int W) {
  __shared__ float in_s[IH][IW];
  __shared__ float mid_s[MH][MW];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TXB, y0 = blockIdx.y * TY;
  const size_t pl = (size_t)plane0 + blockIdx.z;        // (b * G + g) * D + d
  const int g = (int)((pl / D) % G);
  const int dl = dil[g], h1 = 1 + dl;
  const float* src = in + pl * (size_t)H * W;
  float* dst = out + pl * (size_t)H * W;
  float a[9], c[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { a[i] = w1[g * 9 + i]; c[i] = w2[g * 9 + i]; }

  // input tile with halo h1 (zero outside the image)
  const int ih = TY + 2 * h1, iw = TXB + 2 * h1;
  for (int e = tid; e < ih * iw; e += 256) {
    const int r = e / iw, q = e - r * iw;
    const int y = y0 - h1 + r, x = x0 - h1 + q;
    in_s[r][q] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? src[(size_t)y * W + x] : 0.f;
  }
  __syncthreads();
  // first stencil on the tile + halo dl; positions outside the image are the second conv's zero padding
  const int mh = TY + 2 * dl, mw = TXB + 2 * dl;
  for (int e = tid; e < mh * mw; e += 256) {
    const int r = e / mw, q = e - r * mw;
    const int y = y0 - dl + r, x = x0 - dl + q;
    float v = 0.f;
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v = fmaf(a[ky * 3 + kx], in_s[r + ky][q + kx], v);
    }
    mid_s[r][q] = v;
  }
  __syncthreads();
  for (int e = tid; e < TY * TXB; e += 256) {
    const int r = e / TXB, q = e - r * TXB;
    const int y = y0 + r, x = x0 + q;
    if (y >= H || x >= W) continue;
    float v = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) v = fmaf(c[ky * 3 + kx], mid_s[r + ky * dl][q + kx * dl], v);
    dst[(size_t)y * W + x] = v;
  }
}

}  // namespace

extern "C" int dv_patch_volume_f32(const float* gwc, const float* w1, const float* w2, const int* dilation, float* out,
                                   int B, int G, int D, int H, int W, dv_stream_t stream) {
  DV_REQUIRE_PTR(gwc);
  DV_REQUIRE_PTR(w1);
  DV_REQUIRE_PTR(w2);
  DV_REQUIRE_PTR(dilation);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && G > 0 && D > 0 && H > 0 && W > 0, DV_ERR_SHAPE);
  const long long planes = (long long)B * G * D;
  DV_REQUIRE(planes <= 0x7fffffffLL && (H + TY - 1) / TY <= 65535, DV_ERR_SHAPE);
  const int gx = (W + TXB - 1) / TXB, gy = (H + TY - 1) / TY;
  for (long long p0 = 0; p0 < planes; p0 += 65535) {    // grid.z is limited to 65535 planes per launch
    const long long n = planes - p0 < 65535 ? planes - p0 : 65535;
    hipLaunchKernelGGL(patch_volume_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, gwc, w1, w2, dilation, out, p0, G, D, H, W);
  }
  return dv_launch_status();
}
