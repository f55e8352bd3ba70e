// K10: IGEV geometry-encoding-volume lookup WITH the DiffuVolume noise filter.
// Replaces Combined_Geo_Encoding_Volume.__call__ (KITTI15/core/geometry_ddim.py:33-69) --
// executed iters x steps times per pair.  The reference multiplies the WHOLE pyramid level by the
// noise (`geo_volume * noi`, :56) and then grid_samples 9 taps per pixel; here every pixel reads only
// the <=10 disparity entries its taps touch, multiplies them by the noise on the fly, and the level-1
// pyramids (avg_pool over pairs of disparities, :24-25,:41-43) are formed in registers.
//
//   out[b, 0:72 ]  = lerp_x( geo[c,:]*noi0[:] ,  disp      + dx )   c<8, dx=-4..4   (channel = c*9+tap)
//   out[b, 72:81]  = lerp_x( corr0[:],           coords-disp + dx )
//   out[b, 81:153] = lerp_x( pool2(geo[c,:])*pool2(noi0)[:], disp/2 + dx )
//   out[b,153:162] = lerp_x( corr1[:],           coords/2-disp/2 + dx )
// with bilinear, zero padding, align_corners=True in pixel units (utils.py:59-77).
// Quirk kept: the noise row of pixel n is the n-th run of D floats of the flat
// [B,D,h,w] tensor (raw reshape, geometry_ddim.py:37), not the pixel's own channel column.
#include "dv_common.h"

namespace {

// bilinear_sampler + grid_sample(align_corners=True, zeros) along one axis of length n, in the
// reference's float order: xg = 2x/(n-1) - 1 ; ix = ((xg+1)/2)*(n-1)
__device__ __forceinline__ void sample_pos(float x, int n, int& i0, float& w0, float& w1) {
  const float xg = 2.0f * x / (float)(n - 1) - 1.0f;
  const float ix = ((xg + 1.0f) / 2.0f) * (float)(n - 1);
  const float fl = floorf(ix);
  i0 = (int)fl;
  w0 = (fl + 1.0f) - ix;   // weight of i0   (ix_ne - ix)
  w1 = ix - fl;            // weight of i0+1
}

template <int R>
__global__ __launch_bounds__(256) void geo_lookup_kernel(const float* __restrict__ geo,
                                                         const float* __restrict__ corr0,
                                                         const float* __restrict__ corr1,
                                                         const float* __restrict__ disp,
                                                         const float* __restrict__ coords,
                                                         const float* __restrict__ noisy,
                                                         float* __restrict__ out, int C, int D, int h, int w,
                                                         int W2, size_t npix) {
  constexpr int T = 2 * R + 1;
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= npix) return;
  const size_t hw = (size_t)h * w;
  const size_t b = n / hw, p = n - b * hw;
  const float d = disp[n], cx = coords[n];
  const float* nrow = noisy + n * D;                    // raw-reshape row (quirk)
  const float* g = geo + b * C * D * hw + p;            // + (c*D + dd) * hw
  const int W2b = W2 / 2, D1 = D / 2;
  const int nch = 2 * (C * T + T);
  float* o = out + b * nch * hw + p;                    // + channel * hw
  auto noi0 = [&](int i) { return nrow[i]; };
  auto noi1 = [&](int i) { return (nrow[2 * i] + nrow[2 * i + 1]) * 0.5f; };
  // ---- level 0 ----
  for (int t = 0; t < T; ++t) {
    int i0; float w0, w1;
    sample_pos(d + (float)(t - R), D, i0, w0, w1);
    const bool ok0 = i0 >= 0 && i0 < D, ok1 = i0 + 1 >= 0 && i0 + 1 < D;
    const float n0 = ok0 ? noi0(i0) : 0.f, n1 = ok1 ? noi0(i0 + 1) : 0.f;
    for (int c = 0; c < C; ++c) {
      const float v0 = ok0 ? g[((size_t)c * D + i0) * hw] * n0 : 0.f;
      const float v1 = ok1 ? g[((size_t)c * D + i0 + 1) * hw] * n1 : 0.f;
      o[(size_t)(c * T + t) * hw] = v0 * w0 + v1 * w1;
    }
    sample_pos(cx - d + (float)(t - R), W2, i0, w0, w1);
    const float* cr = corr0 + n * W2;
    const float c0 = (i0 >= 0 && i0 < W2) ? cr[i0] : 0.f, c1 = (i0 + 1 >= 0 && i0 + 1 < W2) ? cr[i0 + 1] : 0.f;
    o[(size_t)(C * T + t) * hw] = c0 * w0 + c1 * w1;
  }
  // ---- level 1 (pairs of disparities averaged) ----
  float* o1 = o + (size_t)(C * T + T) * hw;
  for (int t = 0; t < T; ++t) {
    int i0; float w0, w1;
    sample_pos(d / 2.0f + (float)(t - R), D1, i0, w0, w1);
    const bool ok0 = i0 >= 0 && i0 < D1, ok1 = i0 + 1 >= 0 && i0 + 1 < D1;
    const float n0 = ok0 ? noi1(i0) : 0.f, n1 = ok1 ? noi1(i0 + 1) : 0.f;
    for (int c = 0; c < C; ++c) {
      const float* gc = g + (size_t)c * D * hw;
      const float v0 = ok0 ? ((gc[(size_t)(2 * i0) * hw] + gc[(size_t)(2 * i0 + 1) * hw]) * 0.5f) * n0 : 0.f;
      const float v1 = ok1 ? ((gc[(size_t)(2 * i0 + 2) * hw] + gc[(size_t)(2 * i0 + 3) * hw]) * 0.5f) * n1 : 0.f;
      o1[(size_t)(c * T + t) * hw] = v0 * w0 + v1 * w1;
    }
    sample_pos(cx / 2.0f - d / 2.0f + (float)(t - R), W2b, i0, w0, w1);
    const float* cr = corr1 + n * W2b;
    const float c0 = (i0 >= 0 && i0 < W2b) ? cr[i0] : 0.f, c1 = (i0 + 1 >= 0 && i0 + 1 < W2b) ? cr[i0 + 1] : 0.f;
    o1[(size_t)(C * T + t) * hw] = c0 * w0 + c1 * w1;
  }
}

}  // namespace

extern "C" int dv_geo_filter_lookup_f32(const float* geo, const float* corr0, const float* corr1,
                                        const float* disp, const float* coords, const float* noisy, float* out,
                                        int B, int C, int D, int h, int w, int W2, int radius,
                                        dv_stream_t stream) {
  DV_REQUIRE_PTR(geo);
  DV_REQUIRE_PTR(corr0);
  DV_REQUIRE_PTR(corr1);
  DV_REQUIRE_PTR(disp);
  DV_REQUIRE_PTR(coords);
  DV_REQUIRE_PTR(noisy);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && C > 0 && D > 3 && h > 0 && w > 0 && W2 > 3, DV_ERR_SHAPE);
  DV_REQUIRE(radius == 4, DV_ERR_UNSUPPORTED);   // corr_radius of every IGEV config (evaluate_stereo.py)
  const size_t npix = (size_t)B * h * w;
  hipLaunchKernelGGL(geo_lookup_kernel<4>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     geo, corr0, corr1, disp, coords, noisy, out, C, D, h, w, W2, npix);
  return dv_launch_status();
}
