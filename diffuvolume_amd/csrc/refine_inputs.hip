// K12: input assembly of the KITTI12 per-step 2-D refinement (KITTI12/models/pwcnet_ddim.py:486-502):
//   frw  = warp(right_feature, disp)                       (models/submodule.py:137-176: grid_sample with the grid
//                                                            normalised by W-1 / H-1 but sampled align_corners=False,
//                                                            times the `sampled ones >= 0.999` validity mask)
//   cv   = build_corrleation_volume(left, frw, 24, 1)       (:121-135, +-24 shifts, channel mean; for negative
//                                                            shifts the reference pairs the FIRST |i| columns of the
//                                                            left map with the LAST |i| columns of frw -- kept)
//   comb = cat(left - frw, left, Mish(BN(conv1x1(disp))), disp, cv)        [B, 3C + 1 + 49, H, W]
// One pass: the warped row segment (tile + 24-column halo, plus the row's last 24 columns for the wrap-around
// shifts of the first tile) is built in LDS, every output channel is written once, coalesced along W.  The
// PyTorch composition makes ~110 launches and re-reads the feature maps once per shift.
#include "dv_common.h"

namespace {

constexpr int TX = 128, MD = 24, NSH = 2 * MD + 1, CMAX = 32;
constexpr int SW = TX + 2 * MD;       // staged columns per channel

struct RefArgs {
  const float* left;    // [B,C,H,W]
  const float* right;   // [B,C,H,W]
  const float* disp;    // [B,H,W]
  const float* du_a;    // [C] folded conv1x1 * BN scale of `dispupsample`
  const float* du_b;    // [C] folded BN shift
  float* out;           // [B,3C+1+49,H,W]
  int B, C, H, W;
};

// bilinear taps of grid_sample(align_corners=False, zeros padding) for output pixel (y, x) shifted by d
struct Taps {
  int x0, y0;           // north-west corner
  float nw, ne, sw, se; // weights
  float valid;          // 1 if the sampled all-ones image is >= 0.999, else 0
};

__device__ __forceinline__ Taps make_taps(float x, float y, float d, int W, int H) {
  const float gx = 2.0f * (x - d) / (float)(W > 1 ? W - 1 : 1) - 1.0f;
  const float gy = 2.0f * y / (float)(H > 1 ? H - 1 : 1) - 1.0f;
  const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
  const float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  Taps t;
  t.x0 = (int)fx; t.y0 = (int)fy;
  t.nw = (fx + 1.f - ix) * (fy + 1.f - iy);
  t.ne = (ix - fx) * (fy + 1.f - iy);
  t.sw = (fx + 1.f - ix) * (iy - fy);
  t.se = (ix - fx) * (iy - fy);
  const bool xl = (unsigned)t.x0 < (unsigned)W, xr = (unsigned)(t.x0 + 1) < (unsigned)W;
  const bool yt = (unsigned)t.y0 < (unsigned)H, yb = (unsigned)(t.y0 + 1) < (unsigned)H;
  float m = 0.f;
  if (xl && yt) m += t.nw;
  if (xr && yt) m += t.ne;
  if (xl && yb) m += t.sw;
  if (xr && yb) m += t.se;
  t.valid = m < 0.999f ? 0.f : 1.f;
  if (!(xl && yt)) t.nw = 0.f;
  if (!(xr && yt)) t.ne = 0.f;
  if (!(xl && yb)) t.sw = 0.f;
  if (!(xr && yb)) t.se = 0.f;
  return t;
}

__device__ __forceinline__ float warp_one(const float* __restrict__ plane, const Taps& t, int W, int H) {
  // out-of-range taps carry weight 0; clamp their addresses
  const int xa = min(max(t.x0, 0), W - 1), xb = min(max(t.x0 + 1, 0), W - 1);
  const int ya = min(max(t.y0, 0), H - 1), yb = min(max(t.y0 + 1, 0), H - 1);
  float v = plane[(size_t)ya * W + xa] * t.nw;
  v += plane[(size_t)ya * W + xb] * t.ne;
  v += plane[(size_t)yb * W + xa] * t.sw;
  v += plane[(size_t)yb * W + xb] * t.se;
  return v * t.valid;
}

__global__ __launch_bounds__(256) void refine_inputs_kernel(RefArgs a) {
  __shared__ float frw_s[CMAX][SW];      // warped right features, columns x0-MD .. x0+TX+MD-1
  __shared__ float wrap_s[CMAX][MD];     // columns W-MD .. W-1 of the same row (first tile only)
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TX, y = blockIdx.y, b = blockIdx.z;
  const int W = a.W, H = a.H, C = a.C;
  const size_t plane = (size_t)H * W;
  const float* lb = a.left + (size_t)b * C * plane;
  const float* rb = a.right + (size_t)b * C * plane;
  const float* db = a.disp + (size_t)b * plane + (size_t)y * W;

  // ---- stage the warped row segment: thread = column, loop over channels ----
  for (int col = tid; col < SW + MD; col += 256) {
    const bool wrap = col >= SW;
    if (wrap && x0 != 0) break;
    const int x = wrap ? W - MD + (col - SW) : x0 - MD + col;
    if ((unsigned)x < (unsigned)W) {
      const Taps t = make_taps((float)x, (float)y, db[x], W, H);
      for (int c = 0; c < C; ++c) {
        const float v = warp_one(rb + (size_t)c * plane, t, W, H);
        if (wrap) wrap_s[c][col - SW] = v; else frw_s[c][col] = v;
      }
    } else {
      for (int c = 0; c < C; ++c)
        if (wrap) wrap_s[c][col - SW] = 0.f; else frw_s[c][col] = 0.f;
    }
  }
  __syncthreads();

  // ---- outputs: two threads per pixel; half 0: left - frw, disp, shifts -24..0; half 1: left, dispupsample, 1..24 ----
  const int px = tid & (TX - 1), half = tid >> 7;
  const int x = x0 + px;
  if (x >= W) return;
  const size_t row = (size_t)y * W + x;
  float* ob = a.out + (size_t)b * (3 * C + 1 + NSH) * plane + row;
  float lv[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) lv[c] = c < C ? lb[(size_t)c * plane + row] : 0.f;
  const float d = db[x];
  const float inv_c = 1.f / (float)C;
  if (half == 0) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) ob[(size_t)c * plane] = lv[c] - frw_s[c][MD + px];
    ob[(size_t)(3 * C) * plane] = d;
  } else {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        ob[(size_t)(C + c) * plane] = lv[c];
        ob[(size_t)(2 * C + c) * plane] = dv_act(fmaf(a.du_a[c], d, a.du_b[c]), DV_ACT_MISH);
      }
  }
  float* cvb = ob + (size_t)(3 * C + 1) * plane;
  const int s_lo = half == 0 ? -MD : 1, s_hi = half == 0 ? 0 : MD;
  for (int i = s_lo; i <= s_hi; ++i) {
    float acc = 0.f;
    if (i >= 0) {
      if (x >= i) {                                    // ref[x] * tgt[x - i]
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) acc += lv[c] * frw_s[c][MD + px - i];
      }
    } else if (x < -i) {                               // first |i| columns against the LAST |i| columns (as written)
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) acc += lv[c] * wrap_s[c][MD + i + x];
    }
    cvb[(size_t)(i + MD) * plane] = acc * inv_c;
  }
}

}  // namespace

extern "C" int dv_refine_inputs_f32(const float* left, const float* right, const float* disp, const float* du_a,
                                    const float* du_b, float* out, int B, int C, int H, int W, int maxshift,
                                    dv_stream_t stream) {
  DV_REQUIRE_PTR(left);
  DV_REQUIRE_PTR(right);
  DV_REQUIRE_PTR(disp);
  DV_REQUIRE_PTR(du_a);
  DV_REQUIRE_PTR(du_b);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DV_ERR_SHAPE);
  DV_REQUIRE(C <= CMAX && maxshift == MD && W >= MD, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(B <= 65535 && H <= 65535, DV_ERR_SHAPE);
  RefArgs a{left, right, disp, du_a, du_b, out, B, C, H, W};
  hipLaunchKernelGGL(refine_inputs_kernel, dim3((unsigned)((W + TX - 1) / TX), (unsigned)H, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, a);
  return dv_launch_status();
}

// ---- space-to-batch for the dilated layers of the refinement network (KITTI12/models/pwcnet_ddim.py:251-306) ----
// A 3x3 convolution with dilation d is d*d independent dilation-1 convolutions on the sub-images (y % d, x % d).  Run on
// the image as it lies, every load / store of a sub-image is a 4-byte access at a 4 d-byte stride (d = 8: a wave touches 16
// cache lines for 64 floats; conv2d_wino at d = 8 issued 0.34 of the matrix pipe, 0.56 at d = 1).  The dilations of
// refinenet_version3 double from layer to layer (1, 1, 2, 4, 8, 8, 16, 16, 1...), so the stack runs in the sub-image domain
// instead: one de-interleave by 2 in front of every doubling turns [N, C, h, w] into [4 N, C, h/2, w/2] (sub-image
// (y & 1, x & 1) of item n becomes item 4 n + 2 (y & 1) + (x & 1)), every dilated layer -- and the residual block around it:
// 1x1 down-sampling, BatchNorm, Mish and the skip add are pointwise -- becomes a plain dense convolution on that tensor,
// and one pass puts the pixels back after the last dilated block.
namespace {

// thread = two neighbouring x of one row: reads 8 bytes, writes one float to each of the two sub-images of that row parity
// (a wave reads 512 contiguous bytes and writes two runs of 256)
__global__ void space_to_batch2_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int h, int w,
                                       size_t pairs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pairs) return;
  const int w2 = w >> 1, h2 = h >> 1;
  const int xp = (int)(i % w2);
  size_t r = i / w2;
  const int y = (int)(r % h); r /= h;
  const int c = (int)(r % C);
  const size_t n = r / C;
  const float2 v = *reinterpret_cast<const float2*>(in + (((n * C + c) * h + y) * (size_t)w + 2 * xp));
  const size_t plane = (size_t)h2 * w2;
  const size_t o = (((n * 4 + 2 * (y & 1)) * C + c) * plane) + (size_t)(y >> 1) * w2 + xp;
  out[o] = v.x;
  out[o + (size_t)C * plane] = v.y;
}

// the inverse of `levels` de-interleaves at once: thread = one output pixel (b, c, y, x) of [B, C, H, W]
__global__ void batch_to_space_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W, int levels,
                                      size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  size_t r = i / W;
  const int y = (int)(r % H); r /= H;
  const int c = (int)(r % C);
  size_t n = r / C;
  for (int l = 0; l < levels; ++l) n = n * 4 + 2 * ((y >> l) & 1) + ((x >> l) & 1);
  const int hs = H >> levels, ws = W >> levels;
  out[i] = in[((n * C + c) * hs + (y >> levels)) * (size_t)ws + (x >> levels)];
}

}  // namespace

extern "C" int dv_space_to_batch2_f32(const float* in, float* out, int N, int C, int h, int w, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(N > 0 && C > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, DV_ERR_SHAPE);
  DV_REQUIRE((((uintptr_t)in) & 7u) == 0, DV_ERR_ALIGN);
  const size_t pairs = (size_t)N * C * h * (w / 2);
  DV_REQUIRE((pairs + 255) / 256 <= 0x7fffffffull, DV_ERR_SHAPE);
  hipLaunchKernelGGL(space_to_batch2_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     C, h, w, pairs);
  return dv_launch_status();
}

extern "C" int dv_batch_to_space_f32(const float* in, float* out, int B, int C, int H, int W, int levels,
                                     dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && levels >= 1 && levels <= 8, DV_ERR_SHAPE);
  DV_REQUIRE(H % (1 << levels) == 0 && W % (1 << levels) == 0, DV_ERR_SHAPE);
  const size_t total = (size_t)B * C * H * W;
  DV_REQUIRE((total + 255) / 256 <= 0x7fffffffull, DV_ERR_SHAPE);
  hipLaunchKernelGGL(batch_to_space_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     C, H, W, levels, total);
  return dv_launch_status();
}
