// K13: the two depth-wise (1,3,3) convolutions the attention branch applies to the group-wise correlation
// volume before aggregating it (SceneFlow/models/acv_ddim.py:181-188, :377-381; acv.py likewise):
//   gwc  = patch(gwc)                                         Conv3d(40,40,(1,3,3), groups=40, padding (0,1,1))
//   out  = cat(patch_l1(gwc[:, :8]), patch_l2(gwc[:, 8:24]), patch_l3(gwc[:, 24:40]))      dilation 1 / 2 / 3
// Both are per-channel 3x3 stencils inside one (b, g, d) plane, so they fuse into one pass: the input tile
// (+halo 1+dil) goes to LDS, the first stencil is evaluated on the tile + dil halo into a second LDS buffer
// (zero outside the image: the second convolution zero-pads the FIRST convolution's output), the second
// stencil writes the result.  One read + one write of the 1.9 GB volume instead of four PyTorch passes
// through MIOpen's grouped-conv path (26 ms at batch 8).
//
// Round 4 (profiles/r04_e2e_stages.json: 2.07 ms per call = 1.8 TB/s, the HBM-bound kernel of the end-to-end path
// furthest from its roofline): the fast path (W % 4 == 0, 16-byte aligned tensors) moves everything in 16-byte units --
// global loads / stores, and both stencils on four consecutive x per thread from aligned ds_read_b128 (9 LDS reads per
// four outputs instead of 36, no division per element), the dilation a template parameter so that every tap is a
// register index.  The element-wise kernel stays for other widths.
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

// ---- fast path ----------------------------------------------------------------------------------------------------
constexpr int FIW = TXB + 16;        // staged input row: x0 - 8 .. x0 + 136 (tile at column 8; the outer quads are slack)
constexpr int FMW = TXB + 8;         // first-stencil row: x0 - 4 .. x0 + 132 (tile at column 4)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DL>
__global__ __launch_bounds__(256) void patch_volume_vec_kernel(const float* __restrict__ in, const float* __restrict__ w1,
                                                               const float* __restrict__ w2, float* __restrict__ out,
                                                               long long plane0, int g0, int ng, int G, int D, int H, int W) {
  constexpr int H1 = 1 + DL, IHV = TY + 2 * H1, MHV = TY + 2 * DL;
  __shared__ __attribute__((aligned(16))) float in_s[IHV][FIW];
  __shared__ __attribute__((aligned(16))) float mid_s[MHV][FMW];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TXB, y0 = blockIdx.y * TY;
  // blockIdx.z walks the (b, g in [g0, g0+ng), d) planes of this dilation group
  const unsigned z = (unsigned)(plane0 + blockIdx.z);
  const unsigned d = z % (unsigned)D, r1 = z / (unsigned)D;
  const int g = g0 + (int)(r1 % (unsigned)ng), b = (int)(r1 / (unsigned)ng);
  const size_t pl = ((size_t)b * G + g) * D + d;
  const float* src = in + pl * (size_t)H * W;
  float* dst = out + pl * (size_t)H * W;
  float a[9], c[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { a[i] = w1[g * 9 + i]; c[i] = w2[g * 9 + i]; }

  // input rows y0 - H1 .. y0 + TY + H1, columns x0 - 4 .. x0 + TXB + 4 as quads (W % 4 == 0: a quad is all in or all out)
  constexpr int IQ = (TXB + 8) / 4;
  for (int e = tid; e < IHV * IQ; e += 256) {
    const int r = e / IQ, q = e - r * IQ;
    const int y = y0 - H1 + r, x = x0 - 4 + 4 * q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = *reinterpret_cast<const f32x4*>(src + (size_t)y * W + x);
    *reinterpret_cast<f32x4*>(&in_s[r][4 + 4 * q]) = v;
  }
  __syncthreads();
  // first stencil on rows y0 - DL .. y0 + TY + DL, columns x0 - 4 .. x0 + TXB + 4 (quads; the outermost column of each
  // side is never read back).  Zero outside the image: that is the zero padding of the second convolution.
  constexpr int MQ = FMW / 4;
  for (int e = tid; e < MHV * MQ; e += 256) {
    const int r = e / MQ, q = e - r * MQ;
    const int y = y0 - DL + r, x = x0 - 4 + 4 * q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // image row y + ky - 1 is in_s row (y + ky - 1) - (y0 - H1) = r + ky; image column x is in_s column x - x0 + 8
        const float* row = &in_s[r + ky][4 * q];            // columns x - 4 .. x + 7 of the image
        const f32x4 lo = *reinterpret_cast<const f32x4*>(row), mi = *reinterpret_cast<const f32x4*>(row + 4),
                    hi = *reinterpret_cast<const f32x4*>(row + 8);
        const float t[6] = {lo[3], mi[0], mi[1], mi[2], mi[3], hi[0]};        // x - 1 .. x + 4
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) v[i] = fmaf(a[ky * 3 + kx], t[i + kx], v[i]);
      }
    }
    *reinterpret_cast<f32x4*>(&mid_s[r][4 * q]) = v;
  }
  __syncthreads();
  constexpr int OQ = TXB / 4;
  for (int e = tid; e < TY * OQ; e += 256) {
    const int r = e / OQ, q = e - r * OQ;
    const int y = y0 + r, x = x0 + 4 * q;
    if (y >= H || x >= W) continue;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      // image row y + (ky - 1) DL is mid_s row r + ky DL; image column x is mid_s column x - x0 + 4
      const float* row = &mid_s[r + ky * DL][4 * q];          // image x - 4 .. x + 7
      const f32x4 lo = *reinterpret_cast<const f32x4*>(row), mi = *reinterpret_cast<const f32x4*>(row + 4),
                  hi = *reinterpret_cast<const f32x4*>(row + 8);
      const float t[12] = {lo[0], lo[1], lo[2], lo[3], mi[0], mi[1], mi[2], mi[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v[i] = fmaf(c[ky * 3 + kx], t[4 + i + (kx - 1) * DL], v[i]);
    }
    *reinterpret_cast<f32x4*>(dst + (size_t)y * W + x) = v;
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


// The same pass with the dilation given as HOST-side runs of consecutive groups (acv_ddim.py:181-188: dilation 1 for
// groups 0-7, 2 for 8-23, 3 for 24-39): one launch per run on the 16-byte fast path when W % 4 == 0 and the tensors are
// 16-byte aligned, the element-wise kernel otherwise (the dilation table is then read from `dilation_dev`).
extern "C" int dv_patch_volume_runs_f32(const float* gwc, const float* w1, const float* w2, const int* dilation_dev,
                                        float* out, int B, int G, int D, int H, int W, int nruns, const int* run_g0,
                                        const int* run_ng, const int* run_dil, dv_stream_t stream) {
  DV_REQUIRE_PTR(gwc);
  DV_REQUIRE_PTR(w1);
  DV_REQUIRE_PTR(w2);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && G > 0 && D > 0 && H > 0 && W > 0 && nruns > 0, DV_ERR_SHAPE);
  DV_REQUIRE_PTR(run_g0);
  DV_REQUIRE_PTR(run_ng);
  DV_REQUIRE_PTR(run_dil);
  int covered = 0;
  for (int r = 0; r < nruns; ++r) {
    DV_REQUIRE(run_g0[r] == covered && run_ng[r] > 0 && run_dil[r] >= 1 && run_dil[r] <= 3, DV_ERR_SHAPE);
    covered += run_ng[r];
  }
  DV_REQUIRE(covered == G, DV_ERR_SHAPE);
  const bool fast = (W % 4 == 0) && dv_aligned16(gwc) && dv_aligned16(out);
  if (!fast) {
    DV_REQUIRE_PTR(dilation_dev);
    return dv_patch_volume_f32(gwc, w1, w2, dilation_dev, out, B, G, D, H, W, stream);
  }
  DV_REQUIRE((H + TY - 1) / TY <= 65535, DV_ERR_SHAPE);
  const int gx = (W + TXB - 1) / TXB, gy = (H + TY - 1) / TY;
  for (int r = 0; r < nruns; ++r) {
    const long long planes = (long long)B * run_ng[r] * D;
    DV_REQUIRE(planes <= 0x7fffffffLL, DV_ERR_SHAPE);
    for (long long p0 = 0; p0 < planes; p0 += 65535) {
      const long long n = planes - p0 < 65535 ? planes - p0 : 65535;
      const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)n);
      switch (run_dil[r]) {
        case 1: hipLaunchKernelGGL(patch_volume_vec_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, gwc, w1, w2, out, p0,
                                   run_g0[r], run_ng[r], G, D, H, W); break;
        case 2: hipLaunchKernelGGL(patch_volume_vec_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, gwc, w1, w2, out, p0,
                                   run_g0[r], run_ng[r], G, D, H, W); break;
        default: hipLaunchKernelGGL(patch_volume_vec_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, gwc, w1, w2, out, p0,
                                    run_g0[r], run_ng[r], G, D, H, W); break;
      }
    }
  }
  return dv_launch_status();
}
