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

// Round 3: the taps of a pixel overlap -- level 0 touches disparities floor(d) - 5 .. floor(d) + 6 and level 1 (pairs of
// them) 2 floor(d/2) - 10 .. 2 floor(d/2) + 13 -- so a thread first copies the 24 entries dlo .. dlo + 23, dlo =
// 2 floor(d/2) - 10, of a channel (and once of its noise row) into a column of LDS that only it touches, and every tap
// reads from there by its own index (the sample positions keep the reference's float arithmetic, so an index may sit
// one off the integer expectation: the window has room for that).  24 gathers per channel instead of 54: the gathers
// are what the kernel costs when neighbouring pixels disagree about d (each lane then pulls its own cache line).
//
// Round 5: WHICH lanes read WHAT.  The window copy used to be 24 gathers per lane at the lane's own dlo: where neighbouring
// pixels disagree about d (random-init weights, occlusion borders) every lane of a load instruction pulls a different plane,
// i.e. up to 64 cache lines per instruction -- the kernel ran at 145 us per call at batch 4 for ~230 MB of traffic.  Now a
// wave walks the planes k = min(dlo) .. max(dlo) + 23 of its 64 pixels TOGETHER: every lane loads plane k at its own pixel
// (one 256-byte segment per instruction, whatever the disparities are) and keeps the value if k lies in its window.  Smooth
// disparity: 24-26 coalesced loads per channel; the worst case is the D planes of the channel once.
constexpr int GEO_WIN = 24;
constexpr int GEO_BATCH = 8;        // planes requested per round trip

// FUSED: the lookup together with the 1x1 convolution that is its only consumer in IGEV's update block
// (BasicMotionEncoder.convc1 + bias + ReLU, KITTI15/core/update.py:79,:89: 2*(C*9+9) = 162 -> 64 channels), so that the
// [B,162,h,w] tensor -- 78 MB written and read back per GRU iteration at batch 4 -- never exists; out is [B,64,h,w].
// The lookup produces, per geometry channel c, 18 values per pixel (9 taps x 2 levels; the two correlation rows make a
// ninth group): a wave parks them in a private LDS tile vals[k][pixel] and multiplies on the matrix cores,
//     acc[cout][pixel] += Wk[group][k][cout] * vals[k][pixel]       (v_mfma_f32_16x16x4_f32: M = cout, N = pixel, K = k)
// with K padded from 18 to 20 per group (two zero rows).  The A fragments (weights, 20 floats per lane and group) come
// straight from global memory (41 KB, L2-resident), requested before the group's gathers; the 64 x 64 accumulator tile of the
// wave stays in registers until the ReLU.  The sum over the 162 lookup channels has a fixed order: bit-reproducible.
typedef float geo_f32x4 __attribute__((ext_vector_type(4)));
constexpr int GEO_KG = 20;          // k rows per group (18 used)
constexpr int GEO_VS = 80;          // row stride of the value tile: the four k rows of a B fragment fall on disjoint banks
constexpr int GEO_NF = 64;          // output channels of the fused convolution

template <int R, bool FUSED>
__global__ __launch_bounds__(256) void geo_lookup_kernel(const float* __restrict__ geo,
                                                         const float* __restrict__ corr0,
                                                         const float* __restrict__ corr1,
                                                         const float* __restrict__ disp,
                                                         const float* __restrict__ coords,
                                                         const float* __restrict__ noisy,
                                                         const float* __restrict__ wk,      // FUSED: [C+1][20][64]
                                                         const float* __restrict__ bias,    // FUSED: [64] or null
                                                         float* __restrict__ out, int C, int D, int h, int w,
                                                         int W2, size_t npix, int act) {
  constexpr int T = 2 * R + 1;
  static_assert(R == 4, "window sized for radius 4");
  __shared__ float gwin[GEO_WIN * 256];
  __shared__ float nwin[GEO_WIN * 256];
  __shared__ float vals_s[FUSED ? 4 * GEO_KG * GEO_VS : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mj = lane & 15, kq = lane >> 4;
  float* const vals = vals_s + (FUSED ? wave * GEO_KG * GEO_VS : 0);
  geo_f32x4 acc[FUSED ? 4 : 1][FUSED ? 4 : 1];
  if (FUSED) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (geo_f32x4){0.f, 0.f, 0.f, 0.f};
    vals[18 * GEO_VS + lane] = 0.f;                     // the two padding rows of K
    vals[19 * GEO_VS + lane] = 0.f;
  }
  // FUSED: the A fragments of group g (this lane: cout 16*mt + mj, k rows 4*ks + kq), and the group's MFMAs
  float wa[FUSED ? 20 : 1];
  auto fetch_w = [&](int g) __attribute__((always_inline)) {
    if (FUSED) {
      const float* wg = wk + (size_t)g * (GEO_KG * GEO_NF) + kq * GEO_NF + mj;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) wa[ks * 4 + mt] = wg[ks * 4 * GEO_NF + mt * 16];
    }
  };
  auto multiply = [&]() __attribute__((always_inline)) {
    if (FUSED) {
      __builtin_amdgcn_wave_barrier();                  // (LDS requests of a wave are served in order: writes, then reads)
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        float bv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bv[nt] = vals[(ks * 4 + kq) * GEO_VS + nt * 16 + mj];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks * 4 + mt], bv[nt], acc[mt][nt], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
    }
  };
  const int tid = threadIdx.x;
  const size_t n0 = (size_t)blockIdx.x * blockDim.x + tid;
  const bool live = n0 < npix;
  const size_t n = live ? n0 : npix - 1;                // (a lane past the end shadows the last pixel and stores nothing)
  const size_t hw = (size_t)h * w;
  const size_t b = n / hw, p = n - b * hw;
  const float d = disp[n], cx = coords[n];
  const float* nrow = noisy + n * D;                    // raw-reshape row (quirk)
  const float* g = geo + b * C * D * hw + p;            // + (c*D + dd) * hw
  const int W2b = W2 / 2, D1 = D / 2;
  const int nch = 2 * (C * T + T);
  float* o = out + (FUSED ? 0 : b * nch * hw + p);      // + channel * hw
  const int dlo = 2 * (int)floorf(d * 0.5f) - 10;
  float* gw = gwin + tid;                               // entry k at gw[k * 256]
  float* nw = nwin + tid;
  if ((D & 3) == 0 && ((reinterpret_cast<uintptr_t>(noisy) & 15u) == 0)) {   // (uniform: an offset view of the noise takes the per-float loop)
    // the pixel's noise window is 96 contiguous bytes of its own row: seven aligned 16-byte loads (dlo is even, the row
    // starts on a 16-byte boundary when D % 4 == 0) instead of 24 single floats -- every one of them a different cache
    // line per lane
    const int base = dlo & ~3;
#pragma unroll
    for (int k = 0; k < GEO_WIN; ++k) { nw[k * 256] = 0.f; if ((unsigned)(dlo + k) >= (unsigned)D) gw[k * 256] = 0.f; }
    float4 nq[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int kk = base + 4 * q;
      nq[q] = (unsigned)kk < (unsigned)D ? *reinterpret_cast<const float4*>(nrow + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const float e4[4] = {nq[q].x, nq[q].y, nq[q].z, nq[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int slot = base + 4 * q + e - dlo;          // (base - dlo is 0 or -2)
        if ((unsigned)slot < (unsigned)GEO_WIN) nw[slot * 256] = e4[e];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < GEO_WIN; ++k) {
      const bool in = (unsigned)(dlo + k) < (unsigned)D;
      nw[k * 256] = in ? nrow[dlo + k] : 0.f;
      if (!in) gw[k * 256] = 0.f;                       // slots outside [0, D) stay zero for every channel
    }
  }
  // the planes this wave walks: the union of its lanes' windows inside [0, D)
  int klo = dlo, khi = dlo;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    klo = min(klo, __shfl_xor(klo, off));
    khi = max(khi, __shfl_xor(khi, off));
  }
  klo = __builtin_amdgcn_readfirstlane(max(klo, 0));
  khi = __builtin_amdgcn_readfirstlane(min(khi + GEO_WIN, D));     // exclusive
  // sample positions of the 2 x 9 taps (the reference's arithmetic), as window slots
  int s0[T], s1[T];
  float a0[T], b0[T], a1[T], b1[T];
  bool ok10[T], ok11[T];                                // level 1: pooled entries i0, i0 + 1 inside [0, D / 2)
#pragma unroll
  for (int t = 0; t < T; ++t) {
    int i0;
    sample_pos(d + (float)(t - R), D, i0, a0[t], b0[t]);
    s0[t] = i0 - dlo;                                   // entries s0, s0 + 1 (5 .. 18 by construction)
    sample_pos(d / 2.0f + (float)(t - R), D1, i0, a1[t], b1[t]);
    s1[t] = 2 * i0 - dlo;                               // entries s1 .. s1 + 3 (0 .. 23)
    ok10[t] = (unsigned)i0 < (unsigned)D1;
    ok11[t] = (unsigned)(i0 + 1) < (unsigned)D1;
    s0[t] = s0[t] < 0 ? 0 : (s0[t] > GEO_WIN - 2 ? GEO_WIN - 2 : s0[t]);     // (never taken: keeps the reads in the column)
    s1[t] = s1[t] < 0 ? 0 : (s1[t] > GEO_WIN - 4 ? GEO_WIN - 4 : s1[t]);
  }
  // noise factors per tap: entries outside [0, D) are zero in both windows, which is the reference's zero padding
  float n00[T], n01[T], n10[T], n11[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    n00[t] = nw[s0[t] * 256];
    n01[t] = nw[(s0[t] + 1) * 256];
    n10[t] = ok10[t] ? (nw[s1[t] * 256] + nw[(s1[t] + 1) * 256]) * 0.5f : 0.f;      // (an odd D has one entry past the pairs)
    n11[t] = ok11[t] ? (nw[(s1[t] + 2) * 256] + nw[(s1[t] + 3) * 256]) * 0.5f : 0.f;
  }
  float* o1 = o + (FUSED ? 0 : (size_t)(C * T + T) * hw);
  for (int c = 0; c < C; ++c) {
    fetch_w(c);
    const float* gc = g + (size_t)c * D * hw;
    for (int k0 = klo; k0 < khi; k0 += GEO_BATCH) {
      float gv[GEO_BATCH];
#pragma unroll
      for (int i = 0; i < GEO_BATCH; ++i) gv[i] = gc[(size_t)min(k0 + i, khi - 1) * hw];      // every lane, plane k0 + i
#pragma unroll
      for (int i = 0; i < GEO_BATCH; ++i) {
        const int slot = min(k0 + i, khi - 1) - dlo;
        if ((unsigned)slot < (unsigned)GEO_WIN) gw[slot * 256] = gv[i];
      }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float v0 = gw[s0[t] * 256] * n00[t], v1 = gw[(s0[t] + 1) * 256] * n01[t];
      if (FUSED) vals[t * GEO_VS + lane] = v0 * a0[t] + v1 * b0[t];
      else if (live) o[(size_t)(c * T + t) * hw] = v0 * a0[t] + v1 * b0[t];
      const float u0 = ((gw[s1[t] * 256] + gw[(s1[t] + 1) * 256]) * 0.5f) * n10[t];
      const float u1 = ((gw[(s1[t] + 2) * 256] + gw[(s1[t] + 3) * 256]) * 0.5f) * n11[t];
      if (FUSED) vals[(T + t) * GEO_VS + lane] = u0 * a1[t] + u1 * b1[t];
      else if (live) o1[(size_t)(c * T + t) * hw] = u0 * a1[t] + u1 * b1[t];
    }
    multiply();
  }
  fetch_w(C);
  // the two correlation rows (one gather pair per tap each)
#pragma unroll
  for (int t = 0; t < T; ++t) {
    int i0; float w0, w1;
    sample_pos(cx - d + (float)(t - R), W2, i0, w0, w1);
    const float* cr = corr0 + n * W2;
    const float c0 = (i0 >= 0 && i0 < W2) ? cr[i0] : 0.f, c1 = (i0 + 1 >= 0 && i0 + 1 < W2) ? cr[i0 + 1] : 0.f;
    if (FUSED) vals[t * GEO_VS + lane] = c0 * w0 + c1 * w1;
    else if (live) o[(size_t)(C * T + t) * hw] = c0 * w0 + c1 * w1;
    sample_pos(cx / 2.0f - d / 2.0f + (float)(t - R), W2b, i0, w0, w1);
    const float* cr1 = corr1 + n * W2b;
    const float e0 = (i0 >= 0 && i0 < W2b) ? cr1[i0] : 0.f, e1 = (i0 + 1 >= 0 && i0 + 1 < W2b) ? cr1[i0 + 1] : 0.f;
    if (FUSED) vals[(T + t) * GEO_VS + lane] = e0 * w0 + e1 * w1;
    else if (live) o1[(size_t)(C * T + t) * hw] = e0 * w0 + e1 * w1;
  }
  if (FUSED) {
    multiply();
    // acc[mt][nt][i]: cout 16*mt + 4*kq + i of pixel 16*nt + mj of this wave -- 16 consecutive pixels per (cout, store)
    const size_t nw0 = (size_t)blockIdx.x * blockDim.x + wave * 64;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const size_t np = nw0 + nt * 16 + mj;
      if (np >= npix) continue;
      const size_t bb = np / hw, pp = np - bb * hw;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = mt * 16 + kq * 4 + i;
          out[(bb * GEO_NF + co) * hw + pp] = dv_act(acc[mt][nt][i] + (bias ? bias[co] : 0.f), act);
        }
    }
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
  hipLaunchKernelGGL((geo_lookup_kernel<4, false>), dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     geo, corr0, corr1, disp, coords, noisy, nullptr, nullptr, out, C, D, h, w, W2, npix, DV_ACT_NONE);
  return dv_launch_status();
}

extern "C" size_t dv_geo_lookup_conv1x1_packed_floats(int C) { return C > 0 ? (size_t)(C + 1) * GEO_KG * GEO_NF : 0; }

namespace {
// w [64][2*(C*9+9)] (the nn.Conv2d 1x1 weight) -> wk [C+1][20][64]: group c < C holds lookup channels c*9 + t (k = t) and
// half + c*9 + t (k = 9 + t), group C the correlation rows C*9 + t and half + C*9 + t; k = 18, 19 are zero
__global__ void pack_geo_conv_weights_kernel(const float* __restrict__ w, float* __restrict__ wk, int C) {
  const int total = (C + 1) * GEO_KG * GEO_NF;
  const int half = C * 9 + 9, nch = 2 * half;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int co = i % GEO_NF, k = (i / GEO_NF) % GEO_KG, g = i / (GEO_NF * GEO_KG);
    float v = 0.f;
    if (k < 18) {
      const int ch = (k < 9 ? 0 : half) + g * 9 + (k < 9 ? k : k - 9);
      v = w[(size_t)co * nch + ch];
    }
    wk[i] = v;
  }
}
}  // namespace

extern "C" int dv_geo_lookup_conv1x1_pack_weights_f32(const float* w, float* wpacked, int C, dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(C > 0, DV_ERR_SHAPE);
  hipLaunchKernelGGL(pack_geo_conv_weights_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w, wpacked, C);
  return dv_launch_status();
}

extern "C" int dv_geo_filter_lookup_conv1x1_f32(const float* geo, const float* corr0, const float* corr1, const float* disp,
                                                const float* coords, const float* noisy, const float* wpacked,
                                                const float* bias, float* out, int B, int C, int D, int h, int w, int W2,
                                                int radius, int Cout, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(geo);
  DV_REQUIRE_PTR(corr0);
  DV_REQUIRE_PTR(corr1);
  DV_REQUIRE_PTR(disp);
  DV_REQUIRE_PTR(coords);
  DV_REQUIRE_PTR(noisy);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && C > 0 && D > 3 && h > 0 && w > 0 && W2 > 3, DV_ERR_SHAPE);
  DV_REQUIRE(radius == 4 && Cout == GEO_NF, DV_ERR_UNSUPPORTED);      // BasicMotionEncoder.convc1 of every IGEV config
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_TANH, DV_ERR_UNSUPPORTED);
  const size_t npix = (size_t)B * h * w;
  hipLaunchKernelGGL((geo_lookup_kernel<4, true>), dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     geo, corr0, corr1, disp, coords, noisy, wpacked, bias, out, C, D, h, w, W2, npix, act);
  return dv_launch_status();
}
