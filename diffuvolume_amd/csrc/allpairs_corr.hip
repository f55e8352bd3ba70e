// K10a: IGEV all-pairs correlation along the epipolar line and its level-1 pooling, once per stereo pair:
//   corr0[b,y,x1,x2] = sum_c f1[b,c,y,x1] * f2[b,c,y,x2]          (Combined_Geo_Encoding_Volume.corr,
//   corr1[b,y,x1,x2'] = (corr0[..,2x2'] + corr0[..,2x2'+1]) / 2     KITTI15/core/geometry_ddim.py:72-80 and :28-30)
// One small GEMM per image row (M = W1, N = W2, K = C) on v_mfma_f32_16x16x4_f32.  The op is bound by the write
// of its result (B*h*W1*W2 floats: 150 MB for 4 pairs at 312x96), so the kernel keeps no LDS: a block owns a
// 16-wide strip of x1, its A fragments (f1) stay in registers, the f2 row (C*W2 floats, shared by the 20 strips of
// the row) comes from L2, and the pooled level is formed from the accumulators (lane pairs hold adjacent x2) instead
// of re-reading corr0.
#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KSTEPS_MAX = 64;     // A fragments kept in registers: C <= 256

// KS = MFMA k-steps (4 channels each), rounded up to the instantiation; channels past C contribute zeros
template <int KS>
__global__ __launch_bounds__(256) void allpairs_corr_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                            float* __restrict__ corr0, float* __restrict__ corr1,
                                                            int C, int H, int W1, int W2, int ntx) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  unsigned t = blockIdx.x;
  const int tx = t % ntx; t /= ntx;
  const int y = t % H;
  const int b = t / H;
  const int x1_0 = tx * 16;
  const size_t plane1 = (size_t)H * W1, plane2 = (size_t)H * W2;
  const float* a_row = f1 + ((size_t)b * C * H + y) * W1;          // + c * plane1 + x1
  const float* b_row = f2 + ((size_t)b * C * H + y) * W2;          // + c * plane2 + x2
  const bool a_ok = x1_0 + j < W1;
  float areg[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int c = 4 * s + kq;
    areg[s] = (a_ok && c < C) ? a_row[(size_t)c * plane1 + x1_0 + j] : 0.f;
  }
  const int W2h = W2 / 2;
  const int ntn = (W2 + 15) / 16;
  for (int tn = wave; tn < ntn; tn += 4) {
    const int x2 = tn * 16 + j;
    const bool b_ok = x2 < W2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = 4 * s + kq;
      const float bv = (b_ok && c < C) ? b_row[(size_t)c * plane2 + x2] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[s], bv, acc, 0, 0, 0);
    }
    // acc[i] = corr0[x1 = x1_0 + 4*kq + i][x2 = tn*16 + j]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x1 = x1_0 + 4 * kq + i;
      const float v = acc[i];
      const float nb = __shfl_xor(v, 1);                      // the neighbour column x2 ^ 1
      if (x1 < W1) {
        float* o0 = corr0 + (((size_t)b * H + y) * W1 + x1) * W2;
        if (b_ok) o0[x2] = v;
        if (!(j & 1) && (x2 >> 1) < W2h) corr1[(((size_t)b * H + y) * W1 + x1) * W2h + (x2 >> 1)] = (v + nb) * 0.5f;
      }
    }
  }
}

}  // namespace

extern "C" int dv_allpairs_corr_f32(const float* fmap1, const float* fmap2, float* corr0, float* corr1, int B, int C,
                                    int H, int W1, int W2, dv_stream_t stream) {
  DV_REQUIRE_PTR(fmap1);
  DV_REQUIRE_PTR(fmap2);
  DV_REQUIRE_PTR(corr0);
  DV_REQUIRE_PTR(corr1);
  DV_REQUIRE(B > 0 && C > 0 && H > 0 && W1 > 0 && W2 > 1, DV_ERR_SHAPE);
  DV_REQUIRE(C <= 4 * KSTEPS_MAX, DV_ERR_UNSUPPORTED);
  const int ntx = (W1 + 15) / 16;
  const long long blocks = (long long)B * H * ntx;
  DV_REQUIRE(blocks <= 0x7fffffffLL, DV_ERR_SHAPE);
  const int ks = (C + 3) / 4;
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DV_APC(KS) hipLaunchKernelGGL(allpairs_corr_kernel<KS>, grid, block, 0, s, fmap1, fmap2, corr0, corr1, C, H, W1, W2, ntx)
  if (ks <= 8) DV_APC(8);
  else if (ks <= 16) DV_APC(16);
  else if (ks <= 24) DV_APC(24);
  else if (ks <= 32) DV_APC(32);
  else if (ks <= 48) DV_APC(48);
  else DV_APC(64);
#undef DV_APC
  return dv_launch_status();
}
