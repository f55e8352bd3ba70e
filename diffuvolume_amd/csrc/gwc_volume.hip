// K1: group-wise correlation cost volume.
// Replaces the 48-iteration Python loop of build_gwc_volume / groupwise_correlation
// (SceneFlow/models/submodule.py:228-238, :209-215).
//
//   out[b,g,d,y,x] = (1/cpg) * sum_{c<cpg} ref[b,g*cpg+c,y,x] * tgt[b,g*cpg+c,y,x-d]   (0 for x<d)
//
// HBM-bound (read 2*C*H*W, write G*D*H*W floats per pair; ~3 flop/byte).  One wave
// owns one (b,g,y) row: the target row of the group is staged once in LDS behind a
// zero pad of D floats, the reference row lives in registers, and every lane
// produces a 4(x) x 4(d) register tile per step from ONE ds_read_b128 per channel
// (sliding 8-float window), so LDS traffic is 1/16 of the naive scheme and all
// global loads/stores are 16 B per lane, contiguous along W.
#include "dv_common.h"

namespace {

constexpr int kRowsPerBlock = 4;  // one wave per row

template <int CPG>
__global__ __launch_bounds__(256) void gwc_rows_kernel(const float* __restrict__ ref,
                                                       const float* __restrict__ tgt,
                                                       float* __restrict__ out, int C, int H, int W,
                                                       int D, int G, int total_rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int nq = W >> 2;                       // float4 quads per row
  const int eblocks = (D + 3) >> 2;            // blocks of 4 disparities
  const int padq = eblocks + 1;                // zero quads left of x=0
  const int rowq = padq + nq;                  // quads per staged channel row
  float4* rows = reinterpret_cast<float4*>(smem) + (size_t)wave * CPG * rowq;

  const int row = blockIdx.x * (blockDim.x >> 6) + wave;   // (b*G + g)*H + y; one wave per row
  const bool live = row < total_rows;
  const int y = live ? row % H : 0;
  const int bg = live ? row / H : 0;
  const int g = bg % G;
  const int b = bg / G;
  const size_t plane = (size_t)H * W;
  const float* refrow = ref + ((size_t)b * C + (size_t)g * CPG) * plane + (size_t)y * W;
  const float* tgtrow = tgt + ((size_t)b * C + (size_t)g * CPG) * plane + (size_t)y * W;

  // stage the target rows of this group (zero pad on the left)
  if (live) {
    for (int c = 0; c < CPG; ++c) {
      float4* dst = rows + c * rowq;
      for (int q = lane; q < padq; q += DV_WAVE) dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* src = reinterpret_cast<const float4*>(tgtrow + (size_t)c * plane);
      for (int q = lane; q < nq; q += DV_WAVE) dst[padq + q] = src[q];
    }
  }
  __syncthreads();
  if (!live) return;

  float* outrow = out + (((size_t)b * G + g) * D * H + y) * (size_t)W;  // d = 0
  const size_t dstride = plane;
  const float inv = 1.0f / (float)CPG;
  const bool pow2 = (CPG & (CPG - 1)) == 0;

  for (int t = lane; t < nq; t += DV_WAVE) {
    float4 L[CPG], hi[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
      L[c] = reinterpret_cast<const float4*>(refrow + (size_t)c * plane)[t];
      hi[c] = rows[c * rowq + padq + t];
    }
    for (int e = 0; e < eblocks; ++e) {
      float acc[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;
#pragma unroll
      for (int c = 0; c < CPG; ++c) {
        const float4 lo = rows[c * rowq + padq + t - e - 1];
        const float w[8] = {lo.x, lo.y, lo.z, lo.w, hi[c].x, hi[c].y, hi[c].z, hi[c].w};
        const float l[4] = {L[c].x, L[c].y, L[c].z, L[c].w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j)  // product rounded, then added: same rounding as (f1*f2).sum()
            acc[r][j] = __fadd_rn(acc[r][j], __fmul_rn(l[j], w[4 + j - r]));
        hi[c] = lo;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 4 * e + r;
        if (d < D) {
          float4 v;
          if (pow2) {
            v = make_float4(acc[r][0] * inv, acc[r][1] * inv, acc[r][2] * inv, acc[r][3] * inv);
          } else {
            v = make_float4(acc[r][0] / CPG, acc[r][1] / CPG, acc[r][2] / CPG, acc[r][3] / CPG);
          }
          reinterpret_cast<float4*>(outrow + (size_t)d * dstride)[t] = v;
        }
      }
    }
  }
}

// Any shape (W not a multiple of 4, unusual channels-per-group): one thread per output.
__global__ void gwc_generic_kernel(const float* __restrict__ ref, const float* __restrict__ tgt,
                                   float* __restrict__ out, int C, int H, int W, int D, int G,
                                   size_t total) {
  const int cpg = C / G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    size_t r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int d = (int)(r % D);
    r /= D;
    const int g = (int)(r % G);
    const int b = (int)(r / G);
    float acc = 0.f;
    if (x >= d) {
      const size_t base = (((size_t)b * C + (size_t)g * cpg) * H + y) * W + x;
      for (int c = 0; c < cpg; ++c)
        acc = __fadd_rn(acc, __fmul_rn(ref[base + (size_t)c * H * W], tgt[base + (size_t)c * H * W - d]));
      acc = acc / (float)cpg;
    }
    out[i] = acc;
  }
}

template <int CPG>
int launch_rows(const float* ref, const float* tgt, float* out, int B, int C, int H, int W, int D,
                int G, hipStream_t s) {
  const int total_rows = B * G * H;
  const int rowq = ((D + 3) / 4 + 1) + W / 4;
  int rpb = kRowsPerBlock;                       // rows (= waves) per block: as many as fit 64 KB of LDS
  while (rpb > 1 && (size_t)rpb * CPG * rowq * sizeof(float4) > 64 * 1024) rpb >>= 1;
  const size_t lds = (size_t)rpb * CPG * rowq * sizeof(float4);
  if (lds > 64 * 1024) return DV_ERR_UNSUPPORTED;
  const int blocks = (total_rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(gwc_rows_kernel<CPG>, dim3(blocks), dim3(64 * rpb), lds, s, ref, tgt, out, C, H, W,
                     D, G, total_rows);
  return dv_launch_status();
}

}  // namespace

extern "C" int dv_gwc_volume_f32(const float* ref, const float* tgt, float* out, int B, int C, int H,
                                 int W, int D, int G, dv_stream_t stream) {
  DV_REQUIRE_PTR(ref);
  DV_REQUIRE_PTR(tgt);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0 && G > 0, DV_ERR_SHAPE);
  DV_REQUIRE(C % G == 0, DV_ERR_SHAPE);  // the reference asserts this (submodule.py:211)
  hipStream_t s = (hipStream_t)stream;
  const int cpg = C / G;
  const bool fast = (W % 4 == 0) && dv_aligned16(ref) && dv_aligned16(tgt) && dv_aligned16(out);
  if (fast) {
    int rc = DV_ERR_UNSUPPORTED;
    if (cpg == 8) rc = launch_rows<8>(ref, tgt, out, B, C, H, W, D, G, s);
    else if (cpg == 12) rc = launch_rows<12>(ref, tgt, out, B, C, H, W, D, G, s);
    else if (cpg == 4) rc = launch_rows<4>(ref, tgt, out, B, C, H, W, D, G, s);
    if (rc != DV_ERR_UNSUPPORTED) return rc;
  }
  const size_t total = (size_t)B * G * D * H * W;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(gwc_generic_kernel, dim3(blocks), dim3(256), 0, s, ref, tgt, out, C, H, W, D, G,
                     total);
  return dv_launch_status();
}
