// K2: concatenation cost volume, optionally multiplied by softmax_D(attention logits).
// Replaces build_concat_volume (SceneFlow/models/submodule.py:180-191; KITTI12 flavour
// :86-97 with the left half zeroed for x<d) and the `F.softmax(att, dim=2) * volume`
// pass of acv_ddim.py:390 (two extra passes over a 3 GB tensor at batch 8).
//
//   out[b,c,d,y,x]   = p[b,d,y,x] * ref[b,c,y,x]                     c <  C
//   out[b,C+c,d,y,x] = p[b,d,y,x] * (x>=d ? tgt[b,c,y,x-d] : 0)      c <  C
//
// Pure HBM write stream.  A block owns FOUR consecutive rows (b, y0..y0+3) and a chunk of 8 channels of both
// halves; wave <-> one of the rows, lane <-> 4 consecutive x (16-byte stores, contiguous along W), so that the four
// waves of a block write a contiguous 4-row slab (3.8 KB at W = 240) of every (channel, disparity) plane at about
// the same time (round 1 had one row per block, waves on different disparities: 960-byte bursts scattered 122 KB
// apart, 3.6 TB/s) and every wave computes the softmax normaliser of its own row only.  The shifted target comes
// from LDS (rows staged once behind a zero pad) with two ds_read_b128 per 4x4 register tile.
// ATT: 0 = plain concatenation, 1 = attention LOGITS (softmax computed here), 2 = attention PROBABILITIES (the softmax
// was taken by dv_softmax_d_f32 -- same arithmetic, same bits -- and is only multiplied in).  The channel chunks of
// one 4-row slab are neighbours in the XCD-aware block order (dv_xcd_remap), so the attention rows they all read come
// from HBM once and from that XCD's L2 afterwards; the left feature values of a chunk are loaded once per thread.
#include "dv_common.h"

namespace {

constexpr int kChunk = 8;  // channels of each half per block

template <int ATT, bool ZERO_LEFT>
__global__ __launch_bounds__(256) void concat_rows_kernel(const float* __restrict__ ref,
                                                          const float* __restrict__ tgt,
                                                          const float* __restrict__ att,
                                                          float* __restrict__ out, int C, int H,
                                                          int W, int D) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int nq = W >> 2;
  const int eblocks = (D + 3) >> 2;
  const int padq = eblocks + 1;
  const int rowq = padq + nq;
  float4* rows = reinterpret_cast<float4*>(smem);

  const int nchunks = (C + kChunk - 1) / kChunk;
  const int nyt = (H + 3) >> 2;
  const unsigned bid = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int cchunk = bid % nchunks;
  const int by = bid / nchunks;
  const int y0 = (by % nyt) * 4;
  const int b = by / nyt;
  const int c0 = cchunk * kChunk;
  const int nc = min(kChunk, C - c0);
  const size_t plane = (size_t)H * W;

  for (int i = threadIdx.x; i < 4 * nc * rowq; i += blockDim.x) {      // rows[(row * nc + c) * rowq + q]
    const int rc = i / rowq, q = i - rc * rowq;
    const int rw = rc / nc, c = rc - rw * nc;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q >= padq && y0 + rw < H)
      v = reinterpret_cast<const float4*>(tgt + ((size_t)b * C + c0 + c) * plane + (size_t)(y0 + rw) * W)[q - padq];
    rows[i] = v;
  }
  __syncthreads();
  const int y = y0 + wave;
  if (y >= H) return;
  rows += (size_t)wave * nc * rowq;

  const size_t vstride = (size_t)D * plane;  // channel stride of the volume
  const float* attrow = ATT ? att + (size_t)b * D * plane + (size_t)y * W : nullptr;
  float* outL = out + ((size_t)b * 2 * C + c0) * vstride + (size_t)y * W;
  float* outR = out + ((size_t)b * 2 * C + C + c0) * vstride + (size_t)y * W;

  for (int t = lane; t < nq; t += DV_WAVE) {
    float4 mx = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f);
    if (ATT == 1) {
      mx = reinterpret_cast<const float4*>(attrow)[t];
      for (int d = 1; d < D; ++d) {
        const float4 a = reinterpret_cast<const float4*>(attrow + (size_t)d * plane)[t];
        mx.x = fmaxf(mx.x, a.x); mx.y = fmaxf(mx.y, a.y); mx.z = fmaxf(mx.z, a.z); mx.w = fmaxf(mx.w, a.w);
      }
      // exp(a - max) on the compensated exp2 of dv_common.h (~1 ulp, like expf); the softmax is then one reciprocal
      // per pixel, p_d = e_d * (1/sum): at most one ulp of p_d from e_d / sum
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int d = 0; d < D; ++d) {
        const float4 a = reinterpret_cast<const float4*>(attrow + (size_t)d * plane)[t];
        sum.x += dv_exp_le0(a.x - mx.x); sum.y += dv_exp_le0(a.y - mx.y);
        sum.z += dv_exp_le0(a.z - mx.z); sum.w += dv_exp_le0(a.w - mx.w);
      }
      rs = make_float4(1.f / sum.x, 1.f / sum.y, 1.f / sum.z, 1.f / sum.w);
    }
    float l[kChunk][4];
#pragma unroll
    for (int c = 0; c < kChunk; ++c) {
      const float4 Lq = c < nc ? reinterpret_cast<const float4*>(ref + ((size_t)b * C + c0 + c) * plane + (size_t)y * W)[t]
                               : make_float4(0.f, 0.f, 0.f, 0.f);
      l[c][0] = Lq.x; l[c][1] = Lq.y; l[c][2] = Lq.z; l[c][3] = Lq.w;
    }
    for (int e = 0; e < eblocks; ++e) {
      float p[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 4 * e + r;
        if (ATT == 1 && d < D) {
          const float4 a = reinterpret_cast<const float4*>(attrow + (size_t)d * plane)[t];
          p[r][0] = dv_exp_le0(a.x - mx.x) * rs.x; p[r][1] = dv_exp_le0(a.y - mx.y) * rs.y;
          p[r][2] = dv_exp_le0(a.z - mx.z) * rs.z; p[r][3] = dv_exp_le0(a.w - mx.w) * rs.w;
        } else if (ATT == 2 && d < D) {
          const float4 a = reinterpret_cast<const float4*>(attrow + (size_t)d * plane)[t];
          p[r][0] = a.x; p[r][1] = a.y; p[r][2] = a.z; p[r][3] = a.w;
        } else {
          p[r][0] = p[r][1] = p[r][2] = p[r][3] = 1.f;
        }
      }
#pragma unroll
      for (int c = 0; c < kChunk; ++c) {
        if (c >= nc) break;
        const float4 lo = rows[c * rowq + padq + t - e - 1];
        const float4 hi = rows[c * rowq + padq + t - e];
        const float w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 4 * e + r;
          if (d < D) {
            float vl[4], vr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float lv = l[c][j];
              if (ZERO_LEFT && (4 * t + j) < d) lv = 0.f;
              vl[j] = ATT ? p[r][j] * lv : lv;
              vr[j] = ATT ? p[r][j] * w[4 + j - r] : w[4 + j - r];
            }
            reinterpret_cast<float4*>(outL + (size_t)c * vstride + (size_t)d * plane)[t] =
                make_float4(vl[0], vl[1], vl[2], vl[3]);
            reinterpret_cast<float4*>(outR + (size_t)c * vstride + (size_t)d * plane)[t] =
                make_float4(vr[0], vr[1], vr[2], vr[3]);
          }
        }
      }
    }
  }
}

// Any W / alignment: one thread per output element.
template <int ATT, bool ZERO_LEFT>
__global__ void concat_generic_kernel(const float* __restrict__ ref, const float* __restrict__ tgt,
                                      const float* __restrict__ att, float* __restrict__ out, int C,
                                      int H, int W, int D, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    size_t r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int d = (int)(r % D);
    r /= D;
    const int c = (int)(r % (2 * C));
    const int b = (int)(r / (2 * C));
    float v;
    if (c < C) {
      v = (ZERO_LEFT && x < d) ? 0.f : ref[(((size_t)b * C + c) * H + y) * W + x];
    } else {
      v = x >= d ? tgt[(((size_t)b * C + (c - C)) * H + y) * W + x - d] : 0.f;
    }
    if (ATT == 2) {
      v *= att[(((size_t)b * D + d) * H + y) * W + x];
    } else if (ATT == 1) {
      const float* a = att + ((size_t)b * D * H + y) * W + x;
      const size_t plane = (size_t)H * W;
      float mx = a[0];
      for (int k = 1; k < D; ++k) mx = fmaxf(mx, a[k * plane]);
      float sum = 0.f;
      for (int k = 0; k < D; ++k) sum += dv_exp_le0(a[k * plane] - mx);
      v *= dv_exp_le0(a[d * plane] - mx) * (1.f / sum);
    }
    out[i] = v;
  }
}

template <int ATT, bool ZL>
int launch(const float* ref, const float* tgt, const float* att, float* out, int B, int C, int H,
           int W, int D, hipStream_t s) {
  const bool fast = (W % 4 == 0) && dv_aligned16(ref) && dv_aligned16(tgt) && dv_aligned16(out) &&
                    (!ATT || dv_aligned16(att));
  const int rowq = ((D + 3) / 4 + 1) + W / 4;
  const size_t lds = (size_t)4 * kChunk * rowq * sizeof(float4);
  if (fast && lds <= 64 * 1024) {
    const int nchunks = (C + kChunk - 1) / kChunk;
    hipLaunchKernelGGL((concat_rows_kernel<ATT, ZL>), dim3(B * ((H + 3) / 4) * nchunks), dim3(256), lds, s, ref,
                       tgt, att, out, C, H, W, D);
  } else {
    const size_t total = (size_t)B * 2 * C * D * H * W;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL((concat_generic_kernel<ATT, ZL>), dim3(blocks), dim3(256), 0, s, ref, tgt, att,
                       out, C, H, W, D, total);
  }
  return dv_launch_status();
}

}  // namespace

extern "C" int dv_concat_volume_f32(const float* ref, const float* tgt, float* out, int B, int C,
                                    int H, int W, int D, int zero_left, dv_stream_t stream) {
  DV_REQUIRE_PTR(ref);
  DV_REQUIRE_PTR(tgt);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0, DV_ERR_SHAPE);
  hipStream_t s = (hipStream_t)stream;
  return zero_left ? launch<0, true>(ref, tgt, nullptr, out, B, C, H, W, D, s)
                   : launch<0, false>(ref, tgt, nullptr, out, B, C, H, W, D, s);
}

extern "C" int dv_concat_attn_volume_f32(const float* ref, const float* tgt, const float* att,
                                         float* out, int B, int C, int H, int W, int D,
                                         dv_stream_t stream) {
  DV_REQUIRE_PTR(ref);
  DV_REQUIRE_PTR(tgt);
  DV_REQUIRE_PTR(att);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0, DV_ERR_SHAPE);
  return launch<1, false>(ref, tgt, att, out, B, C, H, W, D, (hipStream_t)stream);
}

extern "C" int dv_concat_prob_volume_f32(const float* ref, const float* tgt, const float* p, float* out, int B, int C,
                                         int H, int W, int D, dv_stream_t stream) {
  DV_REQUIRE_PTR(ref);
  DV_REQUIRE_PTR(tgt);
  DV_REQUIRE_PTR(p);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0, DV_ERR_SHAPE);
  return launch<2, false>(ref, tgt, p, out, B, C, H, W, D, (hipStream_t)stream);
}
