// K6: 4x4x4 window multi-head self-attention of the hourglass bottleneck.
// Replaces attention_block.forward (SceneFlow/models/submodule.py:398-429): the view/permute
// copies, qkv Linear(128,384)+bias, per-head softmax(q k^T / sqrt(8)) v, the head merge and
// the final 1x1x1 Conv3d(128,128)+bias become ONE kernel, one workgroup per window, with the
// window's 64 tokens x 128 channels resident in LDS from the first load to the last store.
//
// Token order inside a window is the reference's (permute 0,2,4,6,3,5,7,1): tok = ld*16+lh*4+lw.
// qkv feature f = which*128 + head*8 + dim; merged channel = head*8 + dim.
// The two GEMMs run on v_mfma_f32_16x16x4_f32 (M = tokens, N = features, K = channels), the
// 64x64 per-head attention on the vector ALU (lane = query, wave = head; keys broadcast from LDS).
#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 128, HEADS = 16, HD = 8, TOK = 64;
constexpr int LDX = 80;    // X_s[c][tok]   stride == 16 (mod 32): the 4 k-lanes hit disjoint banks
constexpr int LDW = 130;   // W_s[n][k]     stride == 2 (mod 32): banks 2j+kq are distinct
constexpr int LDO = 130;   // O_s[tok][c]
constexpr int LDQ = 100;   // Q_s[tok][96]  (q|k|v of 4 heads); 4*LDQ == 16 (mod 32)
constexpr int GH = 4;      // heads per group
constexpr int GF = 3 * GH * HD;  // 96 features per group
constexpr int X_FLOATS = C * LDX, O_FLOATS = TOK * LDO, W_FLOATS = GF * LDW, Q_FLOATS = TOK * LDQ;

struct AttnArgs {
  const float* x;
  const float* qkv_w;   // [384][128]
  const float* qkv_b;   // [384]
  const float* proj_w;  // [128][128]
  const float* proj_b;  // [128]
  float* out;
  int B, D, H, W, nd, nh, nw;
  int mask_on;          // both H and W are padded (reference quirk, submodule.py:414-416)
};

// rows [row0, row0+nrows) of a [*,128] row-major matrix -> W_s[n][k] (stride LDW)
__device__ __forceinline__ void stage_rows(float* w_s, int dst_row, const float* src, int nrows, int tid) {
  for (int e = tid; e < nrows * 32; e += 256) {
    const int r = e >> 5, q = e & 31;
    const float4 v = reinterpret_cast<const float4*>(src + (size_t)r * C)[q];
    float2* d = reinterpret_cast<float2*>(w_s + (dst_row + r) * LDW + 4 * q);
    d[0] = make_float2(v.x, v.y);
    d[1] = make_float2(v.z, v.w);
  }
}

__global__ __launch_bounds__(256) void window_attn_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[X_FLOATS + O_FLOATS + W_FLOATS + Q_FLOATS];
  float* x_s = smem;
  float* o_s = x_s + X_FLOATS;
  float* w_s = o_s + O_FLOATS;
  float* q_s = w_s + W_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;

  int t = blockIdx.x;
  const int ww = t % a.nw; t /= a.nw;
  const int wh = t % a.nh; t /= a.nh;
  const int wd = t % a.nd;
  const int b = t / a.nd;
  const int d0 = wd * 4, h0 = wh * 4, w0 = ww * 4;
  const size_t plane = (size_t)a.H * a.W, vol = (size_t)a.D * plane;
  const float* xb = a.x + (size_t)b * C * vol;
  const bool vec = (a.W % 4 == 0) && ((((uintptr_t)a.x) & 15u) == 0);

  // ---- load the window: x_s[c][tok], zero for padded tokens ----
  for (int e = tid; e < C * 16; e += 256) {
    const int c = e >> 4, row = e & 15, ld = row >> 2, lh = row & 3;
    const int gy = h0 + lh;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gy < a.H) {
      const float* src = xb + (size_t)c * vol + (size_t)(d0 + ld) * plane + (size_t)gy * a.W + w0;
      if (vec) {
        const float4 q = *reinterpret_cast<const float4*>(src);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
        for (int i = 0; i < 4; ++i)
          if (w0 + i < a.W) v[i] = src[i];
      }
    }
    *reinterpret_cast<float4*>(x_s + c * LDX + row * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }

  // per-token pad flag of this lane's query (only used when mask_on)
  const int q_ld = lane >> 4, q_lh = (lane >> 2) & 3, q_lw = lane & 3;
  (void)q_ld;
  const bool q_pad = a.mask_on && ((h0 + q_lh >= a.H) || (w0 + q_lw >= a.W));

  for (int g = 0; g < HEADS / GH; ++g) {
    __syncthreads();  // x_s ready / previous group done with w_s, q_s
    // ---- stage the q|k|v weight rows of heads [4g, 4g+4) ----
    for (int which = 0; which < 3; ++which)
      stage_rows(w_s, which * GH * HD, a.qkv_w + (size_t)(which * C + g * GH * HD) * C, GH * HD, tid);
    __syncthreads();
    // ---- GEMM1: q_s[tok][f] = x[tok][:] . Wg[f][:] + b   (wave = 16-token tile) ----
    {
      f32x4 acc[GF / 16];
#pragma unroll
      for (int n = 0; n < GF / 16; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* ap = x_s + kq * LDX + wave * 16 + j;
      const float* bp = w_s + j * LDW + kq;
#pragma unroll 4
      for (int ks = 0; ks < C / 4; ++ks) {
        const float av = ap[ks * 4 * LDX];
#pragma unroll
        for (int n = 0; n < GF / 16; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[n * 16 * LDW + ks * 4], acc[n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < GF / 16; ++n) {
        const int f = n * 16 + j;                         // feature within the group block
        const int which = f / (GH * HD), hf = f % (GH * HD);
        const float bias = a.qkv_b[which * C + g * GH * HD + hf];
#pragma unroll
        for (int r = 0; r < 4; ++r) q_s[(wave * 16 + 4 * kq + r) * LDQ + f] = acc[n][r] + bias;
      }
    }
    __syncthreads();
    // ---- attention of head 4g+wave: lane = query token ----
    {
      const int hoff = wave * HD;
      const float4 qa = *reinterpret_cast<const float4*>(q_s + lane * LDQ + hoff);
      const float4 qb = *reinterpret_cast<const float4*>(q_s + lane * LDQ + hoff + 4);
      const float scale = 0.35355339059327379f;  // 8^-0.5
      float sc[TOK];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < TOK; ++k) {
        const float4 ka = *reinterpret_cast<const float4*>(q_s + k * LDQ + GH * HD + hoff);
        const float4 kb = *reinterpret_cast<const float4*>(q_s + k * LDQ + GH * HD + hoff + 4);
        float s = qa.x * ka.x;
        s = fmaf(qa.y, ka.y, s); s = fmaf(qa.z, ka.z, s); s = fmaf(qa.w, ka.w, s);
        s = fmaf(qb.x, kb.x, s); s = fmaf(qb.y, kb.y, s); s = fmaf(qb.z, kb.z, s); s = fmaf(qb.w, kb.w, s);
        s *= scale;
        if (a.mask_on) {
          const bool k_pad = (h0 + ((k >> 2) & 3) >= a.H) || (w0 + (k & 3) >= a.W);
          if (k_pad != q_pad) s += -1000.0f;
        }
        sc[k] = s;
        mx = fmaxf(mx, s);
      }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < TOK; ++k) {
        sc[k] = expf(sc[k] - mx);
        sum += sc[k];
      }
      float o[HD] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < TOK; ++k) {
        const float p = sc[k] / sum;
        const float4 va = *reinterpret_cast<const float4*>(q_s + k * LDQ + 2 * GH * HD + hoff);
        const float4 vb = *reinterpret_cast<const float4*>(q_s + k * LDQ + 2 * GH * HD + hoff + 4);
        o[0] = fmaf(p, va.x, o[0]); o[1] = fmaf(p, va.y, o[1]); o[2] = fmaf(p, va.z, o[2]); o[3] = fmaf(p, va.w, o[3]);
        o[4] = fmaf(p, vb.x, o[4]); o[5] = fmaf(p, vb.y, o[5]); o[6] = fmaf(p, vb.z, o[6]); o[7] = fmaf(p, vb.w, o[7]);
      }
      float2* dst = reinterpret_cast<float2*>(o_s + lane * LDO + (g * GH + wave) * HD);
      dst[0] = make_float2(o[0], o[1]); dst[1] = make_float2(o[2], o[3]);
      dst[2] = make_float2(o[4], o[5]); dst[3] = make_float2(o[6], o[7]);
    }
  }

  // ---- final 1x1x1 conv: out[tok][co] = o[tok][:] . Wp[co][:] + b, 64 output channels per pass ----
  const size_t ob = (size_t)b * C * vol;
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    stage_rows(w_s, 0, a.proj_w + (size_t)half * 64 * C, 64, tid);
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ap = o_s + (wave * 16 + j) * LDO + kq;
    const float* bp = w_s + j * LDW + kq;
#pragma unroll 4
    for (int ks = 0; ks < C / 4; ++ks) {
      const float av = ap[ks * 4];
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[n * 16 * LDW + ks * 4], acc[n], 0, 0, 0);
    }
    // lane holds tokens (ld = wave, lh = kq, lw = 0..3) of channel co
    const int gy = h0 + kq;
    if (gy < a.H) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int co = half * 64 + n * 16 + j;
        const float bias = a.proj_b[co];
        float* dst = a.out + ob + (size_t)co * vol + (size_t)(d0 + wave) * plane + (size_t)gy * a.W + w0;
        if (vec && ((((uintptr_t)a.out) & 15u) == 0)) {
          *reinterpret_cast<float4*>(dst) =
              make_float4(acc[n][0] + bias, acc[n][1] + bias, acc[n][2] + bias, acc[n][3] + bias);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (w0 + r < a.W) dst[r] = acc[n][r] + bias;
        }
      }
    }
  }
}

}  // namespace

extern "C" int dv_window_attn3d_f32(const float* x, const float* qkv_w, const float* qkv_b,
                                    const float* proj_w, const float* proj_b, float* out, int B, int Cc,
                                    int D, int H, int W, int heads, dv_stream_t stream) {
  DV_REQUIRE_PTR(x);
  DV_REQUIRE_PTR(qkv_w);
  DV_REQUIRE_PTR(qkv_b);
  DV_REQUIRE_PTR(proj_w);
  DV_REQUIRE_PTR(proj_b);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, DV_ERR_SHAPE);
  DV_REQUIRE(Cc == C && heads == HEADS, DV_ERR_UNSUPPORTED);  // every reference use: 128 ch, 16 heads
  DV_REQUIRE(D % 4 == 0, DV_ERR_SHAPE);  // the reference only pads H and W; D%4!=0 fails its view()
  DV_REQUIRE(dv_aligned16(qkv_w) && dv_aligned16(proj_w), DV_ERR_ALIGN);
  AttnArgs a;
  a.x = x; a.qkv_w = qkv_w; a.qkv_b = qkv_b; a.proj_w = proj_w; a.proj_b = proj_b; a.out = out;
  a.B = B; a.D = D; a.H = H; a.W = W;
  a.nd = D / 4; a.nh = (H + 3) / 4; a.nw = (W + 3) / 4;
  a.mask_on = (H % 4 != 0) && (W % 4 != 0);
  const long long blocks = (long long)B * a.nd * a.nh * a.nw;
  if (blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  hipLaunchKernelGGL(window_attn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return dv_launch_status();
}
