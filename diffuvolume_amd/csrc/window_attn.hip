// K6: 4x4x4 window multi-head self-attention of the hourglass bottleneck.
// Replaces attention_block.forward (SceneFlow/models/submodule.py:398-429): the view/permute
// copies, qkv Linear(128,384)+bias, per-head softmax(q k^T / sqrt(8)) v, the head merge and
// the final 1x1x1 Conv3d(128,128)+bias become ONE kernel, one workgroup per window, with the
// window's 64 tokens x 128 channels resident on chip (registers + LDS) from the first load to the last store.
//
// Token order inside a window is the reference's (permute 0,2,4,6,3,5,7,1): tok = ld*16+lh*4+lw.
// qkv feature f = which*128 + head*8 + dim; merged channel = head*8 + dim.
// The two GEMMs run on v_mfma_f32_16x16x4_f32 (M = tokens, N = features, K = channels), the
// 64x64 per-head attention on the vector ALU (lane = query, wave = head; keys broadcast from LDS).
#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 128, HEADS = 16, HD = 8, TOK = 64;
constexpr int LDW = 130;   // W_s[n][k]     stride == 2 (mod 32): banks 2j+kq are distinct
constexpr int LDO = 130;   // O_s[tok][c]

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

// Layout for two blocks per CU (72 KB of LDS; a first version kept the window, 4 heads of q|k|v, 96 weight rows and
// the outputs in 150 KB and ran one block per CU at 1.02 ms -- this one takes 0.63 ms): the window's activations
// live in registers as the A fragments of the qkv GEMM (32 per lane), heads are processed two at a time (48 weight
// rows, 13 KB of q|k|v), and all four waves work on the 2 x 64 x 64 attention: wave = (head, key half), the two
// halves merged with the running-max rule.  Two resident blocks overlap each other's staging, MFMA and vector
// phases.
namespace v2 {
constexpr int GH2 = 2, GF2 = 3 * GH2 * HD;      // 48 features per group
constexpr int LDQ2 = 52;                        // 4*LDQ2 == 16 (mod 32)
constexpr int WROWS = 48;                       // weight rows resident at a time (also >= 32 for the projection)
constexpr int PSTRIDE = 12;                     // partial record: m, s, o[8] (+pad)
constexpr int O2 = TOK * LDO, W2 = WROWS * LDW, Q2 = TOK * LDQ2;
static_assert(2 * 2 * TOK * PSTRIDE <= W2, "partials reuse the weight region");
static_assert((O2 + W2 + Q2) * 4 <= 80 * 1024, "two blocks per CU");
}  // namespace v2

__global__ __launch_bounds__(256, 2) void window_attn_kernel(AttnArgs a) {
  using namespace v2;
  __shared__ __attribute__((aligned(16))) float smem[O2 + W2 + Q2];
  float* o_s = smem;
  float* w_s = o_s + O2;
  float* q_s = w_s + W2;
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

  // A fragments of this wave's 16 tokens (ld = wave, lh = j >> 2, lw = j & 3): channel ks*4 + kq
  float xa[C / 4];
  {
    const int gy = h0 + (j >> 2), gx = w0 + (j & 3);
    const bool in = gy < a.H && gx < a.W;
    const float* src = xb + (size_t)(d0 + wave) * plane + (size_t)(in ? gy : 0) * a.W + (in ? gx : 0) + (size_t)kq * vol;
#pragma unroll
    for (int ks = 0; ks < C / 4; ++ks) xa[ks] = in ? src[(size_t)ks * 4 * vol] : 0.f;
  }
  const int q_lh = (lane >> 2) & 3, q_lw = lane & 3;
  const bool q_pad = a.mask_on && ((h0 + q_lh >= a.H) || (w0 + q_lw >= a.W));

  for (int g = 0; g < HEADS / GH2; ++g) {
    __syncthreads();                                   // previous group's partials / q_s are consumed
    for (int which = 0; which < 3; ++which)
      stage_rows(w_s, which * GH2 * HD, a.qkv_w + (size_t)(which * C + g * GH2 * HD) * C, GH2 * HD, tid);
    __syncthreads();
    {   // GEMM1: q_s[tok][f] = x[tok][:] . Wg[f][:] + b
      f32x4 acc[GF2 / 16];
#pragma unroll
      for (int n = 0; n < GF2 / 16; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* bp = w_s + j * LDW + kq;
#pragma unroll
      for (int ks = 0; ks < C / 4; ++ks)
#pragma unroll
        for (int n = 0; n < GF2 / 16; ++n)
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks], bp[n * 16 * LDW + ks * 4], acc[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < GF2 / 16; ++n) {
        const int f = n * 16 + j;                       // which = n, feature within the 2-head block = j
        const float bias = a.qkv_b[n * C + g * GH2 * HD + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) q_s[(wave * 16 + 4 * kq + r) * LDQ2 + f] = acc[n][r] + bias;
      }
    }
    __syncthreads();
    {   // attention: wave = (head, key half), lane = query
      const int head = wave >> 1, kh = wave & 1, hoff = head * HD;
      const float4 qa = *reinterpret_cast<const float4*>(q_s + lane * LDQ2 + hoff);
      const float4 qb = *reinterpret_cast<const float4*>(q_s + lane * LDQ2 + hoff + 4);
      const float scale = 0.35355339059327379f;  // 8^-0.5
      float sc[TOK / 2];
      float mx = -INFINITY;
#pragma unroll
      for (int kk = 0; kk < TOK / 2; ++kk) {
        const int k = kh * (TOK / 2) + kk;
        const float4 ka = *reinterpret_cast<const float4*>(q_s + k * LDQ2 + GH2 * HD + hoff);
        const float4 kb = *reinterpret_cast<const float4*>(q_s + k * LDQ2 + GH2 * HD + hoff + 4);
        float s = qa.x * ka.x;
        s = fmaf(qa.y, ka.y, s); s = fmaf(qa.z, ka.z, s); s = fmaf(qa.w, ka.w, s);
        s = fmaf(qb.x, kb.x, s); s = fmaf(qb.y, kb.y, s); s = fmaf(qb.z, kb.z, s); s = fmaf(qb.w, kb.w, s);
        s *= scale;
        if (a.mask_on) {
          const bool k_pad = (h0 + ((k >> 2) & 3) >= a.H) || (w0 + (k & 3) >= a.W);
          if (k_pad != q_pad) s += -1000.0f;
        }
        sc[kk] = s;
        mx = fmaxf(mx, s);
      }
      float sum = 0.f;
      float o[HD] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < TOK / 2; ++kk) {
        const int k = kh * (TOK / 2) + kk;
        const float p = dv_exp_le0(sc[kk] - mx);      // sc <= mx: the compensated exp2 of dv_common.h, ~1 ulp like expf
        sum += p;
        const float4 va = *reinterpret_cast<const float4*>(q_s + k * LDQ2 + 2 * GH2 * HD + hoff);
        const float4 vb = *reinterpret_cast<const float4*>(q_s + k * LDQ2 + 2 * GH2 * HD + hoff + 4);
        o[0] = fmaf(p, va.x, o[0]); o[1] = fmaf(p, va.y, o[1]); o[2] = fmaf(p, va.z, o[2]); o[3] = fmaf(p, va.w, o[3]);
        o[4] = fmaf(p, vb.x, o[4]); o[5] = fmaf(p, vb.y, o[5]); o[6] = fmaf(p, vb.z, o[6]); o[7] = fmaf(p, vb.w, o[7]);
      }
      // partial record of (head, key half, query) in the (now idle) weight region
      float* pr = w_s + ((head * 2 + kh) * TOK + lane) * PSTRIDE;
      pr[0] = mx; pr[1] = sum;
#pragma unroll
      for (int i = 0; i < HD; ++i) pr[2 + i] = o[i];
    }
    __syncthreads();
    {   // merge the two key halves: thread = (head, query, 4 of the 8 dims)
      const int head = tid >> 7, q = (tid & 127) >> 1, hd = (tid & 1) * 4;
      const float* p0 = w_s + ((head * 2 + 0) * TOK + q) * PSTRIDE;
      const float* p1 = w_s + ((head * 2 + 1) * TOK + q) * PSTRIDE;
      const float m = fmaxf(p0[0], p1[0]);
      const float e0 = dv_exp_le0(p0[0] - m), e1 = dv_exp_le0(p1[0] - m);
      const float inv = 1.f / (p0[1] * e0 + p1[1] * e1);
      float* dst = o_s + q * LDO + (g * GH2 + head) * HD + hd;
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = (p0[2 + hd + i] * e0 + p1[2 + hd + i] * e1) * inv;
    }
  }

  // ---- final 1x1x1 conv, 32 output channels per pass ----
  const size_t ob = (size_t)b * C * vol;
  const bool vec = (a.W % 4 == 0) && ((((uintptr_t)a.out) & 15u) == 0);
  for (int pass = 0; pass < 4; ++pass) {
    __syncthreads();
    stage_rows(w_s, 0, a.proj_w + (size_t)pass * 32 * C, 32, tid);
    __syncthreads();
    f32x4 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ap = o_s + (wave * 16 + j) * LDO + kq;
    const float* bp = w_s + j * LDW + kq;
#pragma unroll 8
    for (int ks = 0; ks < C / 4; ++ks) {
      const float av = ap[ks * 4];
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[n * 16 * LDW + ks * 4], acc[n], 0, 0, 0);
    }
    const int gy = h0 + kq;
    if (gy < a.H) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int co = pass * 32 + n * 16 + j;
        const float bias = a.proj_b[co];
        float* dst = a.out + ob + (size_t)co * vol + (size_t)(d0 + wave) * plane + (size_t)gy * a.W + w0;
        if (vec) {
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
