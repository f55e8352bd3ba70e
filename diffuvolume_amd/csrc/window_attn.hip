// K6: 4x4x4 window multi-head self-attention of the hourglass bottleneck.
// Replaces attention_block.forward (SceneFlow/models/submodule.py:398-429): the view/permute
// copies, qkv Linear(128,384)+bias, per-head softmax(q k^T / sqrt(8)) v, the head merge and
// the final 1x1x1 Conv3d(128,128)+bias become ONE kernel, one workgroup per window, with the
// window's 64 tokens x 128 channels resident on chip (registers + LDS) from the first load to the last store.
//
// Token order inside a window is the reference's (permute 0,2,4,6,3,5,7,1): tok = ld*16+lh*4+lw.
// qkv feature f = which*128 + head*8 + dim; merged channel = head*8 + dim.
// Everything runs on v_mfma_f32_16x16x4_f32: the qkv GEMM (M = tokens, N = features, K = channels), the per-head
// 64x64 attention (S^T = K Q^T and P V, chained without a transpose) and the projection (accumulated per head group).
#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 128, HEADS = 16, HD = 8, TOK = 64;
constexpr int LDW = 130;   // W_s[n][k]     stride == 2 (mod 32): banks 2j+kq are distinct

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

// Layout for THREE blocks per CU (51 KB of LDS, < 170 VGPRs).  History: a first version kept the window, 4 heads of
// q|k|v, 96 weight rows and the outputs in 150 KB and ran one block per CU at 1.02 ms; round 1's second version
// (two blocks per CU, the 64x64 attention on the vector ALU split over (head, key half) with a merge phase, the merged
// outputs collected in LDS for four final projection passes) took 0.63 -> 0.565 ms.  Now:
//  * the window's activations live in registers as the A fragments of the qkv GEMM (32 per lane);
//  * heads are processed two at a time (48 qkv weight rows + the 16-column slice of the projection weights in LDS,
//    both fetched into registers one group ahead so their L2 latency hides behind the previous group's work);
//  * the attention itself runs on the matrix cores without a transpose (see below);
//  * the final 1x1x1 projection is accumulated group by group in registers (out += O_g . Wp[:, 16g:16g+16]^T), so the
//    merged attention output never exists as a whole: a wave passes its 16 x 16 slice through a private LDS tile.
namespace v2 {
constexpr int GH2 = 2, GF2 = 3 * GH2 * HD;      // 48 qkv features per group
constexpr int GC = GH2 * HD;                    // 16 merged channels per group
constexpr int LDQ2 = 52;                        // q_s[tok][q 16 | k 16 | v 16]
constexpr int WROWS = 48;                       // qkv weight rows resident at a time
constexpr int LDP = 18;                         // wp_s[co][16 channels of the group]: 2*LDP == 4 (mod 32)
constexpr int LDT = 17;                         // ot_s[tok][16 channels]: wave-private tile
constexpr int W2 = WROWS * LDW, Q2 = TOK * LDQ2, P2 = C * LDP, T2 = TOK * LDT;
constexpr int NWQ = WROWS * 32 / 256, NPQ = C * 4 / 256;      // float4 per thread: qkv rows (6), projection slice (2)
static_assert((W2 + Q2 + P2 + T2) * 4 * 3 <= 160 * 1024, "three blocks per CU");
}  // namespace v2

__global__ __launch_bounds__(256, 3) void window_attn_kernel(AttnArgs a) {
  using namespace v2;
  __shared__ __attribute__((aligned(16))) float smem[W2 + Q2 + P2 + T2];
  float* w_s = smem;
  float* q_s = w_s + W2;
  float* wp_s = q_s + Q2;
  float* ot_s = wp_s + P2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;

  // XCD-aware window order: the 4 floats a window takes from a (channel, d, h) row are 16 bytes of a 128-byte line that
  // the seven windows next to it along w share; consecutive block ids go to different XCDs, i.e. eight L2s fetched
  // (and partially wrote) every line -- 8.5x / 2.3x the algorithmic read / write traffic in the round-2 counters.
  // With a contiguous slab of windows per XCD the neighbours find the line in their own L2.
  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int ww = t % a.nw; t /= a.nw;
  const int wh = t % a.nh; t /= a.nh;
  const int wd = t % a.nd;
  const int b = t / a.nd;
  const int d0 = wd * 4, h0 = wh * 4, w0 = ww * 4;
  const size_t plane = (size_t)a.H * a.W, vol = (size_t)a.D * plane;
  const float* xb = a.x + (size_t)b * C * vol;

  // weights of a head group -> registers (issued a group ahead) -> LDS
  float4 wq[NWQ], wpq[NPQ];
  auto fetch_w = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NWQ; ++i) {
      const int e = tid + 256 * i, r = e >> 5, q = e & 31;              // r = row 0..47 = which*16 + feature
      const int which = r >> 4, f = r & 15;
      wq[i] = reinterpret_cast<const float4*>(a.qkv_w + (size_t)(which * C + g * GC + f) * C)[q];
    }
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      const int e = tid + 256 * i, co = e >> 2, q = e & 3;              // projection weights [co][16g + 4q .. +3]
      wpq[i] = reinterpret_cast<const float4*>(a.proj_w + (size_t)co * C + g * GC)[q];
    }
  };
  auto store_w = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NWQ; ++i) {
      const int e = tid + 256 * i, r = e >> 5, q = e & 31;
      float2* d = reinterpret_cast<float2*>(w_s + r * LDW + 4 * q);
      d[0] = make_float2(wq[i].x, wq[i].y);
      d[1] = make_float2(wq[i].z, wq[i].w);
    }
#pragma unroll
    for (int i = 0; i < NPQ; ++i) {
      const int e = tid + 256 * i, co = e >> 2, q = e & 3;
      float2* d = reinterpret_cast<float2*>(wp_s + co * LDP + 4 * q);
      d[0] = make_float2(wpq[i].x, wpq[i].y);
      d[1] = make_float2(wpq[i].z, wpq[i].w);
    }
  };
  fetch_w(0);

  // A fragments of this wave's 16 tokens (ld = wave, lh = j >> 2, lw = j & 3): channel ks*4 + kq
  float xa[C / 4];
  {
    const int gy = h0 + (j >> 2), gx = w0 + (j & 3);
    const bool in = gy < a.H && gx < a.W;
    const float* src = xb + (size_t)(d0 + wave) * plane + (size_t)(in ? gy : 0) * a.W + (in ? gx : 0) + (size_t)kq * vol;
#pragma unroll
    for (int ks = 0; ks < C / 4; ++ks) xa[ks] = in ? src[(size_t)ks * 4 * vol] : 0.f;
  }

  f32x4 oacc[C / 16];                                  // projection accumulators: 16 tokens x 128 output channels
#pragma unroll
  for (int n = 0; n < C / 16; ++n) oacc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float* ot = ot_s + wave * 16 * LDT;                  // this wave's 16 x 16 tile

  for (int g = 0; g < HEADS / GH2; ++g) {
    __syncthreads();                                   // the previous group is done with w_s / wp_s / q_s
    store_w();
    __syncthreads();
    if (g + 1 < HEADS / GH2) fetch_w(g + 1);           // in flight during this group's GEMM and attention
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
        const float bias = a.qkv_b[n * C + g * GC + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) q_s[(wave * 16 + 4 * kq + r) * LDQ2 + f] = acc[n][r] + bias;
      }
    }
    __syncthreads();
    {   // attention on the matrix cores: wave = one tile of 16 queries, both heads of the group in turn.
        // S^T = K Q^T is computed with the KEYS as the MFMA M index, so that a lane (query j, k-group kq) ends up with
        // the scores of its query against keys 16*kt + 4*kq + i -- which is exactly the A-operand layout of the
        // P V product when its k-steps are taken as (kt, i): no transpose, no LDS round trip, no cross-wave merge.
      const int qrow = (wave * 16 + j) * LDQ2;
      const float scale = 0.35355339059327379f;  // 8^-0.5
#pragma unroll
      for (int hd = 0; hd < GH2; ++hd) {
        const int hoff = hd * HD;
        f32x4 sc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const float* kp = q_s + (kt * 16 + j) * LDQ2 + GC + hoff + kq;             // A: key row j, dim 4s + kq
          const float* qp = q_s + qrow + hoff + kq;                                  // B: query column j, dim 4s + kq
#pragma unroll
          for (int s2 = 0; s2 < HD / 4; ++s2)
            sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[4 * s2], qp[4 * s2], sc[kt], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = sc[kt][i] * scale;
            if (a.mask_on) {
              const int k = kt * 16 + 4 * kq + i, q = wave * 16 + j;
              const bool k_pad = (h0 + ((k >> 2) & 3) >= a.H) || (w0 + (k & 3) >= a.W);
              const bool qp_ = (h0 + ((q >> 2) & 3) >= a.H) || (w0 + (q & 3) >= a.W);
              if (k_pad != qp_) v += -1000.0f;
            }
            sc[kt][i] = v;
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            sc[kt][i] = dv_exp_le0(sc[kt][i] - mx);     // the compensated exp2 of dv_common.h, ~1 ulp like expf
            sum += sc[kt][i];
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        // O = P V: M = this wave's queries, N = the head's 8 dims (columns 8..15 of the tile are idle), K = keys
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        const float* vp = q_s + 2 * GC + hoff + (j & 7);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float bv = vp[(kt * 16 + 4 * kq + i) * LDQ2];                     // B: V[key 16kt+4kq+i][dim j]
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[kt][i], bv, o, 0, 0, 0);
          }
        // o[i] = unnormalised output of query 16*wave + 4*kq + i, dim j; the row sums live in the lanes of their queries
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float rs = __shfl(sum, 4 * kq + i);
          if (j < HD) ot[(4 * kq + i) * LDT + hoff + j] = o[i] / rs;
        }
      }
      // projection of this group's 16 merged channels, accumulated: oacc[tok][co] += O_g[tok][c] * Wp[co][16g + c].
      // The tile is this wave's own (a wave's LDS operations execute in order), so no block barrier is needed.
      const float* ap = ot + j * LDT + kq;
      const float* bp = wp_s + j * LDP + kq;
#pragma unroll
      for (int s2 = 0; s2 < GC / 4; ++s2) {
        const float av = ap[4 * s2];
#pragma unroll
        for (int n = 0; n < C / 16; ++n)
          oacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[n * 16 * LDP + 4 * s2], oacc[n], 0, 0, 0);
      }
    }
  }

  // ---- store: out[co][token] = oacc + bias; lane (co j, kq) holds the 4 consecutive w of row h0 + kq ----
  const size_t ob = (size_t)b * C * vol;
  const bool vec = (a.W % 4 == 0) && ((((uintptr_t)a.out) & 15u) == 0);
  const int gy = h0 + kq;
  if (gy < a.H) {
#pragma unroll
    for (int n = 0; n < C / 16; ++n) {
      const int co = n * 16 + j;
      const float bias = a.proj_b[co];
      float* dst = a.out + ob + (size_t)co * vol + (size_t)(d0 + wave) * plane + (size_t)gy * a.W + w0;
      if (vec) {
        *reinterpret_cast<float4*>(dst) =
            make_float4(oacc[n][0] + bias, oacc[n][1] + bias, oacc[n][2] + bias, oacc[n][3] + bias);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (w0 + r < a.W) dst[r] = oacc[n][r] + bias;
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
