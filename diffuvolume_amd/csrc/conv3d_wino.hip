// K4w: the 3x3x3 stride-1 aggregation convolution (convbn_3d, SceneFlow/models/submodule.py:94-97; the
// dres / hourglass / classifier layers of acv_ddim.py:60-70, :200-222) with the two in-plane taps done by the
// Winograd minimal-filtering transform F(2x2, 3x3) and the depth taps kept direct:
//   per depth tap kd and input channel c:   M[p] += V[p](x) * U[p](w),  p = 16 transform positions
//   V = Bt d B  of every 4x4 input patch (stride 2),  U = G g Gt  (packed once),  Y = At M A  (2x2 outputs)
// so a 2x2 output tile costs 16 multiplies per (kd, c) instead of 36: 2.25x fewer MFMA flops than the direct
// implicit GEMM of conv3d.hip, still on the exact-fp32 instruction v_mfma_f32_16x16x4_f32.  The transforms
// only add and subtract (Bt, At entries are 0/+-1; G has the 1/2 folded into the packed weights), so the
// rounding is that of a few extra fp32 additions per product.
//
// Block = 4 waves = a 4(z) x 4(y) x 16(x) output brick x 32 output channels; wave w owns plane z0+w: its MFMA
// M index is the 16 tiles (2 tile rows x 8 tile columns) of that plane, N = 16 output channels, K = 4 input
// channels per step.  Per chunk of 4 input channels the haloed raw brick (6 x 6 x 18) goes through LDS, is
// transformed into V[c][plane 0..5][tile][16 positions] (each plane feeds the three waves that see it as
// kd = 0,1,2), and the weights of the chunk are copied next to it; a wave then runs 3 x 16 x 2 MFMAs whose
// A / B fragments are ds_read_b128 of four positions each.  The raw brick of chunk c+1 is written while chunk c
// computes, its global loads are issued a further chunk ahead.

#include <type_traits>

#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace wg {
constexpr int KC = 4, NT = 2, TD = 4, TH = 4, TW = 16;
constexpr int IZ = TD + 2, IY = TH + 2, IX = TW + 2;
constexpr int PRAW = IZ * IY * IX;          // 648 raw positions per channel
constexpr int RAWP = 656;                   // channel stride of the raw brick: = 16 mod 64, so 8 tiles x 4 channels
                                            // of ds_read_b64 cover the 64 banks once
constexpr int TS = 20;                      // tile (and cout) row stride: 16 positions + 4, j*20 mod 64 distinct slots
constexpr int VPL = 16 * TS;                // one plane of V / one k-row block of U
constexpr int VC = IZ * VPL;                // channel stride of V (= 0 mod 64)
constexpr int V_FLOATS = KC * VC;
constexpr int U_FLOATS = 3 * NT * KC * VPL;
constexpr int RAW_FLOATS = KC * RAWP;
constexpr int U_CHUNK = 3 * NT * KC * 16 * 16;   // packed floats per (chunk, co block)
constexpr int NS = (PRAW + 255) / 256;
static_assert((V_FLOATS + U_FLOATS + RAW_FLOATS) * 4 * 2 <= 160 * 1024, "two blocks per CU");
}  // namespace wg

struct WinoArgs {
  const float* in;
  const float* wpk;      // [Cin/4][Coutp/32][kd 3][nt 2][k 4][n 16][pos 16]
  const float* ch_scale;
  const float* ch_bias;
  const float* in_scale; // [B,D,H,W] or null
  const float* residual;
  float* out;
  int B, Cin, D, H, W, Cout;
  int ntx, nty, ntz, nco;
  int act;
  int fast_ok;           // W % 4 == 0, 16-byte aligned pointers
};

template <bool HAS_SCALE>
__global__ __launch_bounds__(256, 2) void conv3d_wino_kernel(WinoArgs a) {
  using namespace wg;
  __shared__ __attribute__((aligned(16))) float smem[V_FLOATS + U_FLOATS + RAW_FLOATS];
  float* v_s = smem;
  float* u_s = smem + V_FLOATS;
  float* raw_s = u_s + U_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;

  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty; t /= a.nty;
  const int tz = t % a.ntz; t /= a.ntz;
  const int tc = t % a.nco;
  const int b = t / a.nco;
  const int x0 = tx * TW, y0 = ty * TH, z0 = tz * TD, co0 = tc * 32;

  f32x4 acc[16][NT];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[p][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const size_t plane = (size_t)a.H * a.W;
  const size_t vol = (size_t)a.D * plane;
  const float* inb = a.in + (size_t)b * a.Cin * vol;
  const float* scb = (HAS_SCALE && a.in_scale) ? a.in_scale + (size_t)b * vol : nullptr;

  // ---- raw staging plan: NS positions of the haloed brick per thread, the same for every channel ----
  unsigned sob[NS];
  unsigned okmask = 0;
  float scl[HAS_SCALE ? NS : 1];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int r = tid + 256 * i;
    const int zz = r / (IY * IX), r2 = r - zz * (IY * IX);
    const int yy = r2 / IX, xx = r2 - yy * IX;
    const int z = z0 - 1 + zz, y = y0 - 1 + yy, x = x0 - 1 + xx;
    const bool ok = r < PRAW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H &&
                    (unsigned)x < (unsigned)a.W;
    const unsigned sp = ok ? (unsigned)((z * a.H + y) * a.W + x) : 0u;
    sob[i] = sp * 4u;
    okmask |= ok ? (1u << i) : 0u;
    if (HAS_SCALE) scl[i] = (ok && scb) ? scb[sp] : 1.f;
  }
  float vin[KC][NS];
  f32x4 vu[6];
  auto fetch_raw = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int cl = 0; cl < KC; ++cl) {
      const int ch = (c0 + cl) < a.Cin ? c0 + cl : 0;
      const char* src = reinterpret_cast<const char*>(inb + (size_t)ch * vol);
#pragma unroll
      for (int i = 0; i < NS; ++i) vin[cl][i] = *reinterpret_cast<const float*>(src + sob[i]);
    }
  };
  auto commit_raw = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int cl = 0; cl < KC; ++cl) {
      const bool cok = (c0 + cl) < a.Cin;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int r = tid + 256 * i;
        const float v = (cok && ((okmask >> i) & 1u)) ? vin[cl][i] : 0.f;
        if (r < PRAW) raw_s[cl * RAWP + r] = HAS_SCALE ? v * scl[i] : v;
      }
    }
  };
  auto fetch_u = [&](int c0) __attribute__((always_inline)) {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.wpk + ((size_t)(c0 >> 2) * a.nco + tc) * U_CHUNK);
#pragma unroll
    for (int q = 0; q < 6; ++q) vu[q] = src[tid + 256 * q];
  };
  auto commit_u = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int e = tid + 256 * q;
      reinterpret_cast<f32x4*>(u_s)[(e >> 2) * 5 + (e & 3)] = vu[q];
    }
  };
  // ---- input transform: 768 units = 4 channels x 6 planes x 16 tiles x 2 halves (two of the four transform
  // rows each); a 32-lane half reads 8 tile columns x 4 channels = one bank row per ds_read_b64 ----
  auto transform = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int u = tid + 256 * i;
      const int ttx = u & 7, cl = (u >> 3) & 3, rest = u >> 5;       // rest 0..23
      const int half = rest >= 12 ? 1 : 0, rr = rest - half * 12;
      const int pl = rr >> 1, tty = rr & 1;
      const float* rp = raw_s + cl * RAWP + (pl * IY + 2 * tty + half) * IX + 2 * ttx;
      f32x2 e[3][2];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        e[k][0] = *reinterpret_cast<const f32x2*>(rp + k * IX);
        e[k][1] = *reinterpret_cast<const f32x2*>(rp + k * IX + 2);
      }
      float r0[4], r1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float e0 = e[0][c >> 1][c & 1], e1 = e[1][c >> 1][c & 1], e2 = e[2][c >> 1][c & 1];
        // half 0: rows d0-d2, d1+d2 (e = d0,d1,d2);  half 1: rows d2-d1, d1-d3 (e = d1,d2,d3)
        r0[c] = half ? e1 - e0 : e0 - e2;
        r1[c] = half ? e0 - e2 : e1 + e2;
      }
      float* vp = v_s + cl * VC + pl * VPL + (tty * 8 + ttx) * TS + half * 8;
      *reinterpret_cast<f32x4*>(vp) = (f32x4){r0[0] - r0[2], r0[1] + r0[2], r0[2] - r0[1], r0[1] - r0[3]};
      *reinterpret_cast<f32x4*>(vp + 4) = (f32x4){r1[0] - r1[2], r1[1] + r1[2], r1[2] - r1[1], r1[1] - r1[3]};
    }
  };

  fetch_raw(0);
  fetch_u(0);
  commit_raw(0);
  if (KC < a.Cin) fetch_raw(KC);
  const float* ap0 = v_s + kq * VC + wave * VPL + j * TS;
  const float* bp0 = u_s + kq * VPL + j * TS;
  for (int c0 = 0; c0 < a.Cin; c0 += KC) {
    __syncthreads();     // the previous chunk's MFMAs are done with V and U; the raw brick of this chunk is complete
    transform();
    commit_u();
    __syncthreads();
    if (c0 + KC < a.Cin) {
      commit_raw(c0 + KC);
      fetch_u(c0 + KC);
      if (c0 + 2 * KC < a.Cin) fetch_raw(c0 + 2 * KC);
    }
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
      const float* ap = ap0 + kd * VPL;
      const float* bp = bp0 + kd * (NT * KC * VPL);
#pragma unroll
      for (int p4 = 0; p4 < 4; ++p4) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + p4 * 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + n * (KC * VPL) + p4 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[p4 * 4 + e][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc[p4 * 4 + e][n], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: Y = At M A per tile, BN scale/bias, residual, activation.  A lane holds output channel j of
  // tiles 4*kq .. 4*kq+3 = tile row kq>>1, tile columns (kq&1)*4 .. +3: 8 consecutive x of two output rows ----
  const int zo = z0 + wave;
  const int yb = y0 + 2 * (kq >> 1), xb = x0 + (kq & 1) * 8;
  const bool fast = a.fast_ok && x0 + TW <= a.W && y0 + TH <= a.H;
  if (zo >= a.D) return;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = co0 + n * 16 + j;
    if (co >= a.Cout) continue;
    const float sc = a.ch_scale ? a.ch_scale[co] : 1.f;
    const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
    float yv[2][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s0[4], s1[4];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const float m0 = acc[px][n][i], m1 = acc[4 + px][n][i], m2 = acc[8 + px][n][i], m3 = acc[12 + px][n][i];
        s0[px] = m0 + m1 + m2;
        s1[px] = m1 - m2 - m3;
      }
      yv[0][2 * i] = s0[0] + s0[1] + s0[2];
      yv[0][2 * i + 1] = s0[1] - s0[2] - s0[3];
      yv[1][2 * i] = s1[0] + s1[1] + s1[2];
      yv[1][2 * i + 1] = s1[1] - s1[2] - s1[3];
    }
    const size_t cbase = (((size_t)b * a.Cout + co) * a.D + zo) * plane;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int yo = yb + r;
      if (yo >= a.H) continue;
      const size_t o = cbase + (size_t)yo * a.W + xb;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(yv[r][e], sc, bi);
      if (fast) {
        if (a.residual) {
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(a.residual + o);
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(a.residual + o + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
        }
        f32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o0[e] = dv_act(v[e], a.act); o1[e] = dv_act(v[4 + e], a.act); }
        *reinterpret_cast<f32x4*>(a.out + o) = o0;
        *reinterpret_cast<f32x4*>(a.out + o + 4) = o1;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (xb + e < a.W) {
            float u = v[e];
            if (a.residual) u += a.residual[o + e];
            a.out[o + e] = dv_act(u, a.act);
          }
      }
    }
  }
}

// U = G g Gt per (cout, cin, kd);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void pack_wino_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout,
                                         int nchunk, int nco) {
  const size_t total = (size_t)nchunk * nco * 3 * 2 * 4 * 16;   // one thread per (chunk, cb, kd, nt, k, n)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int n = (int)(r % 16); r /= 16;
    const int k = (int)(r % 4); r /= 4;
    const int nt = (int)(r % 2); r /= 2;
    const int kd = (int)(r % 3); r /= 3;
    const int cb = (int)(r % nco);
    const int ch = (int)(r / nco);
    const int co = cb * 32 + nt * 16 + n, ci = ch * 4 + k;
    float g[3][3];
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q)
        g[p][q] = (co < Cout && ci < Cin) ? w[(((size_t)co * Cin + ci) * 3 + kd) * 9 + p * 3 + q] : 0.f;
    float gg[4][3];   // G g
    for (int q = 0; q < 3; ++q) {
      gg[0][q] = g[0][q];
      gg[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
      gg[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
      gg[3][q] = g[2][q];
    }
    float* dst = wpk + i * 16;
    for (int p = 0; p < 4; ++p) {
      dst[p * 4 + 0] = gg[p][0];
      dst[p * 4 + 1] = 0.5f * (gg[p][0] + gg[p][1] + gg[p][2]);
      dst[p * 4 + 2] = 0.5f * (gg[p][0] - gg[p][1] + gg[p][2]);
      dst[p * 4 + 3] = gg[p][2];
    }
  }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

extern "C" size_t dv_conv3d_wino_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)cdiv(Cin, 4) * cdiv(Cout, 32) * wg::U_CHUNK;
}

extern "C" int dv_conv3d_wino_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout,
                                               dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  const int nchunk = cdiv(Cin, 4), nco = cdiv(Cout, 32);
  const size_t total = (size_t)nchunk * nco * 3 * 2 * 4 * 16;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_wino_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpacked, Cin,
                     Cout, nchunk, nco);
  return dv_launch_status();
}

extern "C" int dv_conv3d_wino_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                                  const float* in_scale, const float* residual, float* out, int B, int Cin, int D,
                                  int H, int W, int Cout, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE((size_t)D * H * W * sizeof(float) <= 0xffffffffull, DV_ERR_SHAPE);
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_LEAKY, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(dv_aligned16(wpacked), DV_ERR_ALIGN);
  WinoArgs a;
  a.in = in; a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.in_scale = in_scale;
  a.residual = residual; a.out = out;
  a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
  a.fast_ok = (W % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual));
  a.ntx = cdiv(W, wg::TW); a.nty = cdiv(H, wg::TH); a.ntz = cdiv(D, wg::TD); a.nco = cdiv(Cout, 32);
  const long long blocks = (long long)B * a.nco * a.ntz * a.nty * a.ntx;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (in_scale)
    hipLaunchKernelGGL(conv3d_wino_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(conv3d_wino_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return dv_launch_status();
}
