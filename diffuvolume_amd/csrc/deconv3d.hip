// K5: ConvTranspose3d(k=3, stride=2, padding=1, output_padding=1, bias=False) as a
// parity-decomposed implicit GEMM on v_mfma_f32_16x16x4_f32, with BatchNorm (eval) scale /
// bias, skip-connection add and activation fused in the epilogue.
// Replaces conv5 / conv6 of the hourglass (SceneFlow/models/acv_ddim.py:74-80, :91-92).
//
// out[o] += in[i] * w[k] with o = 2i - 1 + k per dimension, so per dimension
//   k=1 -> even outputs (o = 2i),  k=2 -> odd outputs (o = 2i+1),  k=0 -> odd outputs (o = 2i-1).
// Every one of the 27 taps therefore feeds exactly one of the 8 output parity classes and reads
// the input at offset (k==0 ? +1 : 0): the kernel walks the 27 taps like the forward
// convolution, but each tap accumulates into its class's accumulator.  A block owns a
// 2 x 2 x 32 brick of INPUT positions (= 4 x 4 x 64 outputs) and 32 output channels; each wave
// one input row, all 8 classes.  No zero-stuffing, no wasted MFMA work.
#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTD = 2, kTH = 2, kMTX = 2, kNT = 2;
constexpr int kTW = kMTX * 16;
constexpr int kCOUT = kNT * 16;

// K = 3: ConvTranspose3d(3, stride 2, padding 1, output_padding 1)   (ACV / PCW hourglass)
// K = 4: ConvTranspose3d(4, stride 2, padding 1)                      (IGEV hourglass, igev_stereo_ddim.py:44-51)
// Both double every dimension.  Per dimension o = 2i - 1 + k:  parity(k) = (k+1)&1, input offset (k==0) - (k==3).
template <int K, int KC_>
struct DGeo {
  static constexpr int KC = KC_;
  static constexpr int LO = (K == 4) ? 1 : 0;                 // halo below (k == 3 reads i-1)
  static constexpr int T = K * K * K;
  static constexpr int IZ = kTD + 1 + LO, IY = kTH + 1 + LO, IX = kTW + 1 + LO;
  static constexpr int PRAW = IZ * IY * IX;
  static constexpr int P = PRAW + ((16 - PRAW % 32) + 32) % 32;   // == 16 (mod 32)
  static constexpr int IN_FLOATS = KC * P, W_FLOATS = T * KC * kCOUT;
  static_assert(P % 32 == 16, "bank rule");
  static_assert((IN_FLOATS + W_FLOATS) * 4 <= 80 * 1024, "two blocks per CU");
};
__host__ __device__ constexpr int tap_par(int k) { return (k + 1) & 1; }
__host__ __device__ constexpr int tap_off(int k) { return (k == 0 ? 1 : 0) - (k == 3 ? 1 : 0); }

struct DeconvArgs {
  const float* in;
  const float* wpk;  // [Cinp/2][K^3][Coutp][2]
  const float* ch_scale;
  const float* ch_bias;
  const float* residual;  // [B,Cout,2D,2H,2W] or null
  float* out;
  int B, Cin, D, H, W, Cout, Coutp;
  int ntx, nty, ntz, nco;
  int act;
  int vec_store;
};

template <int K, int KC_>
__global__ __launch_bounds__(256, 2) void deconv3d_mfma_kernel(DeconvArgs a) {
  using G = DGeo<K, KC_>;
  constexpr int kKC = G::KC, kT = G::T, kIY = G::IY, kIX = G::IX, kPRAW = G::PRAW, kP = G::P, LO = G::LO;
  __shared__ __attribute__((aligned(16))) float smem[G::IN_FLOATS + G::W_FLOATS];
  float* in_s = smem;
  float* w_s = smem + G::IN_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;

  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty; t /= a.nty;
  const int tz = t % a.ntz; t /= a.ntz;
  const int tc = t % a.nco;
  const int b = t / a.nco;
  const int x0 = tx * kTW, y0 = ty * kTH, z0 = tz * kTD, co0 = tc * kCOUT;
  const int zl = wave / kTH, yl = wave % kTH;   // this wave's input row

  f32x4 acc[kMTX][8][kNT];
#pragma unroll
  for (int m = 0; m < kMTX; ++m)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int n = 0; n < kNT; ++n) acc[m][c][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int abase = kq * kP + ((zl + LO) * kIY + yl + LO) * kIX + j + LO;
  const int bbase = ((kq >> 1) * kT * kCOUT + j) * 2 + (kq & 1);
  const size_t plane = (size_t)a.H * a.W, vol = (size_t)a.D * plane;
  const float* inb = a.in + (size_t)b * a.Cin * vol;

  // staging plan (same scheme as conv3d.hip): each thread owns NS positions of the haloed input
  // brick; the next chunk's global loads are issued before the current chunk's MFMA stream
  constexpr int NS = (kPRAW + 255) / 256;
  constexpr int ROWQ = kCOUT * 2 / 4;
  constexpr int NQ = (kKC / 2) * kT * ROWQ;
  constexpr int NWQ = (NQ + 255) / 256;
  int sp[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int r = tid + 256 * i;
    const int zz = r / (kIY * kIX), r2 = r - zz * (kIY * kIX);
    const int yy = r2 / kIX, xx = r2 - yy * kIX;
    const int z = z0 - LO + zz, y = y0 - LO + yy, x = x0 - LO + xx;
    sp[i] = (r < kPRAW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
                ? (z * a.H + y) * a.W + x : -1;
  }
  float vin[kKC][NS];
  f32x4 vw[NWQ];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int cl = 0; cl < kKC; ++cl) {
      const float* src = inb + (size_t)(c0 + cl) * vol;
      const bool cok = (c0 + cl) < a.Cin;
#pragma unroll
      for (int i = 0; i < NS; ++i) vin[cl][i] = (cok && sp[i] >= 0) ? src[sp[i]] : 0.f;
    }
    const float* wsrc = a.wpk + ((size_t)(c0 >> 1) * kT * a.Coutp + co0) * 2;
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = tid + 256 * q;
      const int row = e / ROWQ, qq = e - row * ROWQ;
      if (e < NQ) vw[q] = reinterpret_cast<const f32x4*>(wsrc + (size_t)row * a.Coutp * 2)[qq];
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int cl = 0; cl < kKC; ++cl)
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int r = tid + 256 * i;
        if (r < kPRAW) in_s[cl * kP + r] = vin[cl][i];
      }
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = tid + 256 * q;
      if (e < NQ) reinterpret_cast<f32x4*>(w_s)[e] = vw[q];
    }
  };

  fetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += kKC) {
    __syncthreads();
    commit();
    __syncthreads();
    if (c0 + kKC < a.Cin) fetch(c0 + kKC);
#pragma unroll
    for (int kzy = 0; kzy < K * K; ++kzy) {
      const int kz = kzy / K, ky = kzy - kz * K;
      const float* arow = in_s + abase + (tap_off(kz) * kIY + tap_off(ky)) * kIX;
      const float* brow = w_s + bbase + kzy * K * kCOUT * 2;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int cls = (tap_par(kz) << 2) | (tap_par(ky) << 1) | tap_par(kx);
#pragma unroll
        for (int ks = 0; ks < kKC / 4; ++ks) {
          float bf[kNT];
#pragma unroll
          for (int n = 0; n < kNT; ++n) bf[n] = brow[((ks * 2 * kT + kx) * kCOUT + n * 16) * 2];
#pragma unroll
          for (int m = 0; m < kMTX; ++m) {
            const float av = arow[m * 16 + ks * 4 * kP + tap_off(kx)];
#pragma unroll
            for (int n = 0; n < kNT; ++n)
              acc[m][cls][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf[n], acc[m][cls][n], 0, 0, 0);
          }
        }
      }
    }
  }

  // epilogue: px=0 / px=1 classes interleave along x -> 8 consecutive outputs per lane
  const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
  const int zi = z0 + zl, yi = y0 + yl;
  if (zi >= a.D || yi >= a.H) return;
#pragma unroll
  for (int n = 0; n < kNT; ++n) {
    const int co = co0 + n * 16 + j;
    if (co >= a.Cout) continue;
    const float sc = a.ch_scale ? a.ch_scale[co] : 1.f;
    const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
    const size_t cbase = ((size_t)b * a.Cout + co) * ovol;
#pragma unroll
    for (int m = 0; m < kMTX; ++m) {
      const int xi = x0 + m * 16 + 4 * kq;
      if (xi >= a.W) continue;
#pragma unroll
      for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const size_t o = cbase + (size_t)(2 * zi + pz) * oplane + (size_t)(2 * yi + py) * Wo + 2 * xi;
          const f32x4 e0 = acc[m][(pz << 2) | (py << 1)][n], e1 = acc[m][(pz << 2) | (py << 1) | 1][n];
          float v[8] = {e0[0], e1[0], e0[1], e1[1], e0[2], e1[2], e0[3], e1[3]};
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = fmaf(v[r], sc, bi);
          if (a.vec_store) {  // W % 4 == 0 here, so all 8 outputs are in range
            if (a.residual) {
              const float4 r0 = *reinterpret_cast<const float4*>(a.residual + o);
              const float4 r1 = *reinterpret_cast<const float4*>(a.residual + o + 4);
              v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
              v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            }
            *reinterpret_cast<float4*>(a.out + o) = make_float4(
                dv_act(v[0], a.act), dv_act(v[1], a.act), dv_act(v[2], a.act), dv_act(v[3], a.act));
            *reinterpret_cast<float4*>(a.out + o + 4) = make_float4(
                dv_act(v[4], a.act), dv_act(v[5], a.act), dv_act(v[6], a.act), dv_act(v[7], a.act));
          } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
              if (2 * xi + r < Wo) {
                float u = v[r];
                if (a.residual) u += a.residual[o + r];
                a.out[o + r] = dv_act(u, a.act);
              }
          }
        }
    }
  }
}

__global__ void pack_deconv_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin,
                                           int Cout, int Cinp, int Coutp, int kT) {
  const size_t total = (size_t)Cinp * kT * Coutp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int par = (int)(i & 1);
    size_t r = i >> 1;
    const int co = (int)(r % Coutp); r /= Coutp;
    const int tap = (int)(r % kT);
    const int ci = (int)(r / kT) * 2 + par;
    wpk[i] = (ci < Cin && co < Cout) ? w[((size_t)ci * Cout + co) * kT + tap] : 0.f;
  }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

template <int K, int KC_>
int launch_deconv(DeconvArgs a, hipStream_t s) {
  const long long blocks = (long long)a.B * a.nco * a.ntz * a.nty * a.ntx;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  hipLaunchKernelGGL((deconv3d_mfma_kernel<K, KC_>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  return dv_launch_status();
}

int pack_any(const float* w, float* wpacked, int Cin, int Cout, int K, hipStream_t s) {
  const int T = K * K * K, Cinp = pad_to(Cin, 8), Coutp = pad_to(Cout, kCOUT);
  const size_t total = (size_t)Cinp * T * Coutp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_deconv_weights_kernel, dim3(blocks), dim3(256), 0, s, w, wpacked, Cin, Cout, Cinp, Coutp, T);
  return dv_launch_status();
}

int run_any(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
            const float* residual, float* out, int B, int Cin, int D, int H, int W, int Cout, int act, int K,
            hipStream_t s) {
  DeconvArgs a;
  a.in = in; a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.residual = residual;
  a.out = out; a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout;
  a.Coutp = pad_to(Cout, kCOUT);
  a.ntx = (W + kTW - 1) / kTW;
  a.nty = (H + kTH - 1) / kTH;
  a.ntz = (D + kTD - 1) / kTD;
  a.nco = a.Coutp / kCOUT;
  a.act = act;
  a.vec_store = (W % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual));
  return K == 3 ? launch_deconv<3, 8>(a, s) : launch_deconv<4, 4>(a, s);
}

}  // namespace

extern "C" size_t dv_deconv3d_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)pad_to(Cin, 8) * 27 * pad_to(Cout, kCOUT);
}

extern "C" size_t dv_deconv3d_k4_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)pad_to(Cin, 8) * 64 * pad_to(Cout, kCOUT);
}

extern "C" int dv_deconv3d_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout, dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  return pack_any(w, wpacked, Cin, Cout, 3, (hipStream_t)stream);
}

extern "C" int dv_deconv3d_k4_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout, dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  return pack_any(w, wpacked, Cin, Cout, 4, (hipStream_t)stream);
}

#define DV_DECONV_CHECKS                                                             \
  DV_REQUIRE_PTR(in);                                                                \
  DV_REQUIRE_PTR(wpacked);                                                           \
  DV_REQUIRE_PTR(out);                                                               \
  DV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);  \
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_LEAKY, DV_ERR_UNSUPPORTED);         \
  DV_REQUIRE(dv_aligned16(wpacked), DV_ERR_ALIGN);

extern "C" int dv_deconv3d_k3s2_f32(const float* in, const float* wpacked, const float* ch_scale,
                                    const float* ch_bias, const float* residual, float* out, int B,
                                    int Cin, int D, int H, int W, int Cout, int act, dv_stream_t stream) {
  DV_DECONV_CHECKS
  return run_any(in, wpacked, ch_scale, ch_bias, residual, out, B, Cin, D, H, W, Cout, act, 3, (hipStream_t)stream);
}

extern "C" int dv_deconv3d_k4s2_f32(const float* in, const float* wpacked, const float* ch_scale,
                                    const float* ch_bias, const float* residual, float* out, int B,
                                    int Cin, int D, int H, int W, int Cout, int act, dv_stream_t stream) {
  DV_DECONV_CHECKS
  return run_any(in, wpacked, ch_scale, ch_bias, residual, out, B, Cin, D, H, W, Cout, act, 4, (hipStream_t)stream);
}
