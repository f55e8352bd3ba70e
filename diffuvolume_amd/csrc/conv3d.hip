// K4: 3-D convolution (k in {1,3}, stride in {1,2}, pad (k-1)/2) as an implicit GEMM on the
// exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32, with the reference's surrounding
// elementwise work fused in:
//   y = act( conv(x * in_scale) * ch_scale + ch_bias + residual )
// Replaces convbn_3d + ReLU (SceneFlow/models/submodule.py:94-97; uses acv_ddim.py:60-70,
// :82-83, :200-222) and the `volume * noise.unsqueeze(1)` multiply of acv_ddim.py:260.
//
// GEMM view:  M = output voxels (16 consecutive x of one (z,y) row per MFMA tile),
//             N = output channels (16 per tile),  K = taps x input channels.
// A block of 4 waves owns a TD x TH x (16*MTX) output brick and NT*16 output channels.
// Per chunk of KC input channels the haloed input brick and the weight slice are staged
// in LDS (channel-plane stride P chosen so the four k-lanes of an MFMA operand hit
// disjoint banks), then every tap is one MFMA k-step:  A = in_s[cin][voxel+tap]
// (ds_read_b32, lanes contiguous along x), B = w_s[tap][cin][cout].  Several blocks are
// resident per CU so one block's staging overlaps another's MFMA stream.
// MFMA-bound: 27*Cin*2 flop per output float (AI 86-864 flop/byte at fp32).

#include <type_traits>

#include <atomic>
#include <cstdlib>

#include "dv_common.h"

namespace {

// test hook: 0 = the launcher picks the stride-2 tiling, 1 = 2 x 4 x 32 tiles, 2 = 2 x 2 x 32 tiles (dv_conv3d_set_s2_tile)
std::atomic<int> g_s2_tile_pin{0};
// test hook (dv_conv3d_set_c1z): threshold of the z-marching single-channel head in tiles per batch item (default 64) and
// a pinned segment length (0 = the launcher's choice; 3 / 6 / 12)
std::atomic<long long> g_c1z_min_tiles{64};
std::atomic<int> g_c1z_pin_zs{0};

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int round_mod(int v, int mod, int rem) {  // smallest v' >= v with v' % mod == rem
  int r = v % mod;
  return r <= rem ? v + (rem - r) : v + (mod - r) + rem;
}

template <int KS_, int S_, int NT_, int MTX_, int TH_, int TD_, int KC_, int WPS_>
struct Geo {
  static constexpr int KS = KS_, S = S_, NT = NT_, MTX = MTX_, TH = TH_, TD = TD_, KC = KC_;
  static constexpr int WPS = WPS_;  // resident waves per SIMD the register budget is sized for
  static constexpr int PAD = (KS - 1) / 2;
  static constexpr int T = KS * KS * KS;
  static constexpr int TW = MTX * 16;
  static constexpr int IZ = (TD - 1) * S + KS, IY = (TH - 1) * S + KS, IX = (TW - 1) * S + KS;
  static constexpr int PRAW = IZ * IY * IX;
  // bank rule: lanes 0-15 (k=0) and 16-31 (k=1) of a ds_read_b32 must not collide
  static constexpr int P = S == 1 ? round_mod(PRAW, 32, 16) : (PRAW | 1);
  static constexpr int COUT = NT * 16;
  static constexpr int IN_FLOATS = KC * P;
  static constexpr int W_FLOATS = T * KC * COUT;
  static constexpr int ROWS = TD * TH;
  static constexpr int RPW = ROWS / 4;
  static constexpr int MT = RPW * MTX;
  static_assert(ROWS % 4 == 0, "rows must split over 4 waves");
  static_assert(KC % 4 == 0, "K chunk is a multiple of the MFMA k");
  static_assert((IN_FLOATS + W_FLOATS) * 4 <= 160 * 1024, "LDS budget");
};

struct ConvArgs {
  const float* in;
  const float* wpk;      // [Cinp/2][T][Coutp][2]
  const float* ch_scale; // [Cout] or null
  const float* ch_bias;  // [Cout] or null
  const float* in_scale; // [B,D,H,W] or null
  const float* residual; // [B,Cout,Do,Ho,Wo] or null
  float* out;
  int B, Cin, D, H, W, Cout, Coutp, Do, Ho, Wo;
  int ntx, nty, ntz, nco;  // tile counts
  int act;
  int order;               // tile walk order: 0 = x,y,z  1 = z,y,x  2 = z,x,y (fastest first)
  int vec_store;           // Wo % 4 == 0 and 16-byte aligned pointers
  int fast_ok;             // vec_store and 32-bit byte offsets inside one batch item of the output
};

template <class G, bool HAS_SCALE>
__global__ __launch_bounds__(256, G::WPS) void conv3d_mfma_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[G::IN_FLOATS + G::W_FLOATS];
  float* in_s = smem;
  float* w_s = smem + G::IN_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: row bases stay in SGPRs
  const int j = lane & 15, kq = lane >> 4;

  // block -> (b, co-slice, z, y, x) tile; consecutive tiles on one XCD share halos in its L2
  // (the output-channel slices of a tile are neighbours in the linear order: same XCD, same time, one HBM read of the brick)
  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int tc = t % a.nco; t /= a.nco;
  int tx, ty, tz;
  if (a.order == 1) {
    tz = t % a.ntz; t /= a.ntz;
    ty = t % a.nty; t /= a.nty;
    tx = t % a.ntx; t /= a.ntx;
  } else if (a.order == 2) {
    tz = t % a.ntz; t /= a.ntz;
    tx = t % a.ntx; t /= a.ntx;
    ty = t % a.nty; t /= a.nty;
  } else {
    tx = t % a.ntx; t /= a.ntx;
    ty = t % a.nty; t /= a.nty;
    tz = t % a.ntz; t /= a.ntz;
  }
  const int b = t;
  const int x0 = tx * G::TW, y0 = ty * G::TH, z0 = tz * G::TD, co0 = tc * G::COUT;
  const int xi0 = x0 * G::S - G::PAD, yi0 = y0 * G::S - G::PAD, zi0 = z0 * G::S - G::PAD;

  f32x4 acc[G::MT][G::NT];
#pragma unroll
  for (int m = 0; m < G::MT; ++m)
#pragma unroll
    for (int n = 0; n < G::NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int abase[G::MT];
#pragma unroll
  for (int m = 0; m < G::MT; ++m) {
    const int rr = wave * G::RPW + m / G::MTX, xt = m % G::MTX;
    const int zl = rr / G::TH, yl = rr % G::TH;
    abase[m] = kq * G::P + ((zl * G::S) * G::IY + yl * G::S) * G::IX + (xt * 16 + j) * G::S;
  }
  const int bbase = ((kq >> 1) * G::T * G::COUT + j) * 2 + (kq & 1);

  const size_t plane = (size_t)a.H * a.W;
  const size_t vol = (size_t)a.D * plane;
  const float* inb = a.in + (size_t)b * a.Cin * vol;
  const float* scb = (HAS_SCALE && a.in_scale) ? a.in_scale + (size_t)b * vol : nullptr;

  // ---- staging plan: every thread owns NS spatial positions of the haloed brick (the same for
  // every channel), so the index algebra and the bounds checks are done once per block ----
  constexpr int NS = (G::PRAW + 255) / 256;
  constexpr int ROWQ = G::COUT * 2 / 4;                   // float4 per (cinpair, tap) weight row
  constexpr int NQ = (G::KC / 2) * G::T * ROWQ;           // float4 per weight chunk
  constexpr int NWQ = (NQ + 255) / 256;
  unsigned sob[NS];                 // byte offset inside one channel volume (0 where the brick leaves the volume)
  float scl[HAS_SCALE ? NS : 1];    // the `volume * noise` prologue factor of that position
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int r = tid + 256 * i;
    const int zz = r / (G::IY * G::IX), r2 = r - zz * (G::IY * G::IX);
    const int yy = r2 / G::IX, xx = r2 - yy * G::IX;
    const int z = zi0 + zz, y = yi0 + yy, x = xi0 + xx;
    const bool ok = r < G::PRAW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H &&
                    (unsigned)x < (unsigned)a.W;
    const unsigned sp = ok ? (unsigned)((z * a.H + y) * a.W + x) : 0u;
    sob[i] = ok ? sp * 4u : 0x80000000u;     // outside the volume: beyond the buffer's records -> the load returns 0
    if (HAS_SCALE) scl[i] = (ok && scb) ? scb[sp] : 1.f;
  }
  const int vol_bytes = (int)(vol * sizeof(float));   // < 2^31 (checked by the host)
  float vin[G::KC][NS];
  f32x4 vw[NWQ];
  // global -> registers for one chunk of KC input channels (issued one chunk ahead of its use,
  // so HBM/L2 latency hides behind the previous chunk's MFMA stream).  Loads are unconditional (scalar channel
  // base + 32-bit lane offset); the zero padding is applied when the values are committed to LDS.
  // Buffer loads, one descriptor per channel built on the scalar unit: the lane address is a 32-bit offset and both
  // the zero padding and the channel tail (zero records) come out of the hardware range check.
  auto fetch = [&](int c0) {
#pragma unroll
    for (int cl = 0; cl < G::KC; ++cl) {
      const bool cok = (c0 + cl) < a.Cin;
      const uint64_t ba = reinterpret_cast<uint64_t>(inb + (size_t)(cok ? c0 + cl : 0) * vol);
      const uint64_t bu = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba) |
                          ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ba >> 32)) << 32);
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(bu), 0,
                                                        __builtin_amdgcn_readfirstlane(cok ? vol_bytes : 0), 0x00020000);
#pragma unroll
      for (int i = 0; i < NS; ++i)
        vin[cl][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)sob[i], 0, 0));
    }
    const float* wsrc = a.wpk + ((size_t)(c0 >> 1) * G::T * a.Coutp + co0) * 2;
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = tid + 256 * q;
      const int row = e / ROWQ, qq = e - row * ROWQ;
      if (e < NQ) vw[q] = reinterpret_cast<const f32x4*>(wsrc + (size_t)row * a.Coutp * 2)[qq];
    }
  };
  auto commit = [&](int c0) {  // registers -> LDS
#pragma unroll
    for (int cl = 0; cl < G::KC; ++cl) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int r = tid + 256 * i;
        if (r < G::PRAW) in_s[cl * G::P + r] = HAS_SCALE ? vin[cl][i] * scl[i] : vin[cl][i];
      }
    }
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = tid + 256 * q;
      if (e < NQ) reinterpret_cast<f32x4*>(w_s)[e] = vw[q];
    }
  };

  fetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += G::KC) {
    __syncthreads();  // previous chunk's MFMAs are done reading LDS
    commit(c0);
    __syncthreads();
    if (c0 + G::KC < a.Cin) fetch(c0 + G::KC);
    // ---- MFMA stream: one k-step (4 input channels) per tap.  (dz,dy) are real loops so the
    // scheduler's window -- and with it the number of LDS fragments it keeps in flight in
    // registers -- stays bounded; dx and the k-steps of a chunk are unrolled inside ----
#pragma unroll 1
    for (int dzy = 0; dzy < G::KS * G::KS; ++dzy) {
      const int dz = dzy / G::KS, dy = dzy - dz * G::KS;
      const float* arow = in_s + (dz * G::IY + dy) * G::IX;
      const float* brow = w_s + bbase + dzy * G::KS * G::COUT * 2;
#pragma unroll
      for (int dx = 0; dx < G::KS; ++dx) {
#pragma unroll
        for (int ks = 0; ks < G::KC / 4; ++ks) {
          float bf[G::NT];
#pragma unroll
          for (int n = 0; n < G::NT; ++n) bf[n] = brow[((ks * 2 * G::T + dx) * G::COUT + n * 16) * 2];
#pragma unroll
          for (int m = 0; m < G::MT; ++m) {
            const float av = arow[abase[m] + ks * 4 * G::P + dx];
#pragma unroll
            for (int n = 0; n < G::NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf[n], acc[m][n], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: BN scale/bias, residual, activation, store (lane = 4 x of one channel) ----
  const size_t oplane = (size_t)a.Ho * a.Wo;
  const size_t ovol = (size_t)a.Do * oplane;
  // Fast path (interior tiles, 16-byte aligned rows, one batch item < 4 GB): the row base is scalar and the
  // lane part of the address is one of NT*MTX precomputed 32-bit offsets, so an output row costs a packed
  // FMA pair, the activation and one 16-byte store.  The epilogue is issue-bound -- its SIMD is shared with
  // the MFMA stream of the co-resident block -- so instructions saved here are time saved.
  // ReLU / LeakyReLU / identity are max(v, slope*v) with slope 0 / 0.01 / 1; Mish has its own variant.
  const bool fast = a.fast_ok && co0 + G::COUT <= a.Cout && x0 + G::TW <= a.Wo && y0 + G::TH <= a.Ho &&
                    z0 + G::TD <= a.Do;
  const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
  auto epilogue_fast = [&](auto mishc, auto resc, auto reluc) __attribute__((always_inline)) {
    constexpr bool MISH = decltype(mishc)::value;
    constexpr bool RES = decltype(resc)::value;
    constexpr bool RELU = decltype(reluc)::value;
    unsigned loff[G::NT][G::MTX];
    float sc[G::NT], bi[G::NT];
#pragma unroll
    for (int n = 0; n < G::NT; ++n) {
      const int co = co0 + n * 16 + j;
      sc[n] = a.ch_scale ? a.ch_scale[co] : 1.f;
      bi[n] = a.ch_bias ? a.ch_bias[co] : 0.f;
#pragma unroll
      for (int xt = 0; xt < G::MTX; ++xt)
        loff[n][xt] = (unsigned)(((size_t)co * ovol + x0 + xt * 16 + 4 * kq) * sizeof(float));
    }
    const size_t bbase = (size_t)b * a.Cout * ovol;
    // skip values are requested RD-1 rows ahead; the ring has to fit beside the accumulators
    constexpr int RD = (G::MT * G::NT * 4 + 3 * G::NT * G::MTX * 4 <= 200) ? 3 : 2;
    f32x4 rv[RES ? RD : 1][G::NT][G::MTX];
    auto rowo = [&](int r) {   // scalar offset of this wave's r-th output row
      const int rr = wave * G::RPW + r;
      return bbase + (size_t)(z0 + rr / G::TH) * oplane + (size_t)(y0 + rr % G::TH) * a.Wo;
    };
    auto load_res = [&](int r) __attribute__((always_inline)) {
      const char* rrow = reinterpret_cast<const char*>(a.residual + rowo(r));
#pragma unroll
      for (int n = 0; n < G::NT; ++n)
#pragma unroll
        for (int xt = 0; xt < G::MTX; ++xt) rv[r % RD][n][xt] = *reinterpret_cast<const f32x4*>(rrow + loff[n][xt]);
    };
    if (RES) {
#pragma unroll
      for (int r = 0; r < RD - 1 && r < G::RPW; ++r) load_res(r);
    }
#pragma unroll
    for (int r = 0; r < G::RPW; ++r) {
      if (RES && r + RD - 1 < G::RPW) load_res(r + RD - 1);
      __builtin_amdgcn_sched_barrier(0);
      char* orow = reinterpret_cast<char*>(a.out + rowo(r));
#pragma unroll
      for (int n = 0; n < G::NT; ++n)
#pragma unroll
        for (int xt = 0; xt < G::MTX; ++xt) {
          f32x4 v = acc[r * G::MTX + xt][n] * sc[n] + bi[n];
          if (RES) v += rv[r % RD][n][xt];
          if (RELU) {                        // max(v, v*0): NaN stays NaN; packed multiply + one max per element
            v = __builtin_elementwise_max(v, v * 0.f);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = MISH ? dv_act(v[e], DV_ACT_MISH) : fmaxf(v[e], v[e] * slope);
          }
          *reinterpret_cast<f32x4*>(orow + loff[n][xt]) = v;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (fast) {
    if (a.act == DV_ACT_MISH) {
      if (a.residual) epilogue_fast(std::true_type{}, std::true_type{}, std::false_type{});
      else epilogue_fast(std::true_type{}, std::false_type{}, std::false_type{});
    } else if (a.act == DV_ACT_RELU) {
      if (a.residual) epilogue_fast(std::false_type{}, std::true_type{}, std::true_type{});
      else epilogue_fast(std::false_type{}, std::false_type{}, std::true_type{});
    } else {
      if (a.residual) epilogue_fast(std::false_type{}, std::true_type{}, std::false_type{});
      else epilogue_fast(std::false_type{}, std::false_type{}, std::false_type{});
    }
    return;
  }
#pragma unroll
  for (int n = 0; n < G::NT; ++n) {
    const int co = co0 + n * 16 + j;
    if (co >= a.Cout) continue;
    const float sc = a.ch_scale ? a.ch_scale[co] : 1.f;
    const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
    const size_t cbase = ((size_t)b * a.Cout + co) * ovol;
#pragma unroll
    for (int m = 0; m < G::MT; ++m) {
      const int rr = wave * G::RPW + m / G::MTX, xt = m % G::MTX;
      const int zo = z0 + rr / G::TH, yo = y0 + rr % G::TH, xo = x0 + xt * 16 + 4 * kq;
      if (zo >= a.Do || yo >= a.Ho || xo >= a.Wo) continue;
      const size_t o = cbase + (size_t)zo * oplane + (size_t)yo * a.Wo + xo;
      float v[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], sc, bi);
      if (a.vec_store) {
        if (a.residual) {
          const float4 rv = *reinterpret_cast<const float4*>(a.residual + o);
          v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
        *reinterpret_cast<float4*>(a.out + o) =
            make_float4(dv_act(v[0], a.act), dv_act(v[1], a.act), dv_act(v[2], a.act), dv_act(v[3], a.act));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (xo + r < a.Wo) {
            float u = v[r];
            if (a.residual) u += a.residual[o + r];
            a.out[o + r] = dv_act(u, a.act);
          }
      }
    }
  }
}

// ---- Cout == 1 (the classifier head, acv_ddim.py:214/:222): a single output channel would waste
// 15/16 of every MFMA tile, so this layer runs on the vector ALU.  A block owns a 4 x 8 x 64 output
// brick; each thread 8 consecutive x of one (z,y) row, i.e. 8 accumulators fed by 9 LDS rows of 10
// floats per input channel (two ds_read_b128 + one ds_read_b64 per row, conflict-free), with the 27
// weights of the channel in scalar registers.  Reads the input once: HBM-bound in the limit.
//
// Round 2: the staging went from wave-uniform row pointers (30 rows x 64-bit scalar pointers + bounds per chunk:
// 402 SGPR spills, 168 VGPRs, two barriers per 2-channel chunk; 0.90 ms = 29 % of the FMA issue rate) to the scheme
// of the MFMA kernels: one buffer descriptor per channel, 16 per-thread 32-bit offsets computed once per block
// (padding and the channel tail come out of the hardware range check), one channel per step through a
// double-buffered LDS brick with ONE barrier per channel, the next channel's loads in flight during the FMAs, and
// four blocks per CU (32.6 KB LDS, < 128 VGPRs).
namespace c1 {
constexpr int TZ = 4, TY = 8, TX = 64, XT = 8;
constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 2, RW = 68;   // RW/4 odd: rows alternate 16-B slot parity
constexpr int PLANE = IZ * IY * RW;
constexpr int PRAW = IZ * IY * IX;
constexpr int NS = (PRAW + 255) / 256;
}  // namespace c1

template <bool HAS_SCALE>
__global__ __launch_bounds__(256, 4) void conv3d_c1_kernel(ConvArgs a) {
  using namespace c1;
  __shared__ __attribute__((aligned(16))) float in_s[2 * PLANE];
  const int tid = threadIdx.x;
  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty; t /= a.nty;
  const int tz = t % a.ntz;
  const int b = t / a.ntz;
  const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
  const int row = tid >> 3, xs = (tid & 7) * XT;
  const int zl = row / TY, yl = row % TY;
  const size_t plane = (size_t)a.H * a.W, vol = (size_t)a.D * plane;
  const float* inb = a.in + (size_t)b * a.Cin * vol;
  const float* scb = (HAS_SCALE && a.in_scale) ? a.in_scale + (size_t)b * vol : nullptr;
  const float* w = a.wpk;  // raw [1][Cin][27] weights for this path

  float acc[XT];
#pragma unroll
  for (int i = 0; i < XT; ++i) acc[i] = 0.f;

  // staging plan: NS positions of the haloed 6 x 10 x 66 brick per thread, the same for every channel
  unsigned sob[NS];
  int slot[NS];
  float scl[HAS_SCALE ? NS : 1];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int r = tid + 256 * i;
    const int zz = r / (IY * IX), r2 = r - zz * (IY * IX);
    const int yy = r2 / IX, xx = r2 - yy * IX;
    const int z = z0 - 1 + zz, y = y0 - 1 + yy, x = x0 - 1 + xx;
    const bool ok = r < PRAW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    const unsigned sp = ok ? (unsigned)((z * a.H + y) * a.W + x) : 0u;
    sob[i] = ok ? sp * 4u : 0x80000000u;                    // beyond the buffer's records: the load returns 0
    slot[i] = r < PRAW ? (zz * IY + yy) * RW + xx : IX;     // column 66 of row 0: never read
    if (HAS_SCALE) scl[i] = (ok && scb) ? scb[sp] : 1.f;
  }
  const int vol_bytes = __builtin_amdgcn_readfirstlane((int)(vol * sizeof(float)));   // < 2^31 (checked by the host)
  uint64_t fb = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uint64_t>(inb)) |
                ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(reinterpret_cast<uint64_t>(inb) >> 32)) << 32);
  float vin[NS];
  auto fetch = [&](int c) __attribute__((always_inline)) {    // channels are fetched strictly in order: running base
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(fb), 0, c < a.Cin ? vol_bytes : 0, 0x00020000);
    fb += (uint64_t)(unsigned)vol_bytes;
#pragma unroll
    for (int i = 0; i < NS; ++i)
      vin[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)sob[i], 0, 0));
  };
  auto commit = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NS; ++i) buf[slot[i]] = HAS_SCALE ? vin[i] * scl[i] : vin[i];
  };

  fetch(0);
  commit(in_s);
  fetch(1);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < a.Cin; ++c) {
    const float* cur = in_s + (c & 1) * PLANE;
    float* nxt = in_s + ((c + 1) & 1) * PLANE;
    const float* wc = w + (size_t)c * 27;                      // wave-uniform -> scalar loads
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      if (dz == 1) {                       // a third of the way in: the next channel's loads (issued one channel ago)
        commit(nxt);                       // have landed; put them in the other buffer and refill the registers
        fetch(c + 2);
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const float* rp = cur + ((zl + dz) * IY + (yl + dy)) * RW + xs;
        const float4 q0 = *reinterpret_cast<const float4*>(rp);
        const float4 q1 = *reinterpret_cast<const float4*>(rp + 4);
        const float2 q2 = *reinterpret_cast<const float2*>(rp + 8);
        const float v[10] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y};
        const float w0 = wc[(dz * 3 + dy) * 3], w1 = wc[(dz * 3 + dy) * 3 + 1], w2 = wc[(dz * 3 + dy) * 3 + 2];
#pragma unroll
        for (int i = 0; i < XT; ++i) {
          acc[i] = fmaf(v[i], w0, acc[i]);
          acc[i] = fmaf(v[i + 1], w1, acc[i]);
          acc[i] = fmaf(v[i + 2], w2, acc[i]);
        }
      }
    }
    __syncthreads();                       // `cur` is free for channel c+2, `nxt` is complete
  }
  const int zo = z0 + zl, yo = y0 + yl, xo = x0 + xs;
  if (zo >= a.Do || yo >= a.Ho || xo >= a.Wo) return;
  const float sc = a.ch_scale ? a.ch_scale[0] : 1.f, bi = a.ch_bias ? a.ch_bias[0] : 0.f;
  const size_t o = (size_t)b * a.Do * a.Ho * a.Wo + ((size_t)zo * a.Ho + yo) * a.Wo + xo;
#pragma unroll
  for (int i = 0; i < XT; ++i) {
    float u = fmaf(acc[i], sc, bi);
    if (xo + i < a.Wo) {
      if (a.residual) u += a.residual[o + i];
      acc[i] = dv_act(u, a.act);
    }
  }
  if (a.vec_store && xo + XT <= a.Wo) {
    *reinterpret_cast<float4*>(a.out + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(a.out + o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  } else {
#pragma unroll
    for (int i = 0; i < XT; ++i)
      if (xo + i < a.Wo) a.out[o + i] = acc[i];
  }
}


// K4z: the single-channel head again, marching along z.  A block owns a 16 x 64 in-plane tile and a segment of ZS output
// planes; it streams the input planes z0-1 .. z0+ZS of all channels through LDS ONE plane at a time, and every plane
// feeds three rotating accumulator sets (the outputs one plane below, at, and above it), so a voxel is fetched once per
// segment instead of once per 4-plane brick: (ZS+2)/ZS x (18 x 66)/(16 x 64) = 1.35 x the tensor at ZS = 12, where the
// brick kernel above fetches 6 x 10 x 66 per 4 x 8 x 64 outputs = 1.93 x (PMC: 1.95 x, at the HBM ceiling).
namespace c1z {
constexpr int TY = 16, TX = 64, XT = 4;                     // (ZS, the planes of a segment, is a template parameter)
constexpr int IY = TY + 2, IX = TX + 2, RW = 68;            // RW/4 odd: rows alternate 16-byte slot parity
constexpr int PLANE = IY * RW;
constexpr int PRAW = IY * IX;
constexpr int NS = (PRAW + 255) / 256;
constexpr int WMAXC = 128;                                   // channels whose taps fit the LDS image (14 KB)
constexpr int CPS = 1;                                       // channels per step (per block barrier); 2 spills at four blocks per CU and is slower
}  // namespace c1z

template <int ZS>
__global__ __launch_bounds__(256, 4) void conv3d_c1z_kernel(ConvArgs a) {
  using namespace c1z;
  __shared__ __attribute__((aligned(16))) float in_s[2 * CPS * PLANE];
  // the 27 taps of every channel, padded to 28 floats: read as seven broadcast ds_read_b128 a step ahead of their use.
  // (Scalar loads would be the natural form -- the weights are wave-uniform -- but the output stores inside the loop
  // keep the compiler from proving them unclobbered, and it falls back to vector-memory loads with a wait per step.)
  __shared__ __attribute__((aligned(16))) float w_s[(WMAXC + CPS) * 28];
  const int tid = threadIdx.x;
  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty; t /= a.nty;
  const int tz = t % a.ntz;
  const int b = t / a.ntz;
  const int x0 = tx * TX, y0 = ty * TY, z0 = tz * ZS;
  const int yl = tid >> 4, xs = (tid & 15) * XT;
  const size_t plane = (size_t)a.H * a.W, vol = (size_t)a.D * plane;
  const int ngrp = (a.Cin + CPS - 1) / CPS;                  // channel groups of CPS (the tail group's surplus is zeros)
  for (int i = tid; i < ngrp * CPS * 28; i += 256) {
    const int ch = i / 28, tap = i - ch * 28;
    w_s[i] = (tap < 27 && ch < a.Cin) ? a.wpk[ch * 27 + tap] : 0.f;   // raw [1][Cin][27] weights
  }

  // staging plan of one haloed 18 x 66 plane: the same for every (plane, channel) step
  unsigned sob[NS];
  int slot[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int r = tid + 256 * i;
    const int yy = r / IX, xx = r - yy * IX;
    const int y = y0 - 1 + yy, x = x0 - 1 + xx;
    const bool ok = r < PRAW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    sob[i] = ok ? (unsigned)(y * a.W + x) * 4u : 0x80000000u;
    slot[i] = r < PRAW ? yy * RW + xx : IX;                 // column 66 of row 0: never read
  }
  const int plane_bytes = __builtin_amdgcn_readfirstlane((int)(plane * sizeof(float)));
  const int vol_bytes = __builtin_amdgcn_readfirstlane((int)(vol * sizeof(float)));
  const uint64_t in_b = reinterpret_cast<uint64_t>(a.in + (size_t)b * a.Cin * vol);
  const uint64_t in_bs = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)in_b) |
                         ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(in_b >> 32)) << 32);
  // step s = (plane index s / ngrp, channel group s % ngrp); plane z = z0 - 1 + plane index.  The descriptor of a
  // (plane, channel) starts at that plane of that channel and is one plane long: planes outside the volume and channels
  // past the end get zero records (the z padding / the group tail)
  const int nz = (a.D - z0 < ZS ? a.D - z0 : ZS) + 2;       // input planes this segment needs
  const int nsteps = nz * ngrp;
  float vin[CPS][NS];
  int fz = z0 - 1, fg = 0;                                   // (plane, channel group) of the next fetch
  auto fetch = [&]() __attribute__((always_inline)) {
    const bool zok = (unsigned)fz < (unsigned)a.D;
#pragma unroll
    for (int k = 0; k < CPS; ++k) {
      const int ch = fg * CPS + k;
      const bool ok = zok && ch < a.Cin;
      const uint64_t base = in_bs + (uint64_t)(unsigned)(ok ? ch : 0) * (uint64_t)(unsigned)vol_bytes +
                            (uint64_t)(unsigned)(ok ? fz : 0) * (uint64_t)(unsigned)plane_bytes;
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(base), 0, ok ? plane_bytes : 0, 0x00020000);
#pragma unroll
      for (int i = 0; i < NS; ++i)
        vin[k][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)sob[i], 0, 0));
    }
    if (++fg == ngrp) { fg = 0; ++fz; }
  };
  auto commit = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < CPS; ++k)
#pragma unroll
      for (int i = 0; i < NS; ++i) buf[k * PLANE + slot[i]] = vin[k][i];
  };

  const float sc = a.ch_scale ? a.ch_scale[0] : 1.f, bi = a.ch_bias ? a.ch_bias[0] : 0.f;
  const int yo = y0 + yl, xo = x0 + xs;
  const bool lane_in = yo < a.Ho && xo < a.Wo;
  const size_t obase = (size_t)b * a.Do * a.Ho * a.Wo + (size_t)yo * a.Wo + xo;
  auto emit = [&](const float (&acc)[XT], int zo) __attribute__((always_inline)) {   // output plane zo is complete
    if (zo < z0 || zo >= z0 + ZS || zo >= a.Do || !lane_in) return;
    const size_t o = obase + (size_t)zo * a.Ho * a.Wo;
    float u[XT];
#pragma unroll
    for (int i = 0; i < XT; ++i) {
      u[i] = fmaf(acc[i], sc, bi);
      if (a.residual && xo + i < a.Wo) u[i] += a.residual[o + i];
      u[i] = dv_act(u[i], a.act);
    }
    if (a.vec_store && xo + XT <= a.Wo) {
      *reinterpret_cast<float4*>(a.out + o) = make_float4(u[0], u[1], u[2], u[3]);
    } else {
#pragma unroll
      for (int i = 0; i < XT; ++i)
        if (xo + i < a.Wo) a.out[o + i] = u[i];
    }
  };

  float accP[XT], accC[XT], accN[XT];                        // outputs at z - 1, z, z + 1 of the plane in flight
#pragma unroll
  for (int i = 0; i < XT; ++i) accP[i] = accC[i] = accN[i] = 0.f;

  fetch();
  commit(in_s);
  fetch();
  __syncthreads();
  int g = 0, z = z0 - 1;
  float4 wq[CPS][7];                                         // this step's taps
#pragma unroll
  for (int k = 0; k < CPS; ++k)
#pragma unroll
    for (int q = 0; q < 7; ++q) wq[k][q] = reinterpret_cast<const float4*>(w_s + k * 28)[q];
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    const float* cur = in_s + (s & 1) * (CPS * PLANE);
    float* nxt = in_s + ((s + 1) & 1) * (CPS * PLANE);
    float wc[CPS][28];
#pragma unroll
    for (int k = 0; k < CPS; ++k)
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        wc[k][4 * q] = wq[k][q].x; wc[k][4 * q + 1] = wq[k][q].y; wc[k][4 * q + 2] = wq[k][q].z; wc[k][4 * q + 3] = wq[k][q].w;
      }
    {
      const int gn = g + 1 == ngrp ? 0 : g + 1;               // the next step's taps: in flight during this step
#pragma unroll
      for (int k = 0; k < CPS; ++k)
#pragma unroll
        for (int q = 0; q < 7; ++q) wq[k][q] = reinterpret_cast<const float4*>(w_s + (gn * CPS + k) * 28)[q];
    }
#pragma unroll
    for (int k = 0; k < CPS; ++k) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        if (k == 0 && dy == 1) {                             // the next step's loads (issued one step ago) have landed
          commit(nxt);
          fetch();
        }
        const float* rp = cur + k * PLANE + (yl + dy) * RW + xs;
        const float4 q0 = *reinterpret_cast<const float4*>(rp);
        const float2 q1 = *reinterpret_cast<const float2*>(rp + 4);
        const float v[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
        // input plane z is tap dz of output plane z + 1 - dz: dz = 0 -> accN, 1 -> accC, 2 -> accP
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float wN = wc[k][dy * 3 + dx], wC = wc[k][9 + dy * 3 + dx], wP = wc[k][18 + dy * 3 + dx];
#pragma unroll
          for (int i = 0; i < XT; ++i) {
            accN[i] = fmaf(v[i + dx], wN, accN[i]);
            accC[i] = fmaf(v[i + dx], wC, accC[i]);
            accP[i] = fmaf(v[i + dx], wP, accP[i]);
          }
        }
      }
    }
    if (++g == ngrp) {                                       // plane z is through: output z - 1 is complete
      emit(accP, z - 1);
#pragma unroll
      for (int i = 0; i < XT; ++i) { accP[i] = accC[i]; accC[i] = accN[i]; accN[i] = 0.f; }
      g = 0;
      ++z;
    }
    __syncthreads();                                         // `cur` is free for step s + 2, `nxt` is complete
  }
}

template <int ZS>
int launch_c1z(ConvArgs a, hipStream_t s) {
  a.ntx = (a.Wo + c1z::TX - 1) / c1z::TX;
  a.nty = (a.Ho + c1z::TY - 1) / c1z::TY;
  a.ntz = (a.Do + ZS - 1) / ZS;
  const long long blocks = (long long)a.B * a.ntz * a.nty * a.ntx;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  hipLaunchKernelGGL(conv3d_c1z_kernel<ZS>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return dv_launch_status();
}

int launch_c1(ConvArgs a, hipStream_t s) {
  // Plain inputs march along z (one fetch per voxel and segment); the filter prologue and small volumes keep the brick
  // kernel.  WHICH kernel runs must not depend on the batch size -- the two add in different orders, and a shard of a
  // batch has to produce the bits the whole batch produces (tests/test_gpu_fullsize.py, the multi-GPU sharding) -- so the
  // choice looks at one batch item only.  The segment length may follow the batch: every output is summed plane by
  // plane, channel by channel whatever segment it lies in, so ZS changes the block count and not a single bit.
  // dv_conv3d_set_c1z (tests) moves the threshold / pins the segment length.
  const long long min_pp = g_c1z_min_tiles.load(std::memory_order_relaxed);
  const int pin_zs = g_c1z_pin_zs.load(std::memory_order_relaxed);
  const long long tiles = (long long)((a.Ho + c1z::TY - 1) / c1z::TY) * ((a.Wo + c1z::TX - 1) / c1z::TX);
  auto blocks_at = [&](int zs) { return (long long)a.B * ((a.Do + zs - 1) / zs) * tiles; };
  if (!a.in_scale && a.Cin <= c1z::WMAXC && ((a.Do + 2) / 3) * tiles >= min_pp) {
    const int zs = pin_zs ? pin_zs : (blocks_at(12) >= 512 ? 12 : (blocks_at(6) >= 512 ? 6 : 3));
    if (zs >= 12) return launch_c1z<12>(a, s);
    if (zs >= 6) return launch_c1z<6>(a, s);
    return launch_c1z<3>(a, s);
  }
  a.ntx = (a.Wo + c1::TX - 1) / c1::TX;
  a.nty = (a.Ho + c1::TY - 1) / c1::TY;
  a.ntz = (a.Do + c1::TZ - 1) / c1::TZ;
  const long long blocks = (long long)a.B * a.ntz * a.nty * a.ntx;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  if (a.in_scale)
    hipLaunchKernelGGL(conv3d_c1_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(conv3d_c1_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return dv_launch_status();
}

__global__ void pack_conv_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin,
                                         int Cout, int T, int Cinp, int Coutp) {
  const size_t total = (size_t)Cinp * T * Coutp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int par = (int)(i & 1);
    size_t r = i >> 1;
    const int co = (int)(r % Coutp); r /= Coutp;
    const int tap = (int)(r % T);
    const int cp = (int)(r / T);
    const int ci = cp * 2 + par;
    wpk[i] = (ci < Cin && co < Cout) ? w[((size_t)co * Cin + ci) * T + tap] : 0.f;
  }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }
inline int coutp_of(int Cout) { return Cout <= 16 ? 16 : (Cout <= 32 ? 32 : pad_to(Cout, 64)); }

template <class G>
int launch_conv(ConvArgs a, hipStream_t s) {
  a.ntx = (a.Wo + G::TW - 1) / G::TW;
  a.nty = (a.Ho + G::TH - 1) / G::TH;
  a.ntz = (a.Do + G::TD - 1) / G::TD;
  a.nco = a.Coutp / G::COUT;
  const long long blocks = (long long)a.B * a.nco * a.ntz * a.nty * a.ntx;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  // the `volume * noise` prologue is specialised away for the layers that never use it: the hot Cout=32 layers, and
  // (round 4) the stride-2 layers -- with the prologue compiled in, a launch without a filter still multiplied every
  // staged value by 1 (48 vector multiplies per thread and chunk on the pipe the MFMAs issue on): 32->64 1.582 -> 1.536 ms,
  // 64->128 0.701 -> 0.668 ms (profiles/r04_kernel_experiments.txt)
  constexpr bool kSpecialise = (G::KS == 3 && ((G::S == 1 && G::NT == 2) || G::S == 2));
  if constexpr (kSpecialise) {
    if (!a.in_scale) {
      hipLaunchKernelGGL((conv3d_mfma_kernel<G, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
      return dv_launch_status();
    }
  }
  hipLaunchKernelGGL((conv3d_mfma_kernel<G, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  return dv_launch_status();
}

}  // namespace

extern "C" int dv_conv3d_set_s2_tile(int mode) {
  if (mode < 0 || mode > 2) return DV_ERR_UNSUPPORTED;
  g_s2_tile_pin.store(mode, std::memory_order_relaxed);
  return DV_OK;
}

extern "C" int dv_conv3d_set_c1z(int min_tiles, int segment) {
  if (min_tiles < 0 || (segment != 0 && segment != 3 && segment != 6 && segment != 12)) return DV_ERR_UNSUPPORTED;
  g_c1z_min_tiles.store(min_tiles > 0 ? min_tiles : 64, std::memory_order_relaxed);
  g_c1z_pin_zs.store(segment, std::memory_order_relaxed);
  return DV_OK;
}

extern "C" size_t dv_conv3d_packed_floats(int Cin, int Cout, int k) {
  if (Cin <= 0 || Cout <= 0 || (k != 1 && k != 3)) return 0;
  if (Cout == 1 && k == 3) return (size_t)pad_to(Cin * 27, 4);      // vector-ALU path: raw [Cin][27]
  return (size_t)pad_to(Cin, 8) * (k * k * k) * coutp_of(Cout);
}

extern "C" int dv_conv3d_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout, int k,
                                          dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(k == 1 || k == 3, DV_ERR_UNSUPPORTED);
  if (Cout == 1 && k == 3) {
    hipError_t e = hipMemcpyAsync(wpacked, w, (size_t)Cin * 27 * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream);
    return e == hipSuccess ? DV_OK : (int)e;
  }
  const int T = k * k * k, Cinp = pad_to(Cin, 8), Coutp = coutp_of(Cout);
  const size_t total = (size_t)Cinp * T * Coutp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     wpacked, Cin, Cout, T, Cinp, Coutp);
  return dv_launch_status();
}

extern "C" int dv_conv3d_f32(const float* in, const float* wpacked, const float* ch_scale,
                             const float* ch_bias, const float* in_scale, const float* residual,
                             float* out, int B, int Cin, int D, int H, int W, int Cout, int k,
                             int stride, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(k == 1 || k == 3, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(stride == 1 || stride == 2, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(!(k == 1 && stride != 1), DV_ERR_UNSUPPORTED);
  DV_REQUIRE((size_t)D * H * W * sizeof(float) <= 0x7fffffffull, DV_ERR_SHAPE);    // 31-bit byte offsets inside a channel
  DV_REQUIRE(!(Cout == 1 && k == 3 && stride != 1), DV_ERR_UNSUPPORTED);   // the single-channel head is packed for its own stride-1 kernel
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_LEAKY, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(dv_aligned16(wpacked), DV_ERR_ALIGN);
  ConvArgs a;
  a.in = in; a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.in_scale = in_scale;
  a.residual = residual; a.out = out;
  a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.Coutp = coutp_of(Cout);
  const int pad = (k - 1) / 2;
  a.Do = (D + 2 * pad - k) / stride + 1;
  a.Ho = (H + 2 * pad - k) / stride + 1;
  a.Wo = (W + 2 * pad - k) / stride + 1;
  a.act = act;
  a.vec_store = (a.Wo % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual));
  a.fast_ok = a.vec_store && (size_t)Cout * a.Do * a.Ho * a.Wo * sizeof(float) <= 0xffffffffull;
  a.ntx = a.nty = a.ntz = a.nco = 0;
  // tile walk order, measured per layer family at the bench sizes: the wide / strided layers run 5 % faster
  // with z-fastest walks (64->64 k3: 2.76 -> 2.63 ms, 32->64 s2: 1.65 -> 1.57 ms, 64->64 k1: 0.44 -> 0.30 ms);
  // the 32-channel layers prefer x-fastest
  a.order = 0;
  if (k == 3 && (a.Coutp >= 64 || stride == 2)) a.order = 1;
  if (k == 1 && a.Coutp >= 64) a.order = 2;
  hipStream_t s = (hipStream_t)stream;
  //                              KS S NT MTX TH TD KC WPS   (two blocks per CU everywhere)
  if (k == 3 && stride == 1) {
    if (Cout == 1) return launch_c1(a, s);
    if (a.Coutp == 16) return launch_conv<Geo<3, 1, 1, 2, 4, 4, 4, 2>>(a, s);
    if (a.Coutp == 32) {
      // 48-wide tiles when they divide the row (126 TFLOP/s; 32-wide tiles 120, also at three blocks per CU)
      if (a.Wo % 48 == 0) return launch_conv<Geo<3, 1, 2, 3, 4, 4, 4, 2>>(a, s);
      return launch_conv<Geo<3, 1, 2, 2, 4, 4, 4, 2>>(a, s);
    }
    return launch_conv<Geo<3, 1, 4, 2, 4, 2, 4, 2>>(a, s);
  }
  if (k == 3 && stride == 2) {
    if (a.Coutp == 16) return launch_conv<Geo<3, 2, 1, 2, 4, 2, 4, 2>>(a, s);
    if (a.Coutp == 32) return launch_conv<Geo<3, 2, 2, 2, 4, 2, 4, 2>>(a, s);
    // Launches that are small PER BATCH ITEM (the 64 -> 128 layers of the hourglasses: 192 blocks per pair) run 4 % faster
    // on 2 x 2 x 32 tiles at three blocks per CU (53.6 KB of LDS, 135 VGPRs): 0.730 -> 0.700 ms at batch 8; the 32 -> 64
    // layers (768 blocks per pair) are unchanged by it (profiles/r04_kernel_experiments.txt).  The choice looks at one
    // batch item only and both tilings sum every output in the same order (chunk by chunk, tap by tap), so a shard of a
    // batch reproduces the batch's bits.  dv_conv3d_set_s2_tile pins it (tests).
    const int pin = g_s2_tile_pin.load(std::memory_order_relaxed);      // dv_conv3d_set_s2_tile (tests)
    const long long per_item = (long long)(a.Coutp / 64) * ((a.Do + 1) / 2) * ((a.Ho + 3) / 4) * ((a.Wo + 31) / 32);
    const bool small = pin ? pin == 2 : per_item < 512;
    if (small) return launch_conv<Geo<3, 2, 4, 2, 2, 2, 4, 3>>(a, s);
    return launch_conv<Geo<3, 2, 4, 2, 4, 2, 4, 2>>(a, s);
  }
  // k == 1
  if (a.Coutp == 16) return launch_conv<Geo<1, 1, 1, 2, 4, 4, 8, 2>>(a, s);
  if (a.Coutp == 32) return launch_conv<Geo<1, 1, 2, 2, 4, 4, 8, 2>>(a, s);
  return launch_conv<Geo<1, 1, 4, 2, 4, 4, 8, 2>>(a, s);
}
