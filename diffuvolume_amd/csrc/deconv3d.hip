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
#include <atomic>
#include <type_traits>

#include "dv_common.h"

// csrc/deconv3d_pl.hip: the persistent loader-wave form of the k3 flavour (same packed weights)
int dv_deconv3d_pl_run(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias, const float* residual,
                       const float* skip, const float* rw, float* out, int B, int Cin, int D, int H, int W, int Cout, int Cskip,
                       int act, size_t wpk_floats, hipStream_t s);

namespace {

// test hook (dv_deconv3d_set_impl): 0 = the launcher picks, 1 = one-tile blocks (deconv3d_mfma_kernel) always,
// 2 = the persistent kernel wherever it takes the shape
std::atomic<int> g_deconv_impl_pin{0};

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
  const float* wpk;  // see pack_deconv_weights_kernel
  const float* ch_scale;
  const float* ch_bias;
  const float* residual;  // [B,Cout,2D,2H,2W] or null
  const float* skip;      // [B,Cskip,2D,2H,2W] or null: fused 1x1x1 `redir` convolution of the skip tensor
  const float* rw;        // [Cout][Cskip] redir weights (BN scale folded in)
  int Cskip;
  float* out;
  int B, Cin, D, H, W, Cout, Coutp;
  int ntx, nty, ntz, nco;
  int act;
  int vec_store;
  int fast_ok;             // host-side part of the fast-epilogue condition (alignment, 32-bit offsets)
};

template <int N> struct VecOf;
template <> struct VecOf<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int ACT>
__device__ __forceinline__ float act_c(float v) { return dv_act(v, ACT); }

template <int K, int KC_>
__global__ __launch_bounds__(256, 2) void deconv3d_mfma_kernel(DeconvArgs a) {
  using G = DGeo<K, KC_>;
  constexpr int kKC = G::KC, kT = G::T, kIY = G::IY, kIX = G::IX, kPRAW = G::PRAW, kP = G::P, LO = G::LO;
  constexpr int NKS = kKC / 4;                     // k-steps (4 input channels each) per chunk
  constexpr int BV = NKS * kNT;                    // B floats per lane and tap: [ks][n]
  typedef typename VecOf<BV>::type bvec;
  // main-loop image (input brick + weights) followed by four wave-private skip tiles for the fused `redir`
  constexpr int SKP = 4 * 2 * kTW + 32;            // skip-tile channel stride: == 32 (mod 64), k-lanes on disjoint banks
  constexpr int SK_FLOATS = 8 * SKP;               // 8 channels x 4 output rows x 64 columns per wave
  __shared__ __attribute__((aligned(16))) float smem[G::IN_FLOATS + G::W_FLOATS + 4 * SK_FLOATS];
  static_assert((G::IN_FLOATS + G::W_FLOATS + 4 * SK_FLOATS) * 4 <= 80 * 1024, "two blocks per CU");
  static_assert((G::IN_FLOATS + G::W_FLOATS) % 4 == 0, "skip tiles are written in 16-byte pieces");
  float* in_s = smem;
  float* w_s = smem + G::IN_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: row bases below stay in SGPRs
  const int j = lane & 15, kq = lane >> 4;

  // (the output-channel slices of a tile are neighbours in the linear order: same XCD, same time, one HBM read of the brick)
  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int tc = t % a.nco; t /= a.nco;
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty; t /= a.nty;
  const int tz = t % a.ntz;
  const int b = t / a.ntz;
  const int x0 = tx * kTW, y0 = ty * kTH, z0 = tz * kTD, co0 = tc * kCOUT;
  const int zl = wave / kTH, yl = wave % kTH;   // this wave's input row

  f32x4 acc[kMTX][8][kNT];
#pragma unroll
  for (int m = 0; m < kMTX; ++m)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int n = 0; n < kNT; ++n) acc[m][c][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float* arow0 = in_s + kq * kP + ((zl + LO) * kIY + yl + LO) * kIX + j + LO;
  const float* brow0 = w_s + lane * BV;          // LDS weights: [tap][kq][j][ks][n] -> one vector read per tap
  const size_t plane = (size_t)a.H * a.W, vol = (size_t)a.D * plane;
  const float* inb = a.in + (size_t)b * a.Cin * vol;

  // staging plan (as conv3d.hip): each thread owns NS positions of the haloed input brick, addressed as
  // scalar channel base + 32-bit byte offset (0 and a zero mask outside the volume: no divergent loads);
  // the next chunk's global loads are issued before the current chunk's MFMA stream
  constexpr int NS = (kPRAW + 255) / 256;
  constexpr int NQ = G::W_FLOATS / 4;
  constexpr int NWQ = (NQ + 255) / 256;
  static_assert(kP > kPRAW, "need a pad slot per channel plane");
  unsigned sob[NS];     // byte offset in a channel volume, or 2^31 (beyond the buffer's records: the load returns 0)
  int wslot[NS];        // LDS slot of that position; threads past the brick write the plane's pad slot
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int r = tid + 256 * i;
    const int zz = r / (kIY * kIX), r2 = r - zz * (kIY * kIX);
    const int yy = r2 / kIX, xx = r2 - yy * kIX;
    const int z = z0 - LO + zz, y = y0 - LO + yy, x = x0 - LO + xx;
    const bool ok = r < kPRAW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    sob[i] = ok ? (unsigned)((z * a.H + y) * a.W + x) * 4u : 0x80000000u;
    wslot[i] = r < kPRAW ? r : kP - 1;
  }
  const int vol_bytes = (int)(vol * sizeof(float));   // < 2^31 (checked by the host)
  float vin[kKC][NS];
  f32x4 vw[NWQ];
  const int nchunk = (a.Cin + kKC - 1) / kKC;
  auto fetch = [&](int c) {
#pragma unroll
    for (int cl = 0; cl < kKC; ++cl) {
      // buffer loads, one descriptor per channel built on the scalar unit: zero padding and the channel tail (zero
      // records) come out of the hardware range check, the lane address is a 32-bit offset
      const bool cok = (c * kKC + cl) < a.Cin;
      const uint64_t ba = reinterpret_cast<uint64_t>(inb + (size_t)(cok ? c * kKC + cl : 0) * vol);
      const uint64_t bu = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba) |
                          ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ba >> 32)) << 32);
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(bu), 0,
                                                        __builtin_amdgcn_readfirstlane(cok ? vol_bytes : 0), 0x00020000);
#pragma unroll
      for (int i = 0; i < NS; ++i)
        vin[cl][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)sob[i], 0, 0));
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk + ((size_t)c * a.nco + tc) * G::W_FLOATS);
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = tid + 256 * q;
      if (e < NQ) vw[q] = wsrc[e];
    }
  };
  auto commit = [&](int c) {
#pragma unroll
    for (int cl = 0; cl < kKC; ++cl)
#pragma unroll
      for (int i = 0; i < NS; ++i) in_s[cl * kP + wslot[i]] = vin[cl][i];
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = tid + 256 * q;
      if (e < NQ) reinterpret_cast<f32x4*>(w_s)[e] = vw[q];
    }
  };

  // input offsets a tap can ask for per dimension: K=3 -> {0,+1}, K=4 -> {-1,0,+1}
  constexpr int OMIN = (K == 4) ? -1 : 0, NOFF = (K == 4) ? 3 : 2, NOZ = (K == 4) ? 1 : 2;

  // ---- fused `redir` (acv_ddim.py:81-86, :91-92): out += W_r[cout][cskip] . skip[cskip][output voxel] ----
  // Extra K-steps of the same GEMM: for parity class (pz,py,px) the A rows are the skip voxels at
  // (2z+pz, 2y+py, 2x+px).  A wave only needs the four output rows of its own input row, so its skip tile
  // (8 channels x 4 rows x 64 columns) is wave-private: it is filled by LDS-DMA (global_load_lds, one 1-KB
  // instruction per channel, lane-linear image, no staging registers) at the top of a chunk and consumed after
  // that chunk's main MFMA stream -- no extra block barrier, no separate launch, and the 1x1x1 result never
  // goes through HBM.  Chunk c of the main loop carries skip channels 8c .. 8c+7.
  float* sk = smem + G::IN_FLOATS + G::W_FLOATS + wave * SK_FLOATS;
  const int nsk = a.skip ? (a.Cskip + 7) / 8 : 0;
  const int rr = lane >> 4, xq = (lane & 15) * 4;                      // this lane's skip row (pz,py) and column quad
  const size_t ovol2 = 8 * vol;
  const bool lok = nsk > 0 && (z0 + zl) < a.D && (y0 + yl) < a.H && 2 * x0 + xq < 2 * a.W;   // 2W % 4 == 0: quad all in / out
  const float* skl = nullptr;
  if (nsk > 0) {
#pragma unroll
    for (int i = 0; i < SK_FLOATS / 256; ++i)                           // masked lanes / channels must read zeros
      *reinterpret_cast<f32x4*>(sk + i * 256 + lane * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int oz = 2 * (z0 + zl) + (rr >> 1), oy = 2 * (y0 + yl) + (rr & 1);
    skl = a.skip + (size_t)b * a.Cskip * ovol2 + (lok ? ((size_t)oz * (2 * a.H) + oy) * (2 * a.W) + 2 * x0 + xq : 0);
  }
  static_assert(SK_FLOATS % 256 == 0, "zero fill covers the tile");

  fetch(0);
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();
    commit(c);
    __syncthreads();
    if (c + 1 < nchunk) fetch(c + 1);
    float bw[2][kNT];
    if (c < nsk) {
      if (lok) {
#pragma unroll
        for (int cl = 0; cl < 8; ++cl)
          if (c * 8 + cl < a.Cskip)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(skl + (size_t)(c * 8 + cl) * ovol2),
                (__attribute__((address_space(3))) void*)(sk + cl * SKP), 16, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int n = 0; n < kNT; ++n) {
          const int co = co0 + n * 16 + j, cs = c * 8 + ks * 4 + kq;
          bw[ks][n] = (co < a.Cout && cs < a.Cskip) ? a.rw[(size_t)co * a.Cskip + cs] : 0.f;
        }
    }
    // A fragments live in registers: every (z,y,x) input shift a tap can ask for x k-steps x M-tiles is read
    // from LDS once (K=3: all 8 shifts per chunk; K=4: the 9 (y,x) shifts of one z shift, reloaded when kz
    // changes slab), so the MFMA stream only needs one vector B read per tap, and that read is issued one tap
    // ahead (sched_barrier keeps the compiler from sinking it back next to its use).
    float av[NOZ][NOFF][NOFF][NKS][kMTX];
    auto load_a = [&](int slot, int offz) {
#pragma unroll
      for (int oy = 0; oy < NOFF; ++oy)
#pragma unroll
        for (int ox = 0; ox < NOFF; ++ox)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int m = 0; m < kMTX; ++m)
              av[slot][oy][ox][ks][m] = arow0[(offz * kIY + OMIN + oy) * kIX + OMIN + ox + ks * 4 * kP + m * 16];
    };
    if (NOZ > 1) {
#pragma unroll
      for (int oz = 0; oz < NOZ; ++oz) load_a(oz, OMIN + oz);
    }
    bvec bq[2];
    bq[0] = *reinterpret_cast<const bvec*>(brow0);
#pragma unroll
    for (int kz = 0; kz < K; ++kz) {
      if (NOZ == 1 && (kz == 0 || tap_off(kz) != tap_off(kz - 1))) load_a(0, tap_off(kz));
      const int zs = NOZ > 1 ? tap_off(kz) - OMIN : 0;
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int tap = (kz * K + ky) * K + kx;
          const int cls = (tap_par(kz) << 2) | (tap_par(ky) << 1) | tap_par(kx);
          if (tap + 1 < kT) bq[(tap + 1) & 1] = *reinterpret_cast<const bvec*>(brow0 + (tap + 1) * 64 * BV);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int m = 0; m < kMTX; ++m)
#pragma unroll
              for (int n = 0; n < kNT; ++n)
                acc[m][cls][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    av[zs][tap_off(ky) - OMIN][tap_off(kx) - OMIN][ks][m], bq[tap & 1][ks * kNT + n], acc[m][cls][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (c < nsk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the LDS-DMA of this chunk has landed (long ago)
      const float* skr = sk + kq * SKP + 2 * j;
#pragma unroll
      for (int cls = 0; cls < 8; ++cls) {
        float sa[2][kMTX];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int m = 0; m < kMTX; ++m) sa[ks][m] = skr[ks * 4 * SKP + (cls >> 1) * (2 * kTW) + m * 32 + (cls & 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int m = 0; m < kMTX; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n)
              acc[m][cls][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[ks][m], bw[ks][n], acc[m][cls][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // epilogue: px=0 / px=1 classes interleave along x -> 8 consecutive outputs per lane
  const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
  const int zi = z0 + zl, yi = y0 + yl;
  if (zi >= a.D || yi >= a.H) return;
  // Fast path (interior tiles, 16-byte aligned rows): everything address-like is either a scalar row base or
  // one of four precomputed 32-bit lane offsets, so an output row costs 4 packed FMAs, the activation and two
  // 16-byte stores -- the epilogue is issue-bound (it shares its SIMD with another block's MFMA stream).
  // (x tiles that hang over the row end take the same path: W is even, so each 16-byte half of a lane's eight
  // outputs -- two input columns -- is all inside or all outside, and a lane predicate per half suffices)
  const bool fast = a.fast_ok && co0 + kCOUT <= a.Cout;
  // ReLU / LeakyReLU / identity are max(v, slope * v) with slope 0 / 0.01 / 1; Mish has its own variant.
  const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
  auto epilogue_fast = [&](auto mishc, auto resc, auto reluc) __attribute__((always_inline)) {
    constexpr bool MISH = decltype(mishc)::value;
    constexpr bool RES = decltype(resc)::value;
    constexpr bool RELU = decltype(reluc)::value;
    unsigned loff[kNT][kMTX];
    float sc[kNT], bi[kNT];
    bool lo_ok[kMTX], hi_ok[kMTX];
#pragma unroll
    for (int m = 0; m < kMTX; ++m) {
      lo_ok[m] = x0 + m * 16 + 4 * kq + 1 < a.W;
      hi_ok[m] = x0 + m * 16 + 4 * kq + 3 < a.W;
    }
#pragma unroll
    for (int n = 0; n < kNT; ++n) {
      const int co = co0 + n * 16 + j;
      sc[n] = a.ch_scale ? a.ch_scale[co] : 1.f;
      bi[n] = a.ch_bias ? a.ch_bias[co] : 0.f;
#pragma unroll
      for (int m = 0; m < kMTX; ++m)
        loff[n][m] = (unsigned)(((size_t)co * ovol + 2 * (x0 + m * 16 + 4 * kq)) * sizeof(float));
    }
    const size_t bbase = (size_t)b * a.Cout * ovol;
    auto rowo = [&](int k) {   // scalar offset of output row (pz, py) = (k >> 1, k & 1)
      return bbase + (size_t)(2 * zi + (k >> 1)) * oplane + (size_t)(2 * yi + (k & 1)) * Wo;
    };
    // step q = (row k, channel half n); skip values are requested RD-1 steps ahead (16 registers per buffer)
    constexpr int RD = (K == 3) ? 3 : 2;   // ring of skip-value buffers: step q+RD-1 is in flight while step q is stored
    f32x4 rv[RES ? RD : 1][kMTX][2];
    auto load_res = [&](int q) __attribute__((always_inline)) {
      const char* rrow = reinterpret_cast<const char*>(a.residual + rowo(q >> 1));
#pragma unroll
      for (int m = 0; m < kMTX; ++m) {
        rv[q % RD][m][0] = lo_ok[m] ? *reinterpret_cast<const f32x4*>(rrow + loff[q & 1][m]) : (f32x4){0.f, 0.f, 0.f, 0.f};
        rv[q % RD][m][1] = hi_ok[m] ? *reinterpret_cast<const f32x4*>(rrow + loff[q & 1][m] + 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    if (RES) {
#pragma unroll
      for (int q = 0; q < RD - 1; ++q) load_res(q);
    }
#pragma unroll
    for (int q = 0; q < 4 * kNT; ++q) {
      static_assert(kNT == 2, "q & 1 is the channel half");
      const int k = q >> 1, n = q & 1;
      if (RES && q + RD - 1 < 4 * kNT) load_res(q + RD - 1);
      __builtin_amdgcn_sched_barrier(0);
      char* orow = reinterpret_cast<char*>(a.out + rowo(k));
#pragma unroll
      for (int m = 0; m < kMTX; ++m) {
        const f32x4 e0 = acc[m][k << 1][n], e1 = acc[m][(k << 1) | 1][n];
        // (scalar fmas straight into the interleaved order: the packed form needs eight moves to pair the two x
        // parities first; while this block is in its epilogue its SIMDs run on the other block's waves alone, so the
        // LENGTH of the epilogue in vector instructions is what counts, profiles/r03_wino3d_epilogue.txt)
        f32x4 lo, hi;
        lo[0] = fmaf(e0[0], sc[n], bi[n]); lo[1] = fmaf(e1[0], sc[n], bi[n]);
        lo[2] = fmaf(e0[1], sc[n], bi[n]); lo[3] = fmaf(e1[1], sc[n], bi[n]);
        hi[0] = fmaf(e0[2], sc[n], bi[n]); hi[1] = fmaf(e1[2], sc[n], bi[n]);
        hi[2] = fmaf(e0[3], sc[n], bi[n]); hi[3] = fmaf(e1[3], sc[n], bi[n]);
        if (RES) {
          lo += rv[q % RD][m][0];
          hi += rv[q % RD][m][1];
        }
        if (RELU) {                          // max(v, v*0): NaN stays NaN; packed multiply + one max per element
          lo = __builtin_elementwise_max(lo, lo * 0.f);
          hi = __builtin_elementwise_max(hi, hi * 0.f);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            lo[r] = MISH ? dv_act(lo[r], DV_ACT_MISH) : fmaxf(lo[r], lo[r] * slope);
            hi[r] = MISH ? dv_act(hi[r], DV_ACT_MISH) : fmaxf(hi[r], hi[r] * slope);
          }
        }
        if (lo_ok[m]) *reinterpret_cast<f32x4*>(orow + loff[n][m]) = lo;
        if (hi_ok[m]) *reinterpret_cast<f32x4*>(orow + loff[n][m] + 16) = hi;
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
  }
  auto epilogue = [&](auto actc) __attribute__((always_inline)) {
    constexpr int ACT = decltype(actc)::value;
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
              *reinterpret_cast<float4*>(a.out + o) =
                  make_float4(act_c<ACT>(v[0]), act_c<ACT>(v[1]), act_c<ACT>(v[2]), act_c<ACT>(v[3]));
              *reinterpret_cast<float4*>(a.out + o + 4) =
                  make_float4(act_c<ACT>(v[4]), act_c<ACT>(v[5]), act_c<ACT>(v[6]), act_c<ACT>(v[7]));
            } else {
#pragma unroll
              for (int r = 0; r < 8; ++r)
                if (2 * xi + r < Wo) {
                  float u = v[r];
                  if (a.residual) u += a.residual[o + r];
                  a.out[o + r] = act_c<ACT>(u);
                }
            }
          }
      }
    }
  };
  if (!fast) {
    switch (a.act) {
      case DV_ACT_RELU: epilogue(std::integral_constant<int, DV_ACT_RELU>{}); break;
      case DV_ACT_MISH: epilogue(std::integral_constant<int, DV_ACT_MISH>{}); break;
      case DV_ACT_LEAKY: epilogue(std::integral_constant<int, DV_ACT_LEAKY>{}); break;
      default: epilogue(std::integral_constant<int, DV_ACT_NONE>{}); break;
    }
  }
}

// packed weights: [chunk = ci / KC][co block of 32][tap][kq = ci % 4][j = co % 16][ks = (ci % KC) / 4][n = (co % 32) / 16]
// = exactly the order the kernel's LDS image wants, so staging is a straight 16-byte copy.
__global__ void pack_deconv_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin,
                                           int Cout, int nchunk, int nco, int kT, int KC) {
  const int NKS = KC / 4;
  const size_t total = (size_t)nchunk * nco * kT * 64 * NKS * kNT;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int n = (int)(r % kNT); r /= kNT;
    const int ks = (int)(r % NKS); r /= NKS;
    const int j = (int)(r % 16); r /= 16;
    const int kq = (int)(r % 4); r /= 4;
    const int tap = (int)(r % kT); r /= kT;
    const int tc = (int)(r % nco);
    const int c = (int)(r / nco);
    const int ci = c * KC + ks * 4 + kq, co = tc * kCOUT + n * 16 + j;
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

constexpr int kc_of(int K) { return K == 3 ? 8 : 4; }   // input channels per LDS chunk

int pack_any(const float* w, float* wpacked, int Cin, int Cout, int K, hipStream_t s) {
  const int T = K * K * K, KC = kc_of(K), nchunk = pad_to(Cin, 8) / KC, nco = pad_to(Cout, kCOUT) / kCOUT;
  const size_t total = (size_t)nchunk * nco * T * KC * kCOUT;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_deconv_weights_kernel, dim3(blocks), dim3(256), 0, s, w, wpacked, Cin, Cout, nchunk, nco, T, KC);
  return dv_launch_status();
}

int run_any(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
            const float* residual, float* out, int B, int Cin, int D, int H, int W, int Cout, int act, int K,
            hipStream_t s, const float* skip = nullptr, const float* rw = nullptr, int Cskip = 0) {
  DeconvArgs a;
  a.skip = skip; a.rw = rw; a.Cskip = Cskip;
  a.in = in; a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.residual = residual;
  a.out = out; a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout;
  a.Coutp = pad_to(Cout, kCOUT);
  a.ntx = (W + kTW - 1) / kTW;
  a.nty = (H + kTH - 1) / kTH;
  a.ntz = (D + kTD - 1) / kTD;
  a.nco = a.Coutp / kCOUT;
  a.act = act;
  a.vec_store = (W % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual));   // generic path's 32-byte stores
  if ((size_t)D * H * W * sizeof(float) > 0x7fffffffull) return DV_ERR_SHAPE;   // 31-bit in-channel byte offsets
  // fast epilogue: scalar row base + 32-bit per-lane byte offsets inside one batch item
  a.fast_ok = (W % 2 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual)) &&
              (size_t)Cout * 8 * D * H * W * sizeof(float) <= 0xffffffffull;
  // the persistent form (csrc/deconv3d_pl.hip) wherever it takes the shape.  The choice looks at one batch item only: a shard
  // of a batch runs the kernel the batch runs.
  const int pin = g_deconv_impl_pin.load(std::memory_order_relaxed);
  // (with a residual tensor the epilogue's loads share the wave's in-order memory counter with its stores: the one-tile
  // kernel's ring of residual rows handles that better; the persistent kernel takes it only when pinned)
  if (K == 3 && pin != 1 && (pin == 2 || !residual) &&
      dv_deconv3d_pl_supported(Cin, Cout, D, H, W, skip ? Cskip : 0) && dv_aligned16(in) && dv_aligned16(out) &&
      (!residual || dv_aligned16(residual)) && (!skip || dv_aligned16(skip)))
    return dv_deconv3d_pl_run(in, wpacked, ch_scale, ch_bias, residual, skip, rw, out, B, Cin, D, H, W, Cout, Cskip, act,
                              dv_deconv3d_packed_floats(Cin, Cout), s);
  return K == 3 ? launch_deconv<3, 8>(a, s) : launch_deconv<4, 4>(a, s);
}

}  // namespace

extern "C" int dv_deconv3d_set_impl(int mode) {
  DV_REQUIRE(mode >= 0 && mode <= 2, DV_ERR_UNSUPPORTED);
  g_deconv_impl_pin.store(mode, std::memory_order_relaxed);
  return DV_OK;
}

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

extern "C" int dv_deconv3d_k3s2_redir_f32(const float* in, const float* wpacked, const float* ch_bias,
                                          const float* skip, const float* redir_w, float* out, int B, int Cin,
                                          int D, int H, int W, int Cout, int Cskip, int act, dv_stream_t stream) {
  const float* ch_scale = nullptr;
  const float* residual = nullptr;
  (void)ch_scale; (void)residual;
  DV_DECONV_CHECKS
  DV_REQUIRE_PTR(skip);
  DV_REQUIRE_PTR(redir_w);
  DV_REQUIRE(Cskip > 0, DV_ERR_SHAPE);
  DV_REQUIRE(W % 2 == 0 && dv_aligned16(skip), DV_ERR_UNSUPPORTED);     // 16-byte skip quads never straddle a row end
  DV_REQUIRE((Cskip + 7) / 8 <= (Cin + 7) / 8, DV_ERR_UNSUPPORTED);       // skip chunks ride on the main chunks
  return run_any(in, wpacked, nullptr, ch_bias, nullptr, out, B, Cin, D, H, W, Cout, act, 3, (hipStream_t)stream, skip,
                 redir_w, Cskip);
}
