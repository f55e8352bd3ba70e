// K11: 2-D convolution (3x3 with dilation 1..16, or 1x1; stride 1, "same" padding) + BatchNorm (eval) +
// residual + Mish/ReLU/LeakyReLU, fp32 on v_mfma_f32_16x16x4_f32.  This is the per-DDIM-step disparity
// refinement stack of the KITTI12 flavour: refinenet_version3 (KITTI12/models/pwcnet_ddim.py:251-306) built
// from convbn (KITTI12/models/submodule.py:21-24, padding = dilation) and BasicBlock (:192-215, out += x
// with no activation after the add).  At 1248x384 it is 0.85 TFLOP per pair and step.
//
// GEMM view (as conv3d.hip): M = 16 consecutive x of an output row, N = 16 cout, K = tap x cin.  A block owns
// 8 rows x 64 columns x 32 (or 16) output channels; each wave 2 rows x 4 M-tiles.  Per chunk of KC input channels
// the input rows the 9 taps can touch are staged in LDS:
//   * dilation <= 4 ("contiguous"): the haloed brick (8+2d) x (64+2d);
//   * dilation 5..16 ("banded"): three 8-row bands (one per ky, d rows apart) x (64+2d) -- a haloed brick would
//     be (8+32) rows for 8 rows of output.
// The dilation is a run-time value: all tap offsets are wave-uniform scalars added to one per-lane base.
// Weights sit in LDS as [tap][quad][lane][4] so that a lane fetches the B fragments of a tap (all k-steps and
// N tiles) with 16-byte reads; A fragments and the B vector of step s+1 are requested before the MFMAs of
// step s (register double buffer, order pinned with sched_barrier); the next chunk's global loads are issued
// one chunk ahead (as conv3d.hip).
#include <type_traits>

#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


// LDS channel-plane stride: > n (a spare slot) and such that the 4 k-lanes of an A read hit disjoint banks:
// == 16 (mod 32) for unit-stride reads, odd for the stride-2 reads of a strided convolution
__host__ __device__ constexpr int plane_pad(int n, int stride = 1) {
  return stride == 1 ? (n + 1) + ((16 - (n + 1) % 32) + 32) % 32 : ((n + 1) | 1);
}

template <int KS_, int NT_, int KC_, int DMAX_, bool BANDED_, int RPW_, int WPS_ = 2, int MTX_ = 4, int S_ = 1>
struct G2 {
  static constexpr int S = S_;         // stride (2: the down-sampling layers of the 2-D feature CNNs)
  static constexpr int WPS = WPS_;     // blocks per CU the register budget is set for
  static constexpr int MTX = MTX_, TW = MTX_ * 16;     // M tiles (16 columns each) per row
  static constexpr int KS = KS_, NT = NT_, KC = KC_, DMAX = DMAX_;
  static constexpr int RPW = RPW_, TH = 4 * RPW_, MT = RPW_ * MTX;     // 4 waves x RPW rows
  static constexpr bool BANDED = BANDED_;
  static constexpr int T = KS * KS, NKS = KC / 4, COUT = NT * 16;
  static constexpr int BV = NKS * NT;                 // B floats per lane and tap: [ks][n]
  static constexpr int VW = BV < 4 ? BV : 4;          // floats per LDS read
  static constexpr int Q = BV / VW;
  static constexpr int RMAX = BANDED ? 3 * TH : (TH - 1) * S + 1 + (KS - 1) * DMAX;
  static constexpr int CMAX = (TW - 1) * S + 1 + (KS - 1) * DMAX;
  static_assert(!(BANDED && S != 1), "banded staging is stride 1");
  static constexpr int PMAX = plane_pad(RMAX * CMAX, S);
  static constexpr int IN_FLOATS = KC * PMAX, W_FLOATS = T * KC * COUT;
  static_assert(BV == 1 || BV == 2 || BV % 4 == 0, "B vector width");
  static_assert((IN_FLOATS + W_FLOATS) * 4 * WPS <= 160 * 1024, "WPS blocks per CU");
};

struct Conv2dArgs {
  const float* in;        // [B,Cin,H,W] (first source)
  const float* in_more[3];  // further sources of a virtual torch.cat along channels, or null
  int cend[4];            // cumulative channel count after each source (cend[nsrc-1] == Cin)
  const float* wpk;       // see pack_conv2d_weights_kernel
  const float* ch_scale;  // [Cout] or null
  const float* ch_bias;   // [Cout] or null
  const float* residual;  // [B,Cout,H,W] or null
  const float* mul;       // [B,Cout,H,W] or null: v *= mul after the activation
  const float* blend_z;   // [B,Cout,H,W] or null: v = blend_h + blend_z * (v - blend_h)
  const float* blend_h;
  float* out;             // [B,Cout,H,W]
  int B, Cin, H, W, Cout;
  int Ho, Wo;             // output size (= H, W for stride 1)
  int dil;                // dilation (= padding); 0 for 1x1
  int ntx, nty, nco;
  int act, vec_store, fast_ok;
  // K-split (small launches: a single IGEV pair at 1/8 and 1/16 resolution is 240 / 72 blocks, each a serial loop over
  // ~100 four-channel chunks): `kslices` blocks share an output tile, each sums a contiguous range of the input-channel
  // chunks and stores its raw partial tile to scratch[slice][B,Cout,Ho,Wo]; conv2d_ksplit_epilogue_kernel adds the
  // slices in a fixed order and applies scale / bias / residual / activation / gates
  float* scratch;
  int kslices;
};

template <class G>
__global__ __launch_bounds__(256, G::WPS) void conv2d_mfma_kernel(Conv2dArgs a) {
  constexpr int KS = G::KS, NT = G::NT, KC = G::KC, NKS = G::NKS, BV = G::BV, VW = G::VW, Q = G::Q, T = G::T;
  constexpr int TH = G::TH, RPW = G::RPW, MT = G::MT, MTX = G::MTX, TW = G::TW, S = G::S;
  __shared__ __attribute__((aligned(16))) float smem[G::IN_FLOATS + G::W_FLOATS];
  float* in_s = smem;
  float* w_s = smem + G::IN_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;

  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty; t /= a.nty;
  const int tc = t % a.nco; t /= a.nco;
  // (integer division runs on the vector ALU: pin the wave-uniform results back into scalar registers)
  const int slice = __builtin_amdgcn_readfirstlane(a.kslices > 1 ? (int)(t % (unsigned)a.kslices) : 0);
  const int b = a.kslices > 1 ? (int)(t / (unsigned)a.kslices) : (int)t;
  const int x0 = tx * TW, y0 = ty * TH, co0 = tc * G::COUT;
  const int d = a.dil;
  const int C = (TW - 1) * S + 1 + (KS - 1) * d;                        // staged columns
  const int R = G::BANDED ? 3 * TH : (TH - 1) * S + 1 + (KS - 1) * d;   // staged rows
  const int P = plane_pad(R * C, S);                                    // channel plane stride in LDS

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const size_t plane = (size_t)a.H * a.W;
  const int plane_bytes = (int)(plane * sizeof(float));      // < 2^31 (checked by the host)
  const int nchunk = (a.Cin + KC - 1) / KC;
  const int cbeg = __builtin_amdgcn_readfirstlane(a.kslices > 1 ? (nchunk * slice) / a.kslices : 0);
  const int cend = __builtin_amdgcn_readfirstlane(a.kslices > 1 ? (nchunk * (slice + 1)) / a.kslices : nchunk);
  // The channels of the (virtual) concatenation are fetched strictly in order, so the position in it is running scalar
  // state -- byte address of the next channel plane, channels left in its source, a queue of the sources to come -- moved
  // by selects (conv2d_wino.hip; looking the source up per channel cost two dependent kernarg loads, each closed by an
  // s_waitcnt lgkmcnt(0), and a 64-bit multiply chain per channel).
  auto sgpr64 = [](uint64_t v) __attribute__((always_inline)) {
    return (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
           ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
  };
  constexpr int NEVER = 0x7fffffff;                 // an unused source slot: its counter never reaches zero
  const int cw1 = a.cend[1] - a.cend[0], cw2 = a.cend[2] - a.cend[1], cw3 = a.cend[3] - a.cend[2];
  uint64_t fb = sgpr64(reinterpret_cast<uint64_t>(a.in + (size_t)b * a.cend[0] * plane));
  uint64_t nb1 = sgpr64(reinterpret_cast<uint64_t>(a.in_more[0] ? a.in_more[0] + (size_t)b * cw1 * plane : a.in));
  uint64_t nb2 = sgpr64(reinterpret_cast<uint64_t>(a.in_more[1] ? a.in_more[1] + (size_t)b * cw2 * plane : a.in));
  uint64_t nb3 = sgpr64(reinterpret_cast<uint64_t>(a.in_more[2] ? a.in_more[2] + (size_t)b * cw3 * plane : a.in));
  int left = a.cend[0], nl1 = cw1 > 0 ? cw1 : NEVER, nl2 = cw2 > 0 ? cw2 : NEVER, nl3 = cw3 > 0 ? cw3 : NEVER;
  int fc = cbeg * KC;                               // (a K-slice starts in the middle of the concatenation)
  auto queue_up = [&]() __attribute__((always_inline)) {
    fb = nb1; left = nl1;
    nb1 = nb2; nl1 = nl2;
    nb2 = nb3; nl2 = nl3;
    nl3 = NEVER;
  };
  {
    int skip = fc;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (skip >= left) { skip -= left; queue_up(); }
    fb += (uint64_t)(unsigned)skip * (uint64_t)(unsigned)plane_bytes;
    left -= skip;
  }

  // ---- staging plan: each thread owns NS positions of the staged rows (same for every channel) ----
  constexpr int NS = (G::RMAX * G::CMAX + 255) / 256;
  constexpr int NQ = G::W_FLOATS / 4, NWQ = (NQ + 255) / 256;
  unsigned sob[NS];          // byte offset in a channel plane, or 2^31 (beyond the buffer's records: the load returns 0)
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int r = tid + 256 * i;
    const int row = r / C, col = r - row * C;
    const int gy = G::BANDED ? y0 + (row / TH - 1) * d + row % TH : y0 * S - (KS / 2) * d + row;
    const int gx = x0 * S - (KS / 2) * d + col;
    const bool ok = r < R * C && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    sob[i] = ok ? (unsigned)(gy * a.W + gx) * 4u : 0x80000000u;
  }
  float vin[KC][NS];
  f32x4 vw[NWQ];
  auto fetch = [&](int c) {
#pragma unroll
    for (int cl = 0; cl < KC; ++cl) {
      // buffer loads, one descriptor per channel built on the scalar unit: zero padding and the channel tail (zero
      // records) come out of the hardware range check, the lane address is a 32-bit offset
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(fb), 0, fc < a.Cin ? plane_bytes : 0,
                                                        0x00020000);
#pragma unroll
      for (int i = 0; i < NS; ++i)
        vin[cl][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)sob[i], 0, 0));
      ++fc;
      --left;
      fb += (uint64_t)(unsigned)plane_bytes;
      const bool sw = left == 0;                    // source exhausted: the queue moves up
      fb = sw ? nb1 : fb;   left = sw ? nl1 : left;
      nb1 = sw ? nb2 : nb1; nl1 = sw ? nl2 : nl1;
      nb2 = sw ? nb3 : nb2; nl2 = sw ? nl3 : nl2;
      nl3 = sw ? NEVER : nl3;
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
    for (int cl = 0; cl < KC; ++cl) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int r = tid + 256 * i;      // threads past the brick write the plane's spare slot
        in_s[cl * P + (r < R * C ? r : P - 1)] = vin[cl][i];
      }
    }
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = tid + 256 * q;
      if (e < NQ) reinterpret_cast<f32x4*>(w_s)[e] = vw[q];
    }
  };

  // step s = (tap, ks): 8 A fragments (2 rows x 4 M-tiles) + the tap's B vector
  const float* abase = in_s + kq * P + (wave * RPW * S) * C + j * S;
  const float* bbase = w_s + lane * VW;
  // (dq, Cq, Pq) are per-chunk opaque copies of (d, C, P): without them the compiler hoists the 9*NKS*RPW
  // tap addresses out of the channel loop and parks them in ~50 VGPRs; recomputing them is a scalar add each.
  auto load_a = [&](float (&av)[MT], int tap, int ks, int dq, int Cq, int Pq) __attribute__((always_inline)) {
    const int ky = tap / KS, kx = tap - ky * KS;
    const int off = (G::BANDED ? ky * TH : ky * dq) * Cq + kx * dq + ks * 4 * Pq;   // wave-uniform
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int xt = 0; xt < MTX; ++xt) av[r * MTX + xt] = abase[off + r * S * Cq + xt * 16 * S];
  };
  auto load_b = [&](float (&bv)[BV], int tap) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float* p = bbase + (tap * Q + q) * 64 * VW;
      if constexpr (VW == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[q * 4 + e] = v[e];
      } else if constexpr (VW == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(p);
        bv[0] = v[0];
        bv[1] = v[1];
      } else {
        bv[0] = p[0];
      }
    }
  };

  fetch(cbeg);
  for (int c = cbeg; c < cend; ++c) {
    __syncthreads();
    commit(c);
    __syncthreads();
    if (c + 1 < cend) fetch(c + 1);
    float av[2][MT];
    float bv[2][BV];
    int dq = d, Cq = C, Pq = P;
    asm volatile("" : "+s"(dq), "+s"(Cq), "+s"(Pq));
    load_a(av[0], 0, 0, dq, Cq, Pq);
    load_b(bv[0], 0);
#pragma unroll
    for (int s = 0; s < T * NKS; ++s) {
      const int tap = s / NKS, ks = s - tap * NKS;
      if (s + 1 < T * NKS) {
        load_a(av[(s + 1) & 1], (s + 1) / NKS, (s + 1) % NKS, dq, Cq, Pq);
        if ((s + 1) % NKS == 0) load_b(bv[((s + 1) / NKS) & 1], (s + 1) / NKS);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s & 1][m], bv[tap & 1][ks * NT + n], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: BN scale/bias, residual, activation; lane = 4 x of one channel ----
  const size_t oplane = (size_t)a.Ho * a.Wo;
  if (a.kslices > 1) {        // raw partial sums of this slice; the fused epilogue runs in the reduction kernel
    float* sp = a.scratch + (size_t)slice * a.B * a.Cout * oplane;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = co0 + n * 16 + j;
      if (co >= a.Cout) continue;
      const size_t cbase = ((size_t)b * a.Cout + co) * oplane;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int yo = y0 + wave * RPW + m / MTX, xo = x0 + (m % MTX) * 16 + 4 * kq;
        if (yo >= a.Ho || xo >= a.Wo) continue;
        const size_t o = cbase + (size_t)yo * a.Wo + xo;
        if (a.vec_store && xo + 4 <= a.Wo) {
          *reinterpret_cast<f32x4*>(sp + o) = acc[m][n];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (xo + e < a.Wo) sp[o + e] = acc[m][n][e];
        }
      }
    }
    return;
  }
  const bool fast = a.fast_ok && co0 + G::COUT <= a.Cout && x0 + TW <= a.Wo && y0 + TH <= a.Ho;
  const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
  // GEN: activations that need a transcendental (Mish, sigmoid, tanh), chosen per element by a uniform switch;
  // GATED: the ConvGRU operands (mul / blend) are present
  auto epilogue_fast = [&](auto genc, auto resc, auto gatedc) __attribute__((always_inline)) {
    constexpr bool GEN = decltype(genc)::value;
    constexpr bool RES = decltype(resc)::value;
    constexpr bool GATED = decltype(gatedc)::value;
    unsigned loff[NT];
    float sc[NT], bi[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = co0 + n * 16 + j;
      sc[n] = a.ch_scale ? a.ch_scale[co] : 1.f;
      bi[n] = a.ch_bias ? a.ch_bias[co] : 0.f;
      loff[n] = (unsigned)(((size_t)co * oplane + x0 + 4 * kq) * sizeof(float));
    }
    const size_t bbase_o = (size_t)b * a.Cout * oplane;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const size_t rowo = bbase_o + (size_t)(y0 + wave * RPW + r) * a.Wo;    // scalar
      char* orow = reinterpret_cast<char*>(a.out + rowo);
      const char* rrow = reinterpret_cast<const char*>(RES ? a.residual + rowo : nullptr);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4 rv[MTX];
        if (RES) {
#pragma unroll
          for (int xt = 0; xt < MTX; ++xt) rv[xt] = *reinterpret_cast<const f32x4*>(rrow + loff[n] + xt * 64);
        }
#pragma unroll
        for (int xt = 0; xt < MTX; ++xt) {
          f32x4 v = acc[r * MTX + xt][n] * sc[n] + bi[n];
          if (RES) v += rv[xt];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = GEN ? dv_act(v[e], a.act) : fmaxf(v[e], v[e] * slope);
          if (GATED) {
            const size_t go = rowo * sizeof(float) + loff[n] + xt * 64;
            if (a.mul) v *= *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.mul) + go);
            if (a.blend_z) {
              const f32x4 z = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.blend_z) + go);
              const f32x4 h = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.blend_h) + go);
              v = h + z * (v - h);
            }
          }
          *reinterpret_cast<f32x4*>(orow + loff[n] + xt * 64) = v;
        }
      }
    }
  };
  if (fast) {
    const bool gen = a.act == DV_ACT_MISH || a.act == DV_ACT_SIGMOID || a.act == DV_ACT_TANH;
    auto go = [&](auto genc, auto resc) __attribute__((always_inline)) {
      if (a.mul || a.blend_z) epilogue_fast(genc, resc, std::true_type{});
      else epilogue_fast(genc, resc, std::false_type{});
    };
    if (gen) {
      if (a.residual) go(std::true_type{}, std::true_type{});
      else go(std::true_type{}, std::false_type{});
    } else {
      if (a.residual) go(std::false_type{}, std::true_type{});
      else go(std::false_type{}, std::false_type{});
    }
    return;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = co0 + n * 16 + j;
    if (co >= a.Cout) continue;
    const float sc = a.ch_scale ? a.ch_scale[co] : 1.f;
    const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
    const size_t cbase = ((size_t)b * a.Cout + co) * oplane;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int yo = y0 + wave * RPW + m / MTX, xo = x0 + (m % MTX) * 16 + 4 * kq;
      if (yo >= a.Ho || xo >= a.Wo) continue;
      const size_t o = cbase + (size_t)yo * a.Wo + xo;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (xo + e < a.Wo) {
          float u = fmaf(acc[m][n][e], sc, bi);
          if (a.residual) u += a.residual[o + e];
          u = dv_act(u, a.act);
          if (a.mul) u *= a.mul[o + e];
          if (a.blend_z) u = a.blend_h[o + e] + a.blend_z[o + e] * (u - a.blend_h[o + e]);
          a.out[o + e] = u;
        }
    }
  }
}

// packed weights: [chunk = ci / KC][co block][tap][quad q][lane = kq*16 + j][VW floats], where element
// e = ks * NT + n of the lane's B vector lives at quad e / VW, slot e % VW;  ci = chunk*KC + ks*4 + kq,
// co = block*COUT + n*16 + j.  Exactly the LDS image, so staging is a straight 16-byte copy.
__global__ void pack_conv2d_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout,
                                           int nchunk, int nco, int T, int KC, int NT) {
  const int NKS = KC / 4, BV = NKS * NT, VW = BV < 4 ? BV : 4, Q = BV / VW;
  const size_t total = (size_t)nchunk * nco * T * Q * 64 * VW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int slot = (int)(r % VW); r /= VW;
    const int lane = (int)(r % 64); r /= 64;
    const int q = (int)(r % Q); r /= Q;
    const int tap = (int)(r % T); r /= T;
    const int tc = (int)(r % nco);
    const int c = (int)(r / nco);
    const int e = q * VW + slot, ks = e / NT, n = e - ks * NT;
    const int kq = lane >> 4, j = lane & 15;
    const int ci = c * KC + ks * 4 + kq, co = (tc * NT + n) * 16 + j;
    wpk[i] = (ci < Cin && co < Cout) ? w[((size_t)co * Cin + ci) * T + tap] : 0.f;
  }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// geometry choice: N tiles per block by Cout, chunk depth by what fits two blocks per CU
// (NT = 4 would halve the staging traffic of the wide layers but needs 128 accumulator registers: it spills)
inline int nt_of(int Cout) { return Cout > 16 ? 2 : 1; }
inline int kc_of(int k, int /*dil*/) { return k == 3 ? 4 : 8; }

// sum of the K-split partials (fixed slice order: deterministic) + the fused epilogue of conv2d_mfma_kernel
__global__ __launch_bounds__(256) void conv2d_ksplit_epilogue_kernel(const float* __restrict__ scratch, int kslices,
                                                                     size_t total, int Cout, size_t oplane,
                                                                     const float* __restrict__ ch_scale,
                                                                     const float* __restrict__ ch_bias,
                                                                     const float* __restrict__ residual,
                                                                     const float* __restrict__ mul,
                                                                     const float* __restrict__ blend_z,
                                                                     const float* __restrict__ blend_h,
                                                                     float* __restrict__ out, int act) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float v = scratch[i];
  for (int sl = 1; sl < kslices; ++sl) v += scratch[(size_t)sl * total + i];
  const int co = (int)((i / oplane) % (size_t)Cout);
  v = fmaf(v, ch_scale ? ch_scale[co] : 1.f, ch_bias ? ch_bias[co] : 0.f);
  if (residual) v += residual[i];
  v = dv_act(v, act);
  if (mul) v *= mul[i];
  if (blend_z) v = blend_h[i] + blend_z[i] * (v - blend_h[i]);
  out[i] = v;
}

template <class G>
int launch2d(Conv2dArgs a, hipStream_t s) {
  a.ntx = (a.Wo + G::TW - 1) / G::TW;
  a.nty = (a.Ho + G::TH - 1) / G::TH;
  a.nco = pad_to(a.Cout, G::COUT) / G::COUT;
  const int ks = a.kslices > 1 ? a.kslices : 1;
  const long long blocks = (long long)a.B * a.nco * a.nty * a.ntx * ks;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  if (ks > 1) {
    if ((a.Cin + G::KC - 1) / G::KC < ks) return DV_ERR_UNSUPPORTED;       // at least one chunk per slice
    a.vec_store = a.vec_store && dv_aligned16(a.scratch);
    a.fast_ok = a.fast_ok && dv_aligned16(a.scratch);
  }
  hipLaunchKernelGGL((conv2d_mfma_kernel<G>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  int rc = dv_launch_status();
  if (rc != DV_OK || ks == 1) return rc;
  const size_t oplane = (size_t)a.Ho * a.Wo, total = (size_t)a.B * a.Cout * oplane;
  hipLaunchKernelGGL(conv2d_ksplit_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.scratch, ks,
                     total, a.Cout, oplane, a.ch_scale, a.ch_bias, a.residual, a.mul, a.blend_z, a.blend_h, a.out, a.act);
  return dv_launch_status();
}

// Slices the library uses for a stride-1 3x3 (dilation <= 4) or 1x1 launch of this size: 1 unless the launch has
// fewer than 256 blocks of the small-problem tile (4 x 32 pixels x 32 channels) and at least 8 chunks per slice.
// The K-split factor decides the order an output is summed in, so it looks at ONE batch item (round 5: a shard of a batch
// has to reproduce the batch's bits; `B` is kept in the signature and ignored).
inline int auto_kslices(int /*B*/, int Cin, int H, int W, int Cout, int k, int dilation) {
  if (k != 3 || dilation > 4 || Cout < 32) return 1;
  const long long blocks = (long long)((H + 3) / 4) * ((W + 31) / 32) * ((Cout + 31) / 32);
  if (blocks >= 256) return 1;
  const int nchunk = (Cin + 3) / 4;
  int ks = (int)(1024 / blocks);
  if (ks > 8) ks = 8;
  while (ks > 1 && nchunk / ks < 8) --ks;
  return ks < 1 ? 1 : ks;
}

}  // namespace

extern "C" size_t dv_conv2d_packed_floats(int Cin, int Cout, int k, int dilation) {
  if (Cin <= 0 || Cout <= 0 || (k != 1 && k != 3)) return 0;
  const int KC = kc_of(k, dilation), NT = nt_of(Cout);
  return (size_t)pad_to(Cin, KC) * (k * k) * pad_to(Cout, NT * 16);
}

extern "C" int dv_conv2d_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout, int k, int dilation,
                                          dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(k == 1 || k == 3, DV_ERR_UNSUPPORTED);
  const int KC = kc_of(k, dilation), NT = nt_of(Cout);
  const int nchunk = pad_to(Cin, KC) / KC, nco = pad_to(Cout, NT * 16) / (NT * 16);
  const size_t total = (size_t)nchunk * nco * (k * k) * KC * NT * 16;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv2d_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpacked, Cin,
                     Cout, nchunk, nco, k * k, KC, NT);
  return dv_launch_status();
}

static int conv2d_run(const float* in, const float* const* more, const int* more_channels, int n_more,
                      const float* wpacked, const float* ch_scale, const float* ch_bias,
                      const float* residual, const float* mul, const float* blend_z, const float* blend_h, float* out,
                      int B, int Cin, int H, int W, int Cout, int k, int dilation, int stride, int act,
                      dv_stream_t stream, float* scratch = nullptr, int kslices = 1) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(k == 1 || k == 3, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(k == 1 || (dilation >= 1 && dilation <= 16), DV_ERR_UNSUPPORTED);
  DV_REQUIRE(stride == 1 || (stride == 2 && (k == 1 || dilation == 1)), DV_ERR_UNSUPPORTED);
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_TANH, DV_ERR_UNSUPPORTED);
  DV_REQUIRE((blend_z == nullptr) == (blend_h == nullptr), DV_ERR_NULL);
  DV_REQUIRE(dv_aligned16(wpacked), DV_ERR_ALIGN);
  DV_REQUIRE((size_t)H * W * sizeof(float) <= 0x7fffffffull, DV_ERR_SHAPE);
  Conv2dArgs a;
  a.in = in;
  {   // Cin counts all sources; the first one owns what the others do not
    int rest = 0;
    for (int i = 0; i < n_more; ++i) rest += more_channels[i];
    DV_REQUIRE(n_more >= 0 && n_more <= 3 && rest < Cin, DV_ERR_SHAPE);
    a.cend[0] = Cin - rest;
    for (int i = 0; i < 3; ++i) {
      a.in_more[i] = i < n_more ? more[i] : nullptr;
      a.cend[i + 1] = a.cend[i] + (i < n_more ? more_channels[i] : 0);
      if (i < n_more) { DV_REQUIRE_PTR(more[i]); DV_REQUIRE(more_channels[i] > 0, DV_ERR_SHAPE); }
    }
    if (n_more < 3) for (int i = n_more + 1; i < 4; ++i) a.cend[i] = Cin;
  }
  a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.residual = residual; a.out = out;
  a.mul = mul; a.blend_z = blend_z; a.blend_h = blend_h;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.dil = k == 1 ? 0 : dilation; a.act = act;
  a.Ho = (H - 1) / stride + 1;          // 'same' padding: pad = dilation (k 3) / 0 (k 1)
  a.Wo = (W - 1) / stride + 1;
  a.vec_store = (a.Wo % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual)) &&
                (!mul || dv_aligned16(mul)) && (!blend_z || (dv_aligned16(blend_z) && dv_aligned16(blend_h)));
  a.fast_ok = a.vec_store && (size_t)Cout * a.Ho * a.Wo * sizeof(float) <= 0xffffffffull;
  a.ntx = a.nty = a.nco = 0;
  a.scratch = scratch;
  a.kslices = kslices;
  if (kslices > 1) {
    DV_REQUIRE_PTR(scratch);
    DV_REQUIRE(stride == 1 && kslices == auto_kslices(B, Cin, H, W, Cout, k, dilation), DV_ERR_UNSUPPORTED);
  }
  hipStream_t s = (hipStream_t)stream;
  const int NT = nt_of(Cout);
  //                              KS NT KC DMAX BANDED RPW [WPS MTX S]
  if (stride == 2) {
    if (k == 1) return NT == 2 ? launch2d<G2<1, 2, 8, 0, false, 2, 2, 4, 2>>(a, s) : launch2d<G2<1, 1, 8, 0, false, 2, 2, 4, 2>>(a, s);
    return NT == 2 ? launch2d<G2<3, 2, 4, 1, false, 2, 2, 4, 2>>(a, s) : launch2d<G2<3, 1, 4, 1, false, 2, 3, 4, 2>>(a, s);
  }
  if (k == 1) {
    if (NT == 2) return launch2d<G2<1, 2, 8, 0, false, 2>>(a, s);
    return launch2d<G2<1, 1, 8, 0, false, 2>>(a, s);
  }
  // 3x3: chunks of 4 channels and <= 170 registers, so three blocks share a CU (measured 109 vs 100 TFLOP/s for
  // 8-channel chunks at two blocks per CU: a third resident wave per SIMD hides the staging phases better than
  // longer MFMA runs do)
  if (dilation <= 4) {
    if (NT == 2) {
      // 64-column tiles unless 32-column ones use the chip's 768 block slots (3 per CU) clearly better: small
      // images (IGEV's 1/4 .. 1/16 resolution GRUs) otherwise run one and a bit rounds of blocks
      const long long rows = (long long)B * ((H + 7) / 8) * ((Cout + 31) / 32);
      const long long b64 = rows * ((W + 63) / 64), b32 = rows * ((W + 31) / 32);
      auto eff = [](long long blocks) { return (double)blocks / (double)((blocks + 767) / 768 * 768); };
      // very small problems (a single IGEV pair: 480 blocks of 8 x 32 pixels) would leave most of the chip idle
      // behind long serial K loops: 4 x 32-pixel tiles double the block count (16 accumulator registers, many
      // resident waves per SIMD).  (Staging 16 channels per barrier round on top of that was slower.)
      if (b32 < 1024) return launch2d<G2<3, 2, 4, 4, false, 1, 4, 2>>(a, s);
      if (eff(b32) > 1.15 * eff(b64)) return launch2d<G2<3, 2, 4, 4, false, 2, 3, 2>>(a, s);
      return launch2d<G2<3, 2, 4, 4, false, 2, 3>>(a, s);
    }
    {   // <= 16 output channels (heads with one channel): the same rule on the block count
      const long long rows1 = (long long)B * ((H + 7) / 8) * ((Cout + 15) / 16);
      if (rows1 * ((W + 31) / 32) < 1024) return launch2d<G2<3, 1, 4, 4, false, 1, 4, 2>>(a, s);
    }
    return launch2d<G2<3, 1, 4, 4, false, 2, 3>>(a, s);
  }
  if (NT == 2) return launch2d<G2<3, 2, 4, 16, true, 2, 2>>(a, s);   // 9 staged positions per thread: spills at 170 registers
  return launch2d<G2<3, 1, 4, 16, true, 2, 3>>(a, s);
}

extern "C" int dv_conv2d_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                             const float* residual, float* out, int B, int Cin, int H, int W, int Cout, int k,
                             int dilation, int act, dv_stream_t stream) {
  return conv2d_run(in, nullptr, nullptr, 0, wpacked, ch_scale, ch_bias, residual, nullptr, nullptr, nullptr, out, B, Cin, H,
                    W, Cout, k, dilation, 1, act, stream);
}

extern "C" int dv_conv2d_gated_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                                   const float* residual, const float* mul, const float* blend_z, const float* blend_h,
                                   float* out, int B, int Cin, int H, int W, int Cout, int k, int dilation, int act,
                                   dv_stream_t stream) {
  return conv2d_run(in, nullptr, nullptr, 0, wpacked, ch_scale, ch_bias, residual, mul, blend_z, blend_h, out, B, Cin, H, W,
                    Cout, k, dilation, 1, act, stream);
}

extern "C" int dv_conv2d_s2_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                                const float* residual, float* out, int B, int Cin, int H, int W, int Cout, int k,
                                int act, dv_stream_t stream) {
  return conv2d_run(in, nullptr, nullptr, 0, wpacked, ch_scale, ch_bias, residual, nullptr, nullptr, nullptr, out, B, Cin, H,
                    W, Cout, k, 1, 2, act, stream);
}

extern "C" int dv_conv2d_auto_kslices(int B, int Cin, int H, int W, int Cout, int k, int dilation) {
  if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 1;
  return auto_kslices(B, Cin, H, W, Cout, k, dilation);
}

extern "C" int dv_conv2d_cat_ksplit_f32(const float* const* inputs, const int* channels, int n_inputs,
                                        const float* wpacked, const float* ch_scale, const float* ch_bias,
                                        const float* residual, const float* mul, const float* blend_z,
                                        const float* blend_h, float* out, float* scratch, int kslices, int B, int H, int W,
                                        int Cout, int k, int dilation, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(inputs);
  DV_REQUIRE_PTR(channels);
  DV_REQUIRE(n_inputs >= 1 && n_inputs <= 4, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(kslices >= 2 && kslices <= 8, DV_ERR_UNSUPPORTED);
  int cin = 0;
  for (int i = 0; i < n_inputs; ++i) {
    DV_REQUIRE(channels[i] > 0, DV_ERR_SHAPE);
    cin += channels[i];
  }
  return conv2d_run(inputs[0], inputs + 1, channels + 1, n_inputs - 1, wpacked, ch_scale, ch_bias, residual, mul, blend_z,
                    blend_h, out, B, cin, H, W, Cout, k, dilation, 1, act, stream, scratch, kslices);
}

extern "C" int dv_conv2d_cat_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                                 const float* ch_scale, const float* ch_bias, const float* residual, const float* mul,
                                 const float* blend_z, const float* blend_h, float* out, int B, int H, int W, int Cout,
                                 int k, int dilation, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(inputs);
  DV_REQUIRE_PTR(channels);
  DV_REQUIRE(n_inputs >= 1 && n_inputs <= 4, DV_ERR_UNSUPPORTED);
  int cin = 0;
  for (int i = 0; i < n_inputs; ++i) {
    DV_REQUIRE(channels[i] > 0, DV_ERR_SHAPE);
    cin += channels[i];
  }
  return conv2d_run(inputs[0], inputs + 1, channels + 1, n_inputs - 1, wpacked, ch_scale, ch_bias, residual, mul, blend_z,
                    blend_h, out, B, cin, H, W, Cout, k, dilation, 1, act, stream);
}
