// K3f: the FIRST aggregation layer of a DiffuVolume step on its factored input.
//
// Every DDIM step of SceneFlow/models/acv_ddim.py:254-262 feeds `volume * noise` to dres0[0] =
// convbn_3d(64, 32, 3, 1, 1) + ReLU (:200-203), where (:388-390)
//     volume[b, c, d, y, x] = p(b, d, y, x) * L[b, c, y, x]              c <  C   (left half: the same for every d)
//                           = p(b, d, y, x) * R[b, c - C, y, x - d]      c >= C   (right half, 0 for x < d)
// with p = softmax(att_weights, dim=2), and noise = n01(b, d, y, x) is a per-voxel scalar too (:256-258).  So the
// layer's input is  s(b, d, y, x) * [ L(y, x) ; R(y, x - d) ]  with s = p * n01, and the 3x3x3 convolution factors:
//
//   out[co, d, y, x] = sum_{tap = (td, ty, tx)}  s(d', y', x') * ( GL[tap][co](y', x') + GR[tap][co](y', x' - d') )
//                      with (d', y', x') = (d + td - 1, y + ty - 1, x + tx - 1), zero outside the volume,
//   GL[tap][co](y, x) = sum_c W[co, c,     tap] L[c, y, x]        GR[tap][co](y, u) = sum_c W[co, C + c, tap] R[c, y, u]
//                                                                  (0 for u < 0: the zero wedge of the right half)
//
// GL / GR are 1x1 convolutions of the two feature maps (27*Cout output channels each), built ONCE per stereo pair
// (they do not depend on the DDIM step); per step the layer is then 2 * 27 multiply-adds per output instead of
// 27 * 2C -- 32x fewer for C = 32 -- and reads a [B,D,H,W] scalar field instead of the 3 GB volume.  It is the exact
// same function (validated to 9e-15 in float64); in fp32 it is a re-association of the sums like any other tiling.
// The reference's layer is 163 of the 751 GFLOP of a step; here it is bound by the vector ALU (108 flop
// per output).  BN scale / bias and the activation are applied before the store, as in the conv kernels.

#include <type_traits>

#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int R1_DMAX = 48;      // disparities of a quarter-resolution volume (maxdisp 192: hard-coded in the reference)

struct Rank1Args {
  const float* s;        // [B, D, H, W]      p * n01
  const float* gl;       // [B, 27 * Cout, H, W]   channel = tap * Cout + co
  const float* gr;       // [B, 27 * Cout, H, W]
  const float* ch_scale; // [Cout] or null
  const float* ch_bias;  // [Cout] or null
  float* out;            // [B, Cout, D, H, W]
  int B, D, H, W, Cout;
  int ntx, ncg;          // x tiles per row, pairs of output channels
  int act;
};

// ---- the kernel ---------------------------------------------------------------------------------------------------------
// The GL term wants a thread that keeps x fixed while it walks d (GL[tap](y', x') is then constant: registers); the GR
// term wants a thread that keeps u = x - d fixed (GR[tap](y', x' - d') = GR[tap](y', u + tx - td) is then constant too).
// So a block = one image row of one pair of output channels runs both walks side by side: waves 0-3 ("diagonal") own
// one u each and step along (d, x = u + d), waves 4-7 ("straight") own one x each; both keep their 27 x 2 table values
// in registers and issue 27 packed fmas per step.  What they read in the loop is s only, and they read it from LDS: the
// block stages the three rows of six planes at a time (one barrier per six steps; global loads whose range check is
// the zero padding, issued a round ahead), a step consumes ONE plane -- its nine positions feed output e + 1 through
// the td = 0 taps, output e through td = 1 and output e - 1 through td = 2 (three rotating accumulators) -- so nothing
// but the current plane's 15 / 9 values is ever held.  The diagonal waves leave their finished sums in a small LDS ring;
// the straight waves add theirs a round later, apply BN + the activation and store.
// History (round 3): a first form kept the GR rows of all 27 taps in LDS ([tap][u][2 co], 65 KB) and walked d with one
// thread per x: one ds_read_b64 + one packed add per fma, 1.22 ms per launch at B = 8.  This form has a third of the LDS
// reads and no add: 0.95 ms.  A version of it that read s from global memory in both walks was bound by its vector
// memory instructions (24 per output step and wave pair, about one per 8 cycles per CU): 1.28 ms.
constexpr int R1S_G = 6;                       // planes per round (a multiple of the 3-step accumulator rotation)
constexpr int R1S_COLS = 256;                  // threads per role = ring row stride
constexpr int R1S_SROW = 272;                  // staged columns x0 - 3 .. x0 + 268
constexpr int R1S_STAGE = R1S_G * 3 * R1S_SROW;
constexpr int R1S_NS = (R1S_STAGE + 511) / 512;

template <int HI>
__device__ __forceinline__ void r1_mac(f32x2& acc, const f32x2& g, const f32x2& sv) {
  // acc += g * (the HI-th half of sv, broadcast to both channels)
  if constexpr (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(g), "v"(sv));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(g), "v"(sv));
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void rank1_filter_split_kernel(Rank1Args a, int xt) {
  __shared__ __attribute__((aligned(16))) f32x2 ring[2 * R1S_G * R1S_COLS];      // finished GR sums, slot = d mod 12
  __shared__ __attribute__((aligned(16))) float s_ring[2 * R1S_STAGE];           // [round & 1][plane 6][row 3][col]
  const int tid = threadIdx.x & 255;
  const bool diag = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) == 0;
  unsigned t = blockIdx.x;
  const int cg = t % a.ncg; t /= a.ncg;
  const int tx = t % a.ntx; t /= a.ntx;
  const int y = t % a.H;
  const int b = t / a.H;
  const int x0 = tx * xt;
  const int xlim = a.W - x0 < xt ? a.W - x0 : xt;          // columns of this tile
  const int co0 = cg * 2;
  const int nch = a.Cout - co0 < 2 ? a.Cout - co0 : 2;
  const size_t plane = (size_t)a.H * a.W;
  const int plane_b = __builtin_amdgcn_readfirstlane((int)(plane * sizeof(float)));
  const int row_b = __builtin_amdgcn_readfirstlane(a.W * 4);
  const int tab_bytes = __builtin_amdgcn_readfirstlane((int)((size_t)27 * a.Cout * plane * sizeof(float)));
  auto scalar64 = [](const void* p) __attribute__((always_inline)) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    return (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
           ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
  };
  // steps e = 0 .. D (plane D is all zero and completes output D - 1), in rounds of six
  const int nrounds = (a.D + 1 + R1S_G - 1) / R1S_G;

  // ---- staging of s: element i of a round = (plane i / (3 SROW), row (i / SROW) % 3, col i % SROW) ----
  const auto s_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(scalar64(a.s + (size_t)b * a.D * plane)), 0,
                                                      a.D * plane_b, 0x00020000);
  int soff[R1S_NS];
  float sv[R1S_NS];
#pragma unroll
  for (int i = 0; i < R1S_NS; ++i) {
    const int idx = (int)threadIdx.x + 512 * i;
    const int pl = idx / (3 * R1S_SROW), r2 = idx - pl * (3 * R1S_SROW);
    const int row = r2 / R1S_SROW, col = r2 - row * R1S_SROW;
    const int yy = y + row - 1, xx = x0 - 3 + col;
    const bool ok = idx < R1S_STAGE && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
    soff[i] = ok ? pl * plane_b + (yy * a.W + xx) * 4 : (int)0x80000000u;    // + first plane of the round (scalar)
  }
  auto stage_fetch = [&](int round) __attribute__((always_inline)) {
    const int so = round * R1S_G * plane_b;                // planes >= D fall off the descriptor: zeros
#pragma unroll
    for (int i = 0; i < R1S_NS; ++i) sv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s_rs, soff[i], so, 0));
  };
  auto stage_commit = [&](int round) __attribute__((always_inline)) {
    float* dst = s_ring + (round & 1) * R1S_STAGE;
#pragma unroll
    for (int i = 0; i < R1S_NS; ++i) {
      const int idx = (int)threadIdx.x + 512 * i;
      if (idx < R1S_STAGE) dst[idx] = sv[i];
    }
  };
  stage_fetch(0);

  if (diag) {
    // ---------------- diagonal walk: u fixed, x = u + d ----------------
    const int umin = x0 == 0 ? -2 : x0 - (a.D - 1);
    const int u = umin + tid;
    const auto gr_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(scalar64(a.gr + ((size_t)b * 27 * a.Cout) * plane)), 0, tab_bytes, 0x00020000);
    f32x2 gr[27];
    {
      int goff[5];                                        // column u + k - 2, k = tx - td + 2
#pragma unroll
      for (int k = 0; k < 5; ++k) goff[k] = (unsigned)(u + k - 2) < (unsigned)a.W ? (u + k - 2) * 4 : (int)0x80000000u;
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int td = tap / 9, ty = (tap / 3) % 3, tx3 = tap % 3;
        const int yy = y + ty - 1;
        const bool yok = (unsigned)yy < (unsigned)a.H;
        const int so = yok ? (tap * a.Cout + co0) * plane_b + yy * row_b : 0;
        const int vo = yok ? goff[tx3 - td + 2] : (int)0x80000000u;
        f32x2 v = {0.f, 0.f};
        v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr_rs, vo, so, 0));
        if (nch > 1) v[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr_rs, vo, so + plane_b, 0));
        gr[tap] = v;
      }
    }
    stage_commit(0);
    // every table load has landed before the loop: otherwise the first use of a table register inside it carries a
    // vmcnt(n) computed for the first trip, which on every later trip waits for that round's freshly issued stage loads
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    __syncthreads();
    // plane e: this lane reads columns u + e - 2 .. u + e + 2 = staged columns cb .. cb + 4, cb = u + e + 1 - x0.
    // Lanes whose outputs lie outside the tile read whatever is there (clamped into the row): their sums are dropped.
    auto step = [&](int e, const float* pl, f32x2& accP, f32x2& accC, f32x2& accN) __attribute__((always_inline)) {
      int cb = u + e + 1 - x0;
      cb = cb < 0 ? 0 : (cb > R1S_SROW - 5 ? R1S_SROW - 5 : cb);
      const float* sp = pl + cb;
      f32x2 w[3][3];                                       // a row's five values as pairs (k, k + 1): no cross-row packing
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
#pragma unroll
        for (int k = 0; k < 5; ++k) w[ty][k >> 1][k & 1] = sp[ty * R1S_SROW + k];
      }
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx3 = 0; tx3 < 3; ++tx3) {
          const int k0 = tx3 + 2, k1 = tx3 + 1, k2 = tx3;                                  // k = tx - td + 2
          if (k0 & 1) r1_mac<1>(accN, gr[ty * 3 + tx3], w[ty][k0 >> 1]); else r1_mac<0>(accN, gr[ty * 3 + tx3], w[ty][k0 >> 1]);
          if (k1 & 1) r1_mac<1>(accC, gr[9 + ty * 3 + tx3], w[ty][k1 >> 1]); else r1_mac<0>(accC, gr[9 + ty * 3 + tx3], w[ty][k1 >> 1]);
          if (k2 & 1) r1_mac<1>(accP, gr[18 + ty * 3 + tx3], w[ty][k2 >> 1]); else r1_mac<0>(accP, gr[18 + ty * 3 + tx3], w[ty][k2 >> 1]);
        }
      const int d = e - 1, col = u + d - x0;
      if (d >= 0 && (unsigned)col < (unsigned)xlim) ring[(d % (2 * R1S_G)) * R1S_COLS + col] = accP;
      accP = (f32x2){0.f, 0.f};
    };
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll 1
    for (int r = 0; r < nrounds; ++r) {
      if (r + 1 < nrounds) stage_fetch(r + 1);
      const float* base = s_ring + (r & 1) * R1S_STAGE;
      const int e = r * R1S_G;
      step(e, base, a0, a1, a2);
      step(e + 1, base + 3 * R1S_SROW, a1, a2, a0);
      step(e + 2, base + 6 * R1S_SROW, a2, a0, a1);
      step(e + 3, base + 9 * R1S_SROW, a0, a1, a2);
      step(e + 4, base + 12 * R1S_SROW, a1, a2, a0);
      step(e + 5, base + 15 * R1S_SROW, a2, a0, a1);
      if (r + 1 < nrounds) stage_commit(r + 1);
      __syncthreads();
    }
    return;
  }

  // ---------------- straight walk: x fixed ----------------
  const int x = x0 + tid;
  const bool xin = tid < xlim;
  const auto gl_rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<float*>(scalar64(a.gl + ((size_t)b * 27 * a.Cout) * plane)), 0, tab_bytes, 0x00020000);
  f32x2 gl[27];
  {
    int roff[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int yy = y + q / 3 - 1, xx = x + q % 3 - 1;
      roff[q] = (xin && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? (yy * a.W + xx) * 4 : (int)0x80000000u;
    }
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int so = (tap * a.Cout + co0) * plane_b;
      f32x2 v = {0.f, 0.f};
      v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gl_rs, roff[tap % 9], so, 0));
      if (nch > 1) v[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gl_rs, roff[tap % 9], so + plane_b, 0));
      gl[tap] = v;
    }
  }
  float sc[2] = {1.f, 1.f}, bi[2] = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c < nch) {
      if (a.ch_scale) sc[c] = a.ch_scale[co0 + c];
      if (a.ch_bias) bi[c] = a.ch_bias[co0 + c];
    }
  const size_t vol = (size_t)a.D * plane;
  const int vol_bytes = __builtin_amdgcn_readfirstlane((int)(vol * sizeof(float)));
  const auto ors = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<float*>(scalar64(a.out + ((size_t)b * a.Cout + co0) * vol)), 0, nch * vol_bytes, 0x00020000);
  const int ooff = xin ? (y * a.W + x) * 4 : (int)0x80000000u;
  const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
  const bool mish = a.act == DV_ACT_MISH;
  stage_commit(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): as in the diagonal walk
  __syncthreads();
  // plane e: columns x - 1 .. x + 1 = staged columns tid + 2 .. tid + 4
  f32x2 done[R1S_G];                                      // the GL sums of the outputs finished in a round
  auto step = [&](const float* pl, f32x2& accP, f32x2& accC, f32x2& accN, f32x2& fin) __attribute__((always_inline)) {
    const float* sp = pl + tid + 2;
    f32x2 w[3][2];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
#pragma unroll
      for (int k = 0; k < 3; ++k) w[ty][k >> 1][k & 1] = sp[ty * R1S_SROW + k];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int ty = q / 3, k = q % 3;
      if (k & 1) {
        r1_mac<1>(accN, gl[q], w[ty][k >> 1]);
        r1_mac<1>(accC, gl[9 + q], w[ty][k >> 1]);
        r1_mac<1>(accP, gl[18 + q], w[ty][k >> 1]);
      } else {
        r1_mac<0>(accN, gl[q], w[ty][k >> 1]);
        r1_mac<0>(accC, gl[9 + q], w[ty][k >> 1]);
        r1_mac<0>(accP, gl[18 + q], w[ty][k >> 1]);
      }
    }
    fin = accP;
    accP = (f32x2){0.f, 0.f};
  };
  // outputs d = e - 1 of round r - 1 (e = 6 (r - 1) + j): GL sum from `done`, GR sum from the ring
  auto finish_as = [&](int r, auto mishc) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < R1S_G; ++j) {
      const int d = (r - 1) * R1S_G + j - 1;
      if (d < 0 || d >= a.D) continue;
      // (x - d < -2: the whole right half is in its zero wedge, no diagonal thread exists for it)
      const f32x2 rv = ring[(d % (2 * R1S_G)) * R1S_COLS + tid];
      const f32x2 acc = done[j] + (x - d >= -2 ? rv : (f32x2){0.f, 0.f});
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float v = fmaf(acc[c], sc[c], bi[c]);
        v = decltype(mishc)::value ? dv_act(v, DV_ACT_MISH) : fmaxf(v, v * slope);
        // an odd Cout has no second channel in its last pair: skipped by a wave-uniform test (the descriptor's range
        // check is not relied on here -- the scalar offset enters it as num_records - soffset, which would wrap)
        if (c < nch)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ors, ooff, c * vol_bytes + d * plane_b, 0);
      }
    }
  };
  auto finish = [&](int r) __attribute__((always_inline)) {
    if (mish) finish_as(r, std::true_type{});
    else finish_as(r, std::false_type{});
  };
  f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll 1
  for (int r = 0; r < nrounds; ++r) {
    if (r + 1 < nrounds) stage_fetch(r + 1);
    if (r >= 1) finish(r);
    const float* base = s_ring + (r & 1) * R1S_STAGE;
    step(base, a0, a1, a2, done[0]);
    step(base + 3 * R1S_SROW, a1, a2, a0, done[1]);
    step(base + 6 * R1S_SROW, a2, a0, a1, done[2]);
    step(base + 9 * R1S_SROW, a0, a1, a2, done[3]);
    step(base + 12 * R1S_SROW, a1, a2, a0, done[4]);
    step(base + 15 * R1S_SROW, a2, a0, a1, done[5]);
    if (r + 1 < nrounds) stage_commit(r + 1);
    __syncthreads();
  }
  finish(nrounds);
}

// p = softmax over D of att [B, D, HW] with the arithmetic of concat_rows_kernel (max, sum of dv_exp_le0, one
// reciprocal per pixel): the same bits as the attention product inside the fused concat builder
__global__ void softmax_d_kernel(const float* __restrict__ att, float* __restrict__ p, int D, size_t hw, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / hw, px = i - b * hw;
    const float* a = att + b * D * hw + px;
    float mx = a[0];
    for (int d = 1; d < D; ++d) mx = fmaxf(mx, a[(size_t)d * hw]);
    float sum = 0.f;
    for (int d = 0; d < D; ++d) sum += dv_exp_le0(a[(size_t)d * hw] - mx);
    const float rs = 1.f / sum;
    float* o = p + b * D * hw + px;
    for (int d = 0; d < D; ++d) o[(size_t)d * hw] = dv_exp_le0(a[(size_t)d * hw] - mx) * rs;
  }
}

__global__ void mul2_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = x[i] * y[i];
}

}  // namespace

extern "C" int dv_softmax_d_f32(const float* att, float* p, int B, int D, int HW, dv_stream_t stream) {
  DV_REQUIRE_PTR(att);
  DV_REQUIRE_PTR(p);
  DV_REQUIRE(B > 0 && D > 0 && HW > 0, DV_ERR_SHAPE);
  const size_t total = (size_t)B * HW;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(softmax_d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, att, p, D, (size_t)HW, total);
  return dv_launch_status();
}

extern "C" int dv_mul_f32(const float* x, const float* y, float* out, size_t n, dv_stream_t stream) {
  DV_REQUIRE_PTR(x);
  DV_REQUIRE_PTR(y);
  DV_REQUIRE_PTR(out);
  if (n == 0) return DV_OK;
  const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  hipLaunchKernelGGL(mul2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, out, n);
  return dv_launch_status();
}

extern "C" int dv_conv3d_rank1_filter_f32(const float* s, const float* gl, const float* gr, const float* ch_scale,
                                          const float* ch_bias, float* out, int B, int D, int H, int W, int Cout, int act,
                                          dv_stream_t stream) {
  DV_REQUIRE_PTR(s);
  DV_REQUIRE_PTR(gl);
  DV_REQUIRE_PTR(gr);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_LEAKY, DV_ERR_UNSUPPORTED);
  Rank1Args a;
  a.s = s; a.gl = gl; a.gr = gr; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.out = out;
  a.B = B; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
  DV_REQUIRE(D <= R1_DMAX, DV_ERR_UNSUPPORTED);               // a tile's diagonal range x - d has to fit 256 lanes
  DV_REQUIRE((size_t)27 * Cout * H * W * sizeof(float) <= 0x7fffffffull, DV_ERR_SHAPE);    // 31-bit offsets in a table
  DV_REQUIRE((size_t)Cout * D * H * W * sizeof(float) <= 0x7fffffffull && (size_t)D * H * W * sizeof(float) <= 0x7fffffffull,
             DV_ERR_SHAPE);
  // 256 diagonal + 256 straight threads per (row, pair of channels); x tiles whose u = x - d range fits the diagonal lanes
  const int xt = W + 2 <= R1S_COLS ? W : ((R1S_COLS - (D - 1)) & ~3);
  a.ntx = (W + xt - 1) / xt;
  a.ncg = (Cout + 1) / 2;
  const long long nb = (long long)B * H * a.ntx * a.ncg;
  if (nb <= 0 || nb > 0x7fffffffLL) return DV_ERR_SHAPE;
  hipLaunchKernelGGL(rank1_filter_split_kernel, dim3((unsigned)nb), dim3(512), 0, (hipStream_t)stream, a, xt);
  return dv_launch_status();
}
