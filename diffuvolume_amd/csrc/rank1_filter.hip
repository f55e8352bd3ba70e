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
// per output) and by LDS reads of GR.
//
// Kernel: a block = one image row y of one (batch item, group of 4 output channels), all disparities.  Thread = one
// x; it keeps its 27 x 4 GL values in registers (they do not depend on d) and the 3 x 3 x 3 neighbourhood of s as a
// rolling window over d (9 loads per step).  GR depends on x - d: the three source rows of every tap sit in LDS as
// [tap][u][4 co] (16 bytes per u: one conflict-free ds_read_b128 per tap and step, consecutive lanes read consecutive
// slots), zero for u < 0.  BN scale / bias and the activation are applied before the store, as in the conv kernels.

#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N> struct VecOf;
template <> struct VecOf<4> { typedef f32x4 type; };
template <> struct VecOf<2> { typedef f32x2 type; };

constexpr int R1_XT = 256;       // x positions per block (one per thread)
constexpr int R1_DMAX = 48;      // disparities of a quarter-resolution volume (maxdisp 192: hard-coded in the reference)
#ifndef R1_CPT
#define R1_CPT 2
#endif
constexpr int R1_UW = R1_XT + R1_DMAX + 3;   // LDS slots per tap: u = x - d + (tx - td) over a block's x and all d

struct Rank1Args {
  const float* s;        // [B, D, H, W]      p * n01
  const float* gl;       // [B, 27 * Cout, H, W]   channel = tap * Cout + co
  const float* gr;       // [B, 27 * Cout, H, W]
  const float* ch_scale; // [Cout] or null
  const float* ch_bias;  // [Cout] or null
  float* out;            // [B, Cout, D, H, W]
  int B, D, H, W, Cout;
  int ntx, ncg;          // x tiles per row, groups of CPT output channels
  int act;
};

// CPT = output channels per thread (and per block): 4 -> 130 KB of LDS, one block (one wave per SIMD) per CU;
// 2 -> 65 KB, two blocks per CU and half the registers per thread
template <int CPT>
__global__ __launch_bounds__(R1_XT, CPT == 4 ? 1 : 2) void rank1_filter_conv_kernel(Rank1Args a) {
  typedef typename VecOf<CPT>::type vec;
  constexpr int UW = R1_UW;
  __shared__ __attribute__((aligned(16))) float gr_s[27 * R1_UW * CPT];   // [27][UW][CPT co]
  const int tid = threadIdx.x;
  unsigned t = blockIdx.x;
  const int cg = t % a.ncg; t /= a.ncg;      // the channel groups of a row side by side: they share s and the row's lines
  const int tx = t % a.ntx; t /= a.ntx;
  const int y = t % a.H;
  const int b = t / a.H;
  const int x0 = tx * R1_XT, x = x0 + tid;
  const int co0 = cg * CPT;
  const size_t plane = (size_t)a.H * a.W;
  const float* glb = a.gl + ((size_t)b * 27 * a.Cout) * plane;
  const float* grb = a.gr + ((size_t)b * 27 * a.Cout) * plane;
  const float* sb = a.s + (size_t)b * a.D * plane;

  // ---- GR rows -> LDS: slot ui <-> u = x0 - (D - 1) - 2 + ui; value 0 for u outside [0, W) or y' outside the image.
  // Buffer loads (descriptor = this batch item's table, lane offset 2^31 where the value is 0 by definition), nine
  // taps = 18 x CPT independent loads in flight before the first LDS store: the fill is a latency chain otherwise
  // (one L2 round trip per slot), and it is a fifth of the block's memory instructions.
  const int umin = x0 - (a.D - 1) - 2;
  constexpr int NK = (UW + R1_XT - 1) / R1_XT;
  const int tab_bytes = __builtin_amdgcn_readfirstlane((int)((size_t)27 * a.Cout * plane * sizeof(float)));   // < 2^31 (host check)
  const int plane_b = __builtin_amdgcn_readfirstlane((int)(plane * sizeof(float)));
  auto table_rsrc = [&](const float* base) __attribute__((always_inline)) {
    const uint64_t p64 = reinterpret_cast<uint64_t>(base);
    const uint64_t ps = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p64) |
                        ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p64 >> 32)) << 32);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(ps), 0, tab_bytes, 0x00020000);
  };
  const auto gr_rs = table_rsrc(grb);
  int uoff[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int u = umin + tid + k * R1_XT;
    uoff[k] = (tid + k * R1_XT < UW && (unsigned)u < (unsigned)a.W) ? u * 4 : (int)0x80000000u;
  }
#pragma unroll 1
  for (int t9 = 0; t9 < 27; t9 += 9) {
    float v[9][NK][CPT];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int tap = t9 + i;
      const int yy = y + (tap / 3) % 3 - 1;
      const bool yok = (unsigned)yy < (unsigned)a.H;
      // scalar part: channel (tap * Cout + co0 + c) and row yy; rows outside the image: the invalid lane offset
      const int so = yok ? (tap * a.Cout + co0) * plane_b + yy * a.W * 4 : 0;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int vo = yok ? uoff[k] : (int)0x80000000u;
#pragma unroll
        for (int c = 0; c < CPT; ++c)
          v[i][k][c] = (co0 + c < a.Cout)
                           ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr_rs, vo, so + c * plane_b, 0))
                           : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int ui = tid + k * R1_XT;
        if (ui < UW) {
          vec q;
#pragma unroll
          for (int c = 0; c < CPT; ++c) q[c] = v[i][k][c];
          reinterpret_cast<vec*>(gr_s)[(t9 + i) * UW + ui] = q;
        }
      }
  }

  // ---- GL: 27 x CPT registers, constant over d (branch-free buffer loads like the fill above)
  const auto gl_rs = table_rsrc(glb);
  vec gl[27];
  {
    int goff[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int yy = y + q / 3 - 1, xx = x + q % 3 - 1;
      goff[q] = ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? (yy * a.W + xx) * 4 : (int)0x80000000u;
    }
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int so = (tap * a.Cout + co0) * plane_b;
      vec v = {};
#pragma unroll
      for (int c = 0; c < CPT; ++c)
        if (co0 + c < a.Cout)
          v[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gl_rs, goff[tap % 9], so + c * plane_b, 0));
      gl[tap] = v;
    }
  }

  // ---- rolling 3 x 3 x 3 window of s: plane p of the window = s(d + p - 1, y + ty - 1, x + tx - 1), q = ty*3 + tx.
  // Buffer loads: one descriptor per batch item, a constant 32-bit lane offset per neighbour (2^31 outside the image:
  // the range check returns 0), the plane as a scalar offset; planes outside [0, D) get a zero-record descriptor.
  // No branch, no 64-bit vector arithmetic in the loop (one wave per SIMD: nothing would hide them).
  int roff[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int yy = y + q / 3 - 1, xx = x + q % 3 - 1;
    roff[q] = ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? (yy * a.W + xx) * 4 : (int)0x80000000u;
  }
  const int plane_bytes = __builtin_amdgcn_readfirstlane((int)(plane * sizeof(float)));
  const uint64_t sp64 = reinterpret_cast<uint64_t>(sb);
  const uint64_t sps = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sp64) |
                       ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sp64 >> 32)) << 32);
  auto load_plane = [&](int d, f32x2* dst) __attribute__((always_inline)) {
    const bool dok = (unsigned)d < (unsigned)a.D;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(sps), 0, dok ? a.D * plane_bytes : 0, 0x00020000);
    const int so = dok ? d * plane_bytes : 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) dst[q][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, roff[q], so, 0));
  };
  // (a window value lives in the low half of a register pair: the packed fma below broadcasts it to both channels
  // through op_sel instead of a move per tap)
  f32x2 sw[3][9];
#pragma unroll
  for (int p3 = 0; p3 < 3; ++p3)
#pragma unroll
    for (int q = 0; q < 9; ++q) sw[p3][q] = (f32x2){0.f, 0.f};       // plane -1 stays zero
  load_plane(0, sw[1]);
  load_plane(1, sw[2]);

  float sc[CPT], bi[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    sc[c] = 1.f;
    bi[c] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < CPT; ++c)
    if (co0 + c < a.Cout) {
      if (a.ch_scale) sc[c] = a.ch_scale[co0 + c];
      if (a.ch_bias) bi[c] = a.ch_bias[co0 + c];
    }
  __syncthreads();

  // output: one descriptor over the 4 channel volumes of this block, lane offset (y*W + x)*4 (2^31 for x >= W: the
  // store is dropped), channel and plane as a scalar offset
  const size_t vol = (size_t)a.D * plane;
  const int vol_bytes = __builtin_amdgcn_readfirstlane((int)(vol * sizeof(float)));
  const int nch = a.Cout - co0 < CPT ? a.Cout - co0 : CPT;
  const uint64_t op64 = reinterpret_cast<uint64_t>(a.out + ((size_t)b * a.Cout + co0) * vol);
  const uint64_t ops = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)op64) |
                       ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(op64 >> 32)) << 32);
  const auto ors = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(ops), 0, nch * vol_bytes, 0x00020000);
  const int ooff = x < a.W ? (y * a.W + x) * 4 : (int)0x80000000u;
  const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
  const bool mish = a.act == DV_ACT_MISH;
  // LDS slot of tap (td, ., tx) at step d for this lane: u = x - d + (tx - td)  ->  ui = tid + (D - 1) + 2 - d + tx - td
  const vec* grq = reinterpret_cast<const vec*>(gr_s) + tid + (a.D - 1) + 2;
  // one step: planes (pa, pb, pc) of the window are (d - 1, d, d + 1); plane d + 2 is requested into pa's registers
  // once they have been read (three steps per trip: the window rotates through its three register sets, no moves)
  auto mac = [&](vec& acc, const f32x2& sv, const vec& g) __attribute__((always_inline)) {
    if constexpr (CPT == 2) {
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(g), "v"(sv));
    } else {
      acc += sv[0] * g;
    }
  };
  auto step = [&](int d, f32x2* pa, f32x2* pb, f32x2* pc) __attribute__((always_inline)) {
    // all 27 LDS reads of the step first (two waves per SIMD: latencies are hidden by issuing early, not by other
    // waves), three independent accumulation chains, the next plane of s requested as soon as plane d - 1 is consumed
    const vec* gq = grq - d;
    vec g[27];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      g[q] = gq[q * UW + q % 3];
      g[9 + q] = gq[(9 + q) * UW + q % 3 - 1];
      g[18 + q] = gq[(18 + q) * UW + q % 3 - 2];
    }
    vec acc0 = {}, acc1 = {}, acc2 = {};
#pragma unroll
    for (int q = 0; q < 9; ++q) mac(acc0, pa[q], gl[q] + g[q]);
    load_plane(d + 2, pa);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      mac(acc1, pb[q], gl[9 + q] + g[9 + q]);
      mac(acc2, pc[q], gl[18 + q] + g[18 + q]);
    }
    const vec acc = acc0 + acc1 + acc2;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      float v = fmaf(acc[c], sc[c], bi[c]);
      v = mish ? dv_act(v, DV_ACT_MISH) : fmaxf(v, v * slope);
      if (c < nch) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ors, ooff, c * vol_bytes + d * plane_bytes, 0);
    }
  };
#pragma unroll 1
  for (int d = 0; d < a.D; d += 3) {
    step(d, sw[0], sw[1], sw[2]);
    if (d + 1 < a.D) step(d + 1, sw[1], sw[2], sw[0]);
    if (d + 2 < a.D) step(d + 2, sw[2], sw[0], sw[1]);
  }
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
  constexpr int CPT = R1_CPT;
  a.ntx = (W + R1_XT - 1) / R1_XT;
  a.ncg = (Cout + CPT - 1) / CPT;
  DV_REQUIRE(D <= R1_DMAX, DV_ERR_UNSUPPORTED);               // the LDS image is sized for 48 disparities
  DV_REQUIRE((size_t)27 * Cout * H * W * sizeof(float) <= 0x7fffffffull, DV_ERR_SHAPE);    // 31-bit offsets in a table
  DV_REQUIRE((size_t)Cout * D * H * W * sizeof(float) <= 0x7fffffffull && (size_t)D * H * W * sizeof(float) <= 0x7fffffffull,
             DV_ERR_SHAPE);
  const long long blocks = (long long)B * H * a.ntx * a.ncg;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  hipLaunchKernelGGL(rank1_filter_conv_kernel<CPT>, dim3((unsigned)blocks), dim3(R1_XT), 0, (hipStream_t)stream, a);
  return dv_launch_status();
}
