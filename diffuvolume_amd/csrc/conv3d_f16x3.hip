// K4': the 3x3x3 stride-1 aggregation convolution with fp32 operands carried as TWO fp16 numbers each.
//
// Same layer and fused epilogue as conv3d.hip (convbn_3d + ReLU, SceneFlow/models/submodule.py:94-97;
// `volume * noise` prologue acv_ddim.py:260), but the MFMA stream runs on v_mfma_f32_16x16x32_f16 at
// 16x the fp32-MFMA rate:  every fp32 activation / weight x is split exactly as x = hi + lo + O(2^-22 |x|),
// hi = fp16(x), lo = fp16(x - hi), and   x*w  ~=  hi*hi' + hi*lo' + lo*hi'   (three MFMAs, fp32 accumulate;
// fp16 x fp16 products are exact in fp32).  The dropped lo*lo' term and the 22-bit split are below the
// rounding of the fp32 accumulation itself: on the 26-layer stack the split alone gives 3.5e-7 relative
// error against float64, the reference's own fp32 CPU path 1.4e-6.  Weights are
// pre-scaled by 2^10 and activations by 2^-2 (exact) so that lo parts stay out of the fp16 subnormal range
// and activations up to 2.6e5 cannot overflow; the accumulator is rescaled by 2^-8 in the epilogue.
//
// GEMM view: M = 16 consecutive x of a (z,y) row, N = 16 cout, K-block (32) = 4 taps x 8 input channels:
// lane (i = lane&15, kq = lane>>4) supplies the 8 channels of voxel i at tap slot 4*kb+kq as ONE
// ds_read_b128 from the channels-last LDS image [hi|lo][position][8 halves].  The 27 taps (+1 dummy with
// zero weights) are paired so that the two taps sharing a 16-lane LDS group sit a multiple of 16 positions
// apart (row stride RW = 8n, (dz,dz+1) and (dy,dy+2) pairs): 13 of the 14 pairs are bank-conflict free.
// Per chunk of 8 channels the fp32 brick is fetched one chunk ahead into registers (as conv3d.hip), then
// split and committed to LDS; 2 blocks per CU.
#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int TD = 2, TH = 4, MTX = 3, NT = 2, KC = 8;
constexpr int TW = MTX * 16;
constexpr int IZ = TD + 2, IY = TH + 2, IX = TW + 2;
constexpr int RW = (IX + 7) / 8 * 8;                 // 56: IY*RW and 2*RW are multiples of 16 positions
constexpr int NPOS = IZ * IY * RW;                   // LDS positions (16 B each, per part)
constexpr int NREAL = IZ * IY * IX;                  // positions that carry data
constexpr int KB = 7;                                // k-blocks of 4 tap slots
constexpr int COUT = NT * 16;
constexpr int W_H8 = KB * 4 * COUT;                  // h8 vectors per weight part and chunk
constexpr int ROWS = TD * TH, RPW = ROWS / 4, MT = RPW * MTX;
static_assert((IY * RW) % 16 == 0 && (2 * RW) % 16 == 0, "tap pairing needs 16-position multiples");
static_assert(2 * (NPOS + W_H8) * 16 <= 80 * 1024, "two blocks per CU");

constexpr float kActScale = 0.25f, kWScale = 1024.0f, kOutScale = 1.0f / (0.25f * 1024.0f);

// tap slot -> (dz,dy,dx); slot 27 repeats a tap with zero weights
__host__ __device__ constexpr int slot_tap(int s) {
  // pairs: 9 x [(0,dy,dx),(1,dy,dx)], 3 x [(2,0,dx),(2,2,dx)], [(2,1,0),(2,1,1)], [(2,1,2), dummy]
  if (s < 18) { const int p = s >> 1, dz = s & 1; return (dz * 3 + p / 3) * 3 + p % 3; }
  if (s < 24) { const int p = (s - 18) >> 1, dy = ((s - 18) & 1) * 2; return (2 * 3 + dy) * 3 + p; }
  if (s == 24) return (2 * 3 + 1) * 3 + 0;
  if (s == 25) return (2 * 3 + 1) * 3 + 1;
  return (2 * 3 + 1) * 3 + 2;  // 26 real, 27 dummy
}
__host__ __device__ constexpr int tap_off(int tap) { return ((tap / 9) * IY + (tap / 3) % 3) * RW + tap % 3; }

struct Args {
  const float* in;
  const _Float16* wpk;   // [Cin/8][part 2][KB][4][Coutp][8]
  const float* ch_scale;
  const float* ch_bias;
  const float* in_scale;
  const float* residual;
  float* out;
  int* overflow;         // set to 1 when an activation leaves the fp16 range (may be null)
  int B, Cin, D, H, W, Cout, Coutp;
  int ntx, nty, ntz, nco;
  int act, vec_store;
};

template <bool HAS_SCALE>
__global__ __launch_bounds__(256, 2) void conv3d_f16x3_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) h8 smem[2 * NPOS + 2 * W_H8];
  h8* in_s = smem;                 // [part][pos]
  h8* w_s = smem + 2 * NPOS;       // [part][kb][kq][cout]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;

  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty; t /= a.nty;
  const int tz = t % a.ntz; t /= a.ntz;
  const int tc = t % a.nco;
  const int b = t / a.nco;
  const int x0 = tx * TW, y0 = ty * TH, z0 = tz * TD, co0 = tc * COUT;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this wave's rows are (2*wave, 2*wave+1) of the brick and never straddle a z slab (TH even), so
  // every M-tile is a compile-time offset from one base position
  static_assert(RPW == 2 && TH % 2 == 0, "row layout assumed by a_off()");
  const int abase0 = (((wave * RPW) / TH) * IY + (wave * RPW) % TH) * RW + j;
  int toff[KB];     // position offset of this lane's tap in every k-block
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int o0 = tap_off(slot_tap(4 * kb)), o1 = tap_off(slot_tap(4 * kb + 1));
    const int o2 = tap_off(slot_tap(4 * kb + 2)), o3 = tap_off(slot_tap(4 * kb + 3));
    toff[kb] = kq == 0 ? o0 : (kq == 1 ? o1 : (kq == 2 ? o2 : o3));
  }
  const int bbase = kq * COUT + j;

  const size_t plane = (size_t)a.H * a.W, vol = (size_t)a.D * plane;
  const float* inb = a.in + (size_t)b * a.Cin * vol;
  const float* scb = (HAS_SCALE && a.in_scale) ? a.in_scale + (size_t)b * vol : nullptr;

  // staging plan: thread owns NS brick positions (all 8 channels of each)
  constexpr int NS = (NREAL + 255) / 256;
  constexpr int NWQ = (2 * W_H8 + 255) / 256;
  unsigned sob[NS];                 // byte offset inside one channel volume (2^31 outside: the buffer load returns 0)
  float scl[HAS_SCALE ? NS : 1];    // activation scale x `volume * noise` factor
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int r = tid + 256 * i;
    const int zz = r / (IY * IX), r2 = r - zz * (IY * IX);
    const int yy = r2 / IX, xx = r2 - yy * IX;
    const int z = z0 - 1 + zz, y = y0 - 1 + yy, x = x0 - 1 + xx;
    const bool ok = r < NREAL && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H &&
                    (unsigned)x < (unsigned)a.W;
    const unsigned sp = ok ? (unsigned)((z * a.H + y) * a.W + x) : 0u;
    sob[i] = ok ? sp * 4u : 0x80000000u;
    if (HAS_SCALE) scl[i] = (ok && scb) ? scb[sp] * kActScale : kActScale;
  }
  const int vol_bytes = (int)(vol * sizeof(float));   // < 2^31 (checked by the host)
  float vin[NS][KC];
  f32x4 vw[NWQ];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int cl = 0; cl < KC; ++cl) {
      // buffer loads (one descriptor per channel, scalar): zero padding and channel tail from the range check
      const bool cok = (c0 + cl) < a.Cin;
      const uint64_t ba = reinterpret_cast<uint64_t>(inb + (size_t)(cok ? c0 + cl : 0) * vol);
      const uint64_t bu = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba) |
                          ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ba >> 32)) << 32);
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(bu), 0,
                                                        __builtin_amdgcn_readfirstlane(cok ? vol_bytes : 0), 0x00020000);
#pragma unroll
      for (int i = 0; i < NS; ++i)
        vin[i][cl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)sob[i], 0, 0));
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk) + (size_t)(c0 / KC) * 2 * KB * 4 * a.Coutp;
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = tid + 256 * q;                         // -> (part, kb*4+kq, cout)
      const int part = e / W_H8, r = e - part * W_H8;
      const int sl = r / COUT, n = r - sl * COUT;
      if (e < 2 * W_H8) vw[q] = wsrc[((size_t)part * KB * 4 + sl) * a.Coutp + co0 + n];
    }
  };
  float vmax = 0.f;   // largest scaled magnitude this thread has split (range guard)
  auto commit = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int r = tid + 256 * i;
      if (r >= NREAL) continue;
      const int zz = r / (IY * IX), r2 = r - zz * (IY * IX);
      const int lpos = (zz * IY + r2 / IX) * RW + r2 % IX;
      h8 hi, lo;
#pragma unroll
      for (int cl = 0; cl < KC; ++cl) {
        // explicit, un-contracted roundings: if the compiler fused the multiply into the subtraction
        // (fma) it would round hi from fl32(x*s) but lo against the exact product, and at fp16 ties the
        // pair would be one fp16 ulp apart (seen once per ~10^4 elements before this was pinned)
        // (hipcc fused it even through __fmul_rn/__fsub_rn: v_fma_mixlo_f16 for lo, v_cvt_pk_f16_f32 of
        // the rounded product for hi -- the empty asm makes v opaque so the two cannot be re-derived.)
        float v = vin[i][cl] * (HAS_SCALE ? scl[i] : kActScale);
        asm volatile("" : "+v"(v));
        vmax = fmaxf(vmax, fabsf(v));
        const _Float16 h = (_Float16)v;
        hi[cl] = h;
        lo[cl] = (_Float16)(v - (float)h);
      }
      in_s[lpos] = hi;
      in_s[NPOS + lpos] = lo;
    }
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      const int e = tid + 256 * q;
      if (e < 2 * W_H8) reinterpret_cast<f32x4*>(w_s)[e] = vw[q];
    }
  };

  fetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += KC) {
    __syncthreads();
    commit(c0);
    __syncthreads();
    if (c0 + KC < a.Cin) fetch(c0 + KC);
    // MFMA stream.  (A hand-pipelined variant -- A fragments two M-tiles ahead through a register ring,
    // issue order pinned with sched_group_barrier -- was 35 % SLOWER: the kernel is bound by vector-ALU
    // issue slots (fp32->fp16 split, addressing) shared with the MFMA issue, not by LDS latency, and the
    // ring cost 50 spilled registers.  See DESIGN.md 5b.)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      h8 bh[NT], bl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bh[n] = w_s[bbase + kb * 4 * COUT + n * 16];
        bl[n] = w_s[W_H8 + bbase + kb * 4 * COUT + n * 16];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int p = abase0 + (m / MTX) * RW + (m % MTX) * 16 + toff[kb];
        const h8 ah = in_s[p], al = in_s[NPOS + p];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], acc[m][n], 0, 0, 0);
        }
      }
    }
  }

  // range guard: fp16 tops out at 65504; NaN inputs also trip it (fmaxf ignores NaN, so test v != v via !(<=))
  if (a.overflow && !(vmax <= 65000.f)) *a.overflow = 1;

  // epilogue (as conv3d.hip): descale, BN scale/bias, residual, activation, 16-B stores along x
  const size_t ovol = vol;  // stride 1: output volume == input volume
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = co0 + n * 16 + j;
    if (co >= a.Cout) continue;
    const float sc = (a.ch_scale ? a.ch_scale[co] : 1.f) * kOutScale;
    const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
    const size_t cbase = ((size_t)b * a.Cout + co) * ovol;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int rr = wave * RPW + m / MTX, xt = m % MTX;
      const int zo = z0 + rr / TH, yo = y0 + rr % TH, xo = x0 + xt * 16 + 4 * kq;
      if (zo >= a.D || yo >= a.H || xo >= a.W) continue;
      const size_t o = cbase + (size_t)zo * plane + (size_t)yo * a.W + xo;
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
          if (xo + r < a.W) {
            float u = v[r];
            if (a.residual) u += a.residual[o + r];
            a.out[o + r] = dv_act(u, a.act);
          }
      }
    }
  }
}

// w [Cout][Cin][27] fp32 -> [Cin/8][part][KB][4][Coutp][8] fp16 (hi / lo of w * 2^10; dummy slot = 0)
__global__ void pack_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ wpk, int Cin, int Cout,
                                  int nchunk, int Coutp) {
  const size_t per_part = (size_t)KB * 4 * Coutp * 8;
  const size_t total = (size_t)nchunk * 2 * per_part;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int jj = (int)(i % 8);
    size_t r = i / 8;
    const int n = (int)(r % Coutp); r /= Coutp;
    const int sl = (int)(r % (KB * 4)); r /= (KB * 4);
    const int part = (int)(r % 2);
    const int chunk = (int)(r / 2);
    const int ci = chunk * 8 + jj;
    float v = 0.f;
    if (sl < 27 && ci < Cin && n < Cout) v = w[((size_t)n * Cin + ci) * 27 + slot_tap(sl)] * kWScale;
    const _Float16 h = (_Float16)v;
    wpk[i] = part == 0 ? h : (_Float16)(v - (float)h);
  }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" size_t dv_conv3d_f16x3_packed_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)pad_to(Cin, 8) / 8 * 2 * KB * 4 * pad_to(Cout, COUT) * 8 * sizeof(_Float16);
}

extern "C" int dv_conv3d_f16x3_pack_weights(const float* w, void* wpacked, int Cin, int Cout, dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  const int nchunk = pad_to(Cin, 8) / 8, Coutp = pad_to(Cout, COUT);
  const size_t total = (size_t)nchunk * 2 * KB * 4 * Coutp * 8;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_f16x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)wpacked, Cin,
                     Cout, nchunk, Coutp);
  return dv_launch_status();
}

extern "C" int dv_conv3d_f16x3_f32(const float* in, const void* wpacked, const float* ch_scale, const float* ch_bias,
                                   const float* in_scale, const float* residual, float* out, int* overflow_flag,
                                   int B, int Cin, int D, int H, int W, int Cout, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE((size_t)D * H * W * sizeof(float) <= 0x7fffffffull, DV_ERR_SHAPE);   // 31-bit byte offsets in a channel
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_LEAKY, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(dv_aligned16(wpacked), DV_ERR_ALIGN);
  Args a;
  a.in = in; a.wpk = (const _Float16*)wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.in_scale = in_scale;
  a.residual = residual; a.out = out; a.overflow = overflow_flag;
  a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.Coutp = pad_to(Cout, COUT);
  a.ntx = (W + TW - 1) / TW; a.nty = (H + TH - 1) / TH; a.ntz = (D + TD - 1) / TD; a.nco = a.Coutp / COUT;
  a.act = act;
  a.vec_store = (W % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual));
  const long long blocks = (long long)B * a.nco * a.ntz * a.nty * a.ntx;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (in_scale)
    hipLaunchKernelGGL(conv3d_f16x3_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(conv3d_f16x3_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return dv_launch_status();
}
