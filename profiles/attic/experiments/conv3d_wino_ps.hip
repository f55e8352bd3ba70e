// K4p: the Winograd F(2x2,3x3) form of the 3x3x3 stride-1 aggregation convolution (see conv3d_wino.hip for the
// algebra: V = Bt d B per 4x4 input patch, M[p] += V[p] U[p] over depth taps and input channels on
// v_mfma_f32_16x16x4_f32, Y = At M A; convbn_3d, SceneFlow/models/submodule.py:94-97, the dres / hourglass /
// classifier layers of acv_ddim.py:60-70, :200-222) as a PRODUCER / CONSUMER kernel.
//
// conv3d_wino.hip gives every wave the whole job (stage, transform, multiply, store) and runs two such blocks per
// CU; its matrix pipe is busy 66 % of the time: each plane is transformed by the three waves that use it, the
// staging / transform instructions of a wave sit in its own MFMA stream, and the per-block prologue and epilogue are
// covered only by the one other block of the CU.  Here a block is 8 waves with two jobs:
//   * waves 0-3 (consumers, one per SIMD) own the 128 accumulator registers of one output plane each and issue
//     nothing but ds_read_b128 (A quads from V, B quads from U) and MFMAs: 36 LDS reads per 96 MFMAs;
//   * waves 4-7 (producers, one per SIMD, one input channel of the chunk each) fetch the haloed raw brick of their
//     channel with buffer loads (zero padding and the channel tail from the hardware range check), keep it in a
//     wave-private LDS region, transform each of the 6 planes ONCE (v_pk_add_f32) and write V -- laid out exactly
//     like the weight image, so the A fragment of a transform position quad is one conflict-free ds_read_b128 --
//     and copy the packed weights of the chunk by LDS-DMA.
// V and U are double buffered: the producers prepare chunk m+1 while the consumers multiply chunk m; one block
// barrier per chunk.  Blocks are persistent (one per CU, 111 KB of LDS): a block walks its tiles in the XCD-slab
// order of the other conv kernels, and the producers run ahead across tile boundaries, so the first loads and the
// first transform of the next tile -- and the consumers' own output transform and stores -- are off the matrix
// pipe's critical path except for the epilogue itself.
//
// Same packed weights, same arithmetic and the same accumulation order as conv3d_wino.hip: bit-identical results.

#include <type_traits>

#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// SHAPE = how the 16 Winograd tiles of a plane's M index lie in the plane (as in conv3d_wino.hip): 0 -> 2 tile rows x
// 8 tile columns (4 x 16 outputs), 1 -> 4 x 4 (8 x 8 outputs), 2 -> 8 x 2 (16 x 4 outputs).
template <int SHAPE_>
struct PG {
  static constexpr int SHAPE = SHAPE_;
  static constexpr int TR = SHAPE == 0 ? 2 : (SHAPE == 1 ? 4 : 8), TC = 16 / TR;
  static constexpr int KC = 4, NT = 2, TD = 4, TH = 2 * TR, TW = 2 * TC;
  static constexpr int IZ = TD + 2, IY = TH + 2, IX = TW + 2;
  static constexpr int PRAW = IZ * IY * IX;          // raw positions per channel (648 / 600 / 648)
  // raw brick of ONE channel [z][y][RX] with plane stride PZ.  Bank plan of the producers' patch reads: a 32-lane half
  // of a ds_read_b64 holds the 16 tiles of two planes; inside a plane the tile columns (2 floats apart) and tile
  // rows (2*RX apart) tile 32 of the 64 banks without overlap (RX as in conv3d_wino.hip), and PZ = 32 (mod 64) puts
  // the second plane on the other 32.
  static constexpr int RX = SHAPE == 0 ? 24 : (SHAPE == 1 ? 12 : 6);
  static constexpr int PZ = 160;
  static constexpr int RAWC = IZ * PZ;               // floats per channel region (wave-private)
  static constexpr int NSP = (PRAW + 63) / 64;       // raw positions staged per producer lane
  static constexpr int DUMP = IY * RX;               // where lanes past the brick put their value: plane 0's padding
  static_assert(IY * RX < PZ && PZ % 64 == 32, "plane stride");
  static_assert(RX >= IX && RX % 2 == 0, "row stride");
};
namespace pg {
constexpr int U_CHUNK = 3 * 2 * 4 * 16 * 16;   // packed weight floats per (chunk, cout block): the LDS image, 24 KB
constexpr int V_PLANE = 4 * 16 * 16;           // [kq 4][tile 16][position 16]
constexpr int V_BUF = 6 * V_PLANE;             // 6 transformed planes of a chunk, 24 KB
}  // namespace pg

#ifdef DV_PS_STAMPS
// diagnostic build only (tools/wino_ps_bench.cpp -DDV_PS_STAMPS): s_memtime stamps of block 0, consumer wave 0 and
// producer wave 4, first 48 items; written to a buffer nothing else reads
__device__ unsigned long long g_ps_stamps[2][48][8];
#define PS_STAMP(who, item, slot)                                                              \
  do {                                                                                         \
    if (blockIdx.x == 0 && (item) < 48 && lane == 0) g_ps_stamps[who][item][slot] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define PS_STAMP(who, item, slot) do {} while (0)
#endif

struct WinoPsArgs {
  const float* in;
  const float* wpk;      // [Cin/4][Coutp/32][kd 3][nt 2][k 4][n 16][pos 16]   (dv_conv3d_wino_pack_weights_f32)
  const float* ch_scale;
  const float* ch_bias;
  const float* in_scale; // [B,D,H,W] or null
  const float* residual;
  float* out;
  int B, Cin, D, H, W, Cout;
  int ntx, nty, ntz, nco;
  int ntiles;
  int act;
  int fast_ok;           // W % 4 == 0, 16-byte aligned pointers
};

struct TileXY {
  int tc, x0, y0, z0, b;
};

template <typename G>
__device__ __forceinline__ TileXY decode_tile(const WinoPsArgs& a, unsigned t) {
  TileXY r;
  r.tc = t % a.nco; t /= a.nco;
  r.x0 = (t % a.ntx) * G::TW; t /= a.ntx;
  r.y0 = (t % a.nty) * G::TH; t /= a.nty;
  r.z0 = (t % a.ntz) * G::TD;
  r.b = t / a.ntz;
  return r;
}

template <bool HAS_SCALE, int SHAPE>
__global__ __launch_bounds__(512, 1) void conv3d_wino_ps_kernel(WinoPsArgs a) {
  using G = PG<SHAPE>;
  constexpr int KC = G::KC, NT = G::NT, TD = G::TD, TH = G::TH, TW = G::TW, IY = G::IY, IX = G::IX, PRAW = G::PRAW;
  constexpr int RX = G::RX, PZ = G::PZ, RAWC = G::RAWC, NSP = G::NSP;
  constexpr int U_CHUNK = pg::U_CHUNK, V_PLANE = pg::V_PLANE, V_BUF = pg::V_BUF;
  static_assert((2 * U_CHUNK + 2 * V_BUF + KC * RAWC) * 4 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) float smem[2 * U_CHUNK + 2 * V_BUF + KC * RAWC];
  float* const u_s = smem;
  float* const v_s = smem + 2 * U_CHUNK;
  float* const raw_s = smem + 2 * U_CHUNK + 2 * V_BUF;
  char* const smem_b = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- this block's tiles: XCD x (blocks b with b % 8 == x share an L2) owns a contiguous slab of the linear tile
  // order (cout slice fastest, then x, y, z, batch); its blocks take the slab's tiles round-robin, so the blocks of an
  // XCD work on neighbouring tiles at any time and share their halos (and the cout slices their whole brick) in L2 ----
  const unsigned nblk = gridDim.x, xcd = blockIdx.x & 7u, bidx = blockIdx.x >> 3;
  const unsigned nbx = (nblk >> 3) + (xcd < (nblk & 7u) ? 1u : 0u);                 // blocks of this XCD
  const unsigned tq = (unsigned)a.ntiles >> 3, trm = (unsigned)a.ntiles & 7u;
  const unsigned slab_lo = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq;
  const unsigned slab_n = tq + (xcd < trm ? 1u : 0u);
  const int nt_mine = bidx < slab_n ? (int)((slab_n - bidx + nbx - 1) / nbx) : 0;
  if (nt_mine == 0) return;
  const int nchunk = (a.Cin + KC - 1) / KC;
  const int M = nt_mine * nchunk;                                                   // (tile, chunk) items of this block
  auto tile_of = [&](int k) { return decode_tile<G>(a, slab_lo + bidx + (unsigned)k * nbx); };

  const size_t plane = (size_t)a.H * a.W;
  const size_t vol = (size_t)a.D * plane;
  const int j = lane & 15, kq = lane >> 4;
  constexpr int GC = G::TC / 2;                       // 2x2-tile groups per row of groups: 4 / 2 / 1

  if (wave < 4) {
    // =========================================== consumers ===========================================
    // lane (tile j, channel kq) of wave w: A quad of transform positions 4*p4 .. 4*p4+3 of plane w + kd = one
    // ds_read_b128 of V row (kq, j); B quad = the same row of U for cout j.  Slot s of a row holds quad s ^ (j >> 2):
    // the 16 lanes of a ds_read_b128 lane group then cover 16 distinct 16-byte slots of the 256-byte bank row.
#ifdef DV_PS_CONS_PRIO
    __builtin_amdgcn_s_setprio(DV_PS_CONS_PRIO);
#endif
    // Between two MFMAs of a stream the matrix pipe accepts another vector-ALU instruction only after it has drained:
    // an isolated v_or / v_add costs ~60 cycles (tools/probes/mfma_f32_neighbours.hip), so the chunk body contains
    // none -- every LDS address is a loop-invariant register plus an immediate, with the buffer index a template
    // parameter.
    // lane (tile j, channel kq) of wave w: A quad of transform positions 4*p4 .. 4*p4+3 of plane w + kd = one
    // ds_read_b128 of V row (kq, j); B quad = the same row of U for cout j.  The buffer flip of the eight address
    // registers is left to the compiler: it clusters the eight vector ops with the first MFMAs of a chunk and hoists
    // the block barrier over the last ~20 MFMAs of the previous one, so that the first LDS reads of a chunk are in
    // flight behind them (pinning the flip behind the barrier by inline asm lost that: consumer-only 2.74 vs 2.20 ms).
    int ab_lo[4];
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) ab_lo[p4] = 4 * ((kq * 16 + j) * 16 + ((p4 ^ ((j >> 2) & 3)) * 4));
    const int a_base = 4 * (2 * U_CHUNK + wave * V_PLANE);
    f32x4 acc[16][NT];

    auto compute = [&](int buf, auto first_tag) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const char* const vb = smem_b + a_base + buf * (4 * V_BUF);
      const char* const ub = smem_b + buf * (4 * U_CHUNK);
      f32x4 aq[2], bq[2][NT];
      auto load_ab = [&](int g, int slot) __attribute__((always_inline)) {
        const int kd = g >> 2, p4 = g & 3;
        aq[slot] = *reinterpret_cast<const f32x4*>(vb + ab_lo[p4] + kd * (4 * V_PLANE));
#pragma unroll
        for (int n = 0; n < NT; ++n)
          bq[slot][n] = *reinterpret_cast<const f32x4*>(ub + ab_lo[p4] + 4 * ((kd * NT + n) * (KC * 256)));
      };
      load_ab(0, 0);
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        const int kd = g >> 2, p4 = g & 3;
        if (g + 1 < 12) load_ab(g + 1, (g + 1) & 1);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (FIRST && kd == 0)      // the tile's first products start the accumulators (no 128 zero moves)
              acc[p4 * 4 + e][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[g & 1][e], bq[g & 1][n][e],
                                                                        (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else
              acc[p4 * 4 + e][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[g & 1][e], bq[g & 1][n][e],
                                                                        acc[p4 * 4 + e][n], 0, 0, 0);
          }
      }
    };
    auto compute_item = [&](int m, auto first_tag) __attribute__((always_inline)) { compute(m & 1, first_tag); };

    // epilogue of one tile: Y = At M A per 2x2-tile group, BN scale / bias, residual, activation, 16-byte stores.
    // Accumulator rows 4kq..4kq+3 = the 2x2 tiles of group kq = a 4 x 4 output patch at (4*(kq/GC), 4*(kq%GC)).
    const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
    const bool mish = a.act == DV_ACT_MISH;
    auto epilogue = [&](const TileXY& t) __attribute__((always_inline)) {
      const int xb = t.x0 + 4 * (kq % GC), yq = 4 * (kq / GC);
      const bool fast = a.fast_ok && t.x0 + TW <= a.W && t.y0 + TH <= a.H;
      const int zo = t.z0 + wave;
      if (zo >= a.D) return;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = t.tc * 32 + n * 16 + j;
        if (co >= a.Cout) continue;
        const float sc = a.ch_scale ? a.ch_scale[co] : 1.f;
        const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
        const size_t cbase = (((size_t)t.b * a.Cout + co) * a.D + zo) * plane + (size_t)(t.y0 + yq) * a.W + xb;
        f32x4 rv[4];
        if (fast && a.residual) {
#pragma unroll
          for (int r = 0; r < 4; ++r) rv[r] = *reinterpret_cast<const f32x4*>(a.residual + cbase + (size_t)r * a.W);
        }
#pragma unroll
        for (int tr = 0; tr < 2; ++tr) {       // tile row
          float yv[2][4];                      // two output rows x 4 x
#pragma unroll
          for (int tcx = 0; tcx < 2; ++tcx) {  // tile column 2kq + tcx = accumulator element i = tr + 2*tcx
            const int i = tr + 2 * tcx;
            float s0[4], s1[4];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
              const float m0 = acc[px][n][i], m1 = acc[4 + px][n][i], m2 = acc[8 + px][n][i], m3 = acc[12 + px][n][i];
              s0[px] = m0 + m1 + m2;
              s1[px] = m1 - m2 - m3;
            }
            yv[0][2 * tcx] = s0[0] + s0[1] + s0[2];
            yv[0][2 * tcx + 1] = s0[1] - s0[2] - s0[3];
            yv[1][2 * tcx] = s1[0] + s1[1] + s1[2];
            yv[1][2 * tcx + 1] = s1[1] - s1[2] - s1[3];
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int yr = 2 * tr + r;
            const size_t o = cbase + (size_t)yr * a.W;
            if (fast) {
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaf(yv[r][e], sc, bi);
              if (a.residual) v += rv[yr];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = mish ? dv_act(v[e], DV_ACT_MISH) : fmaxf(v[e], v[e] * slope);
#ifdef DV_PS_ABLATE_STORES
              asm volatile("" ::"v"(v));
#else
              *reinterpret_cast<f32x4*>(a.out + o) = v;
#endif
            } else if (t.y0 + yq + yr < a.H) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (xb + e < a.W) {
                  float u = fmaf(yv[r][e], sc, bi);
                  if (a.residual) u += a.residual[o + e];
                  a.out[o + e] = dv_act(u, a.act);
                }
            }
          }
        }
      }
    };

    __syncthreads();                               // item 0 is in V[0] / U[0]
    int m = 0;
#pragma unroll 1
    for (int k = 0; k < nt_mine; ++k) {
      if (wave == 0) PS_STAMP(0, m, 0);
#ifndef DV_PS_ABLATE_CONSUMER
      compute_item(m, std::true_type{});
#endif
      if (wave == 0) PS_STAMP(0, m, 1);
      __syncthreads();                             // done with buffers m & 1; item m + 1 is ready
      if (wave == 0) PS_STAMP(0, m, 2);
      ++m;
#pragma unroll 1
      for (int c = 1; c < nchunk; ++c) {
        if (wave == 0) PS_STAMP(0, m, 0);
#ifndef DV_PS_ABLATE_CONSUMER
        compute_item(m, std::false_type{});
#endif
        if (wave == 0) PS_STAMP(0, m, 1);
        __syncthreads();
        if (wave == 0) PS_STAMP(0, m, 2);
        ++m;
      }
#if !defined(DV_PS_ABLATE_CONSUMER) && !defined(DV_PS_ABLATE_EPILOGUE)
      epilogue(tile_of(k));                        // (the producers are already preparing the item after next)
#elif defined(DV_PS_ABLATE_EPILOGUE)
      if (a.act == 12345) epilogue(tile_of(k));    // timing-only build: keeps the accumulators alive, never runs
#endif
      if (wave == 0) PS_STAMP(0, m - 1, 3);
#ifdef DV_PS_STAMPS
      if (wave == 0 && blockIdx.x == 0 && lane == 0) { g_ps_stamps[0][47][7] = __builtin_amdgcn_s_memtime(); g_ps_stamps[0][47][6] = (unsigned long long)m; }
      if (wave == 0 && blockIdx.x == 0 && lane == 0 && k == 0) g_ps_stamps[0][47][5] = __builtin_amdgcn_s_memrealtime();
      if (wave == 0 && blockIdx.x == 0 && lane == 0) g_ps_stamps[0][47][4] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    return;
  }

  // ============================================= producers =============================================
#ifdef DV_PS_PROD_PRIO
  __builtin_amdgcn_s_setprio(DV_PS_PROD_PRIO);
#endif
  const int p = wave - 4;                          // this wave's channel of every chunk; V / U rows kq = p
  const int vol_bytes = __builtin_amdgcn_readfirstlane((int)(vol * sizeof(float)));   // < 2^31 (checked by the host)
  const int pl = kq;
  const int p_tr = 2 * ((j >> 2) / GC) + (j & 1), p_tc = 2 * ((j >> 2) % GC) + ((j >> 1) & 1);
  const int patch_lo = pl * PZ + (2 * p_tr) * RX + 2 * p_tc;
  const int patch2_lo = lane < 32 ? patch_lo + 4 * PZ : patch_lo;          // lanes 32..63: any valid address
  const int v_row = (p * 16 + j) * 16;
  const int swz = (j >> 2) & 3;
  const int dma_lo = (lane >> 2) * 16 + (((lane & 3) ^ ((lane >> 4) & 3)) * 4);
  float* const rawp = raw_s + p * RAWC;            // wave-private raw brick [z][y][RX], plane stride PZ

  struct Item {
    int k, c0;           // tile ordinal, first channel of the chunk
    TileXY t;
  };
  auto next_item = [&](Item& it) __attribute__((always_inline)) {      // -> true when the tile changed
    it.c0 += KC;
    if (it.c0 < a.Cin) return false;
    it.c0 = 0;
    ++it.k;
    if (it.k < nt_mine) it.t = tile_of(it.k);
    return true;
  };

  // ---- raw brick: NSP dword positions per lane (buffer loads: zero padding and the channel tail from the range check,
  // offset 2^31 / zero records), the filter value multiplied in, dword LDS stores into the wave-private padded brick.
  // Two register sets: a brick is fetched TWO slots (~3.5 us) before it is committed -- one slot is less than a loaded
  // HBM round trip, and then this wave reaches the barrier late and all eight wait.
  int lro[NSP];
  unsigned sob[NSP];
  float scl_plan[HAS_SCALE ? NSP : 1];
  float vin[2][NSP], scl[2][HAS_SCALE ? NSP : 1];
#pragma unroll
  for (int i = 0; i < NSP; ++i) {
    const int r = lane + 64 * i;
    const int zz = r / (IY * IX), r2 = r - zz * (IY * IX);
    const int yy = r2 / IX, xx = r2 - yy * IX;
    lro[i] = r < PRAW ? zz * PZ + yy * RX + xx : G::DUMP;
  }
  auto plan_tile = [&](const TileXY& t) __attribute__((always_inline)) {
    const float* scb = (HAS_SCALE && a.in_scale) ? a.in_scale + (size_t)t.b * vol : nullptr;
#pragma unroll
    for (int i = 0; i < NSP; ++i) {
      const int r = lane + 64 * i;
      const int zz = r / (IY * IX), r2 = r - zz * (IY * IX);
      const int yy = r2 / IX, xx = r2 - yy * IX;
      const int z = t.z0 - 1 + zz, y = t.y0 - 1 + yy, x = t.x0 - 1 + xx;
      const bool ok = r < PRAW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H &&
                      (unsigned)x < (unsigned)a.W;
      const unsigned sp = ok ? (unsigned)((z * a.H + y) * a.W + x) : 0u;
      sob[i] = ok ? sp * 4u : 0x80000000u;
      if (HAS_SCALE) scl_plan[i] = (ok && scb) ? scb[sp] : 1.f;
    }
  };
  // The loads of this pipeline are issued as inline asm and waited for by hand: a register set is consumed two slots
  // (two loop iterations) after its loads were issued, with the loads of the later items still in flight behind them,
  // and the compiler's wait-count insertion answers a load carried around a loop back-edge with s_waitcnt vmcnt(0)
  // -- which cuts the prefetch distance back to one slot.  vmcnt counts in issue order, so "all but the N youngest"
  // with N = the loads issued after the set's own is exact (VM_AFTER_*).
  auto fetch = [&](const Item& it, auto set_tag) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
    const int c = it.c0 + p;
    const uint64_t base = reinterpret_cast<uint64_t>(a.in + ((size_t)it.t.b * a.Cin + (c < a.Cin ? c : 0)) * vol);
    i32x4 desc;                                     // raw buffer, stride 0; zero records for the channel tail
    desc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    desc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(base >> 32) & 0xffffu));
    desc[2] = c < a.Cin ? vol_bytes : 0;
    desc[3] = 0x00020000;
#pragma unroll
    for (int i = 0; i < NSP; ++i) {
      float v;
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(sob[i]), "s"(desc) : "memory");
      vin[SET][i] = v;
      if (HAS_SCALE) scl[SET][i] = scl_plan[i];
    }
  };
  auto scale_vin = [&](auto set_tag) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
    if (HAS_SCALE) {
#pragma unroll
      for (int i = 0; i < NSP; ++i) vin[SET][i] *= scl[SET][i];
    }
  };
  auto commit_raw = [&](auto set_tag) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
#pragma unroll
    for (int i = 0; i < NSP; ++i) rawp[lro[i]] = vin[SET][i];
  };
  // weights of a chunk: the packed image is the LDS image; each producer wave moves six 1-KB pieces (16 cout rows x 4
  // position quads) through registers -- six independent 16-byte buffer loads (descriptor and piece offset on the
  // scalar unit, one constant lane offset), six 16-byte LDS stores two slots later.  Slot s of a row holds quad
  // s ^ (row >> 2): conflict-free B-fragment reads without padding.
  f32x4 uq[2][6];
  i32x4 u_desc;
  {
    const uint64_t wb = reinterpret_cast<uint64_t>(a.wpk);
    u_desc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)wb);
    u_desc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(wb >> 32) & 0xffffu));
    u_desc[2] = (int)((size_t)nchunk * a.nco * U_CHUNK * 4);
    u_desc[3] = 0x00020000;
  }
  const int u_voff = dma_lo * 4;
  auto load_u = [&](const Item& it, auto set_tag) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
    const int so = ((it.c0 >> 2) * a.nco + it.t.tc) * (U_CHUNK * 4) + p * 1024;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      f32x4 v;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(u_voff), "s"(u_desc), "s"(so + q * 4096) : "memory");
      uq[SET][q] = v;
    }
  };
  // younger loads at the point of use, steady state: the set's own loads were issued two slots ago as [LU(i) F(i+1)],
  // then came [LU(i+1) F(i+2)]
  constexpr int VM_AFTER_LU = NSP + 6 + NSP;       // SU(i) waits for LU(i)
  constexpr int VM_AFTER_F = 6 + NSP;              // C(i + 1) waits for F(i + 1)

  // ---- the producer pipeline.  Slot m = the interval between two block barriers in which the consumers multiply item
  // m.  Beside their MFMA stream a vector-ALU instruction of this wave has to wait for the matrix pipe to drain (~60
  // cycles for an isolated one, ~5 inside a dense burst; tools/probes/mfma_f32_neighbours.hip).  So ALL vector
  // arithmetic of a slot -- the 32 packed adds of V = Bt d B for this lane's two patches, the filter multiplies, a new
  // tile's staging plan -- runs as one burst right behind the barrier, when the matrix pipe is empty anyway (the
  // consumers wait for their first LDS reads), and everything behind it is data movement:
  //   slot m:  T(m+1) [scale(m+2)] [plan(m+4)]  |  W(m+1) SU(m+1)  C(m+2) R(m+2)  LU(m+3) F(m+4)  | barrier
  //   F = raw loads -> vin[set]   C = vin -> this wave's brick in LDS   R = patch rows -> d   T = d -> o (vector burst)
  //   W = o -> V[buf]             LU = weight loads -> uq[set]          SU = uq -> U[buf]
  f32x2 d[2][4][2], o[2][4][2];
  auto read_patches = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* src = rawp + (q ? patch2_lo : patch_lo);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[q][r][0] = *reinterpret_cast<const f32x2*>(src + r * RX);
        d[q][r][1] = *reinterpret_cast<const f32x2*>(src + r * RX + 2);
      }
    }
  };
  auto transform = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x2 t0, t1, t2, t3;
      // rows of Bt d: r0 = d0 - d2, r1 = d1 + d2, r2 = d2 - d1, r3 = d1 - d3 (two column pairs each), then per row the
      // column combinations (t0-t2, t1+t2) and (t2-t1, t1-t3) as one v_pk_add_f32 each (op_sel picks the halves)
      asm volatile(
          "v_pk_add_f32 %8, %12, %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %9, %13, %17 neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %10, %14, %16\n\t"
          "v_pk_add_f32 %11, %15, %17\n\t"
          "v_pk_add_f32 %0, %8, %9 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
          "v_pk_add_f32 %1, %9, %8 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\t"
          "v_pk_add_f32 %2, %10, %11 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
          "v_pk_add_f32 %3, %11, %10 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\t"
          "v_pk_add_f32 %8, %16, %14 neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %9, %17, %15 neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %10, %14, %18 neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %11, %15, %19 neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %4, %8, %9 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
          "v_pk_add_f32 %5, %9, %8 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\t"
          "v_pk_add_f32 %6, %10, %11 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
          "v_pk_add_f32 %7, %11, %10 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]"
          : "=&v"(o[q][0][0]), "=&v"(o[q][0][1]), "=&v"(o[q][1][0]), "=&v"(o[q][1][1]), "=&v"(o[q][2][0]), "=&v"(o[q][2][1]),
            "=&v"(o[q][3][0]), "=&v"(o[q][3][1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
          : "v"(d[q][0][0]), "v"(d[q][0][1]), "v"(d[q][1][0]), "v"(d[q][1][1]), "v"(d[q][2][0]), "v"(d[q][2][1]),
            "v"(d[q][3][0]), "v"(d[q][3][1]));
    }
  };
  float* const v_dst = v_s + pl * V_PLANE + v_row;
  auto write_v = [&](auto buf_tag) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_tag)::value;
    float* dst = v_dst + BUF * V_BUF;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<f32x4*>(dst + ((r ^ swz) * 4)) = (f32x4){o[0][r][0][0], o[0][r][0][1], o[0][r][1][0], o[0][r][1][1]};
    if (lane < 32) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x4*>(dst + 4 * V_PLANE + ((r ^ swz) * 4)) =
            (f32x4){o[1][r][0][0], o[1][r][0][1], o[1][r][1][0], o[1][r][1][1]};
    }
  };
  auto store_ub = [&](auto buf_tag) __attribute__((always_inline)) {       // weights of an item i: set i & 1 -> U[i & 1]
    constexpr int BUF = decltype(buf_tag)::value;
    float* ub = u_s + BUF * U_CHUNK + lane * 4;
#pragma unroll
    for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(ub + (p + 4 * q) * 256) = uq[BUF][q];
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // prologue: item 0 into V[0] / U[0]; d = patches of item 1; weights of items 1, 2 and raw bricks of items 2, 3 on
  // their way (item i: register set i & 1)
  Item fi;                                          // the item whose raw brick was fetched last
  fi.k = 0; fi.c0 = 0; fi.t = tile_of(0);
  plan_tile(fi.t);
  fetch(fi, S0{});
  load_u(fi, S0{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  scale_vin(S0{});
  commit_raw(S0{});
  read_patches();
  transform();
  write_v(S0{});
  store_ub(S0{});
  if (1 < M) {
    if (next_item(fi)) plan_tile(fi.t);
    fetch(fi, S1{});
    load_u(fi, S1{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    scale_vin(S1{});
    commit_raw(S1{});
    read_patches();
    if (2 < M) {
      if (next_item(fi)) plan_tile(fi.t);
      fetch(fi, S0{});
      load_u(fi, S0{});
      if (3 < M) {
        if (next_item(fi)) plan_tile(fi.t);
        fetch(fi, S1{});
      }
    }
  }
  __syncthreads();                                 // item 0 ready

  auto slot = [&](int m, auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;  // m & 1: this slot fills buffers (m + 1) & 1 = 1 - PAR
    using SP = std::integral_constant<int, PAR>;
    using NB = std::integral_constant<int, 1 - PAR>;
#ifndef DV_PS_ABLATE_PRODUCER
    if (wave == 4) PS_STAMP(1, m, 0);
    // ---- vector burst
    Item nx = fi;
    bool new_tile = false;
#ifndef DV_PS_ABLATE_T
    if (m + 1 < M) transform();                     // d (item m + 1) -> o
    if (m + 2 < M) scale_vin(SP{});                 // registers of item m + 2
#endif
    if (m + 4 < M) {
      new_tile = next_item(nx);
      if (new_tile) plan_tile(nx.t);
    }
    if (wave == 4) PS_STAMP(1, m, 1);
    // ---- data movement only
    const bool steady = m + 4 < M;                  // every load of the pattern above was really issued
#ifndef DV_PS_ABLATE_W
    if (m + 1 < M) {
      write_v(NB{});
      if (steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_AFTER_LU) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      store_ub(NB{});                               // weights of item m + 1, loaded two slots ago
    }
#endif
    if (wave == 4) PS_STAMP(1, m, 2);
    if (m + 2 < M) {
#ifndef DV_PS_ABLATE_C
      if (steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_AFTER_F) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      commit_raw(SP{});                             // brick of item m + 2, fetched two slots ago
#endif
#ifndef DV_PS_ABLATE_R
      read_patches();
#endif
    }
    if (m + 3 < M) load_u(fi, NB{});                // weights of item m + 3 (fi = item m + 3 on entry)
    if (m + 4 < M) {
      fi = nx;
#ifndef DV_PS_ABLATE_F
      fetch(fi, SP{});                              // brick of item m + 4
#endif
    }
    if (wave == 4) PS_STAMP(1, m, 3);
#endif
    __syncthreads();
    if (wave == 4) PS_STAMP(1, m, 6);
  };
#pragma unroll 1
  for (int m = 0; m < M; m += 2) {
    slot(m, S0{});
    if (m + 1 < M) slot(m + 1, S1{});
  }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

int persistent_grid() {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  return ncu;
}

}  // namespace

#ifdef DV_PS_STAMPS
extern "C" int dv_ps_read_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ps_stamps), sizeof(g_ps_stamps));
}
#endif

// Same contract as dv_conv3d_wino_f32 (include/diffuvolume_hip.h); `wpacked` from dv_conv3d_wino_pack_weights_f32.
extern "C" int dv_conv3d_wino_ps_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                                     const float* in_scale, const float* residual, float* out, int B, int Cin, int D,
                                     int H, int W, int Cout, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE((size_t)D * H * W * sizeof(float) <= 0x7fffffffull, DV_ERR_SHAPE);   // 31-bit byte offsets in a channel
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_LEAKY, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(dv_aligned16(wpacked), DV_ERR_ALIGN);
  WinoPsArgs a;
  a.in = in; a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.in_scale = in_scale;
  a.residual = residual; a.out = out;
  a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
  a.fast_ok = (W % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual));
  hipStream_t s = (hipStream_t)stream;
  auto launch = [&](auto shape) {
    constexpr int SHAPE = decltype(shape)::value;
    using G = PG<SHAPE>;
    a.ntx = cdiv(W, G::TW); a.nty = cdiv(H, G::TH); a.ntz = cdiv(D, G::TD); a.nco = cdiv(Cout, 32);
    const long long tiles = (long long)B * a.nco * a.ntz * a.nty * a.ntx;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return (int)DV_ERR_SHAPE;
    a.ntiles = (int)tiles;
    const int ncu = persistent_grid();
    const unsigned blocks = (unsigned)(tiles < ncu ? tiles : ncu);
    if (in_scale)
      hipLaunchKernelGGL((conv3d_wino_ps_kernel<true, SHAPE>), dim3(blocks), dim3(512), 0, s, a);
    else
      hipLaunchKernelGGL((conv3d_wino_ps_kernel<false, SHAPE>), dim3(blocks), dim3(512), 0, s, a);
    return dv_launch_status();
  };
  auto padded = [&](int tw, int th) { return (long long)cdiv(W, tw) * tw * cdiv(H, th) * th; };
  const long long p0 = padded(16, 4), p1 = padded(8, 8), p2 = padded(4, 16);
  int shape = 0;
  if (p1 < p0 && p1 <= p2) shape = 1;
  else if (p2 < p0 && p2 < p1) shape = 2;
  if (shape == 1) return launch(std::integral_constant<int, 1>{});
  if (shape == 2) return launch(std::integral_constant<int, 2>{});
  return launch(std::integral_constant<int, 0>{});
}
