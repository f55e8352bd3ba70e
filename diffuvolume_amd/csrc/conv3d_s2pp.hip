// K4p: the 3x3x3 STRIDE-2 aggregation convolution (convbn_3d with stride 2: hourglass conv1 / conv3,
// SceneFlow/models/acv_ddim.py:60,:66; KITTI12/models/pwcnet_ddim.py:137-147) in its polyphase minimal-filtering form.
//
// Per in-plane axis a stride-2 3-tap filter reads, for the output pair (2t, 2t+1), the inputs r0..r4 = in[4t-1 .. 4t+3]:
//     out[2t]   = w0 r0 + w1 r1 + w2 r2            out[2t+1] = w0 r2 + w1 r3 + w2 r4
// The odd input phase (r0, r2, r4) sees a 2-tap filter (w0, w2), the even phase (r1, r3) one tap.  F(2,2) on the odd
// phase -- m1 = (r0-r2) w0, m2 = r2 (w0+w2), m3 = (r4-r2) w2 -- needs 3 multiplies for its 4 products, and products that
// end in the same output sum share an accumulator:
//     A0 = (r0-r2) w0 + r1 w1      A1 = r2 (w0+w2)      A2 = r3 w1 + (r4-r2) w2        out[2t] = A0 + A1,  out[2t+1] = A1 + A2
// 5 multiplies per 2 outputs instead of 6.  In-plane (both axes): a 2x2 output tile reads a 5x5 input patch, transformed
// by 20 subtractions into 25 A operands V[i][j]; they meet 16 distinct weight combinations U[by][bx] (by, bx over
// {w0, w1, w0+w2, w2}, summed once by the host pack) in 25 MFMA k-steps that accumulate into 3x3 sums: 25 multiplies per
// 2x2 outputs instead of 36, i.e. 18.75 instead of 27 per output with the depth taps kept direct (1.44x fewer MFMAs than
// conv3d.hip's implicit GEMM), still on the exact-fp32 instruction v_mfma_f32_16x16x4_f32.  The transform entries are
// 0 / +-1 only: the rounding is that of one extra fp32 subtraction per operand (tools/probes/polyphase_f22_numerics.py:
// 0.6x the direct sum's error on a 32 -> 64 plane).
//
// Tile = 2 output planes x 16 patches (2x2 outputs each) x 64 output channels = four MFMA waves; wave = (plane, half of
// the channels): M = the 16 patches, N = 16 output channels (two N-tiles per wave), K = 4 input channels per step, 72
// accumulator registers.  Per chunk of 4 input channels the haloed raw brick -- 5 planes x (4 TR + 1) rows x (4 TC + 4)
// floats, rows cut at multiples of 4 so that they travel as 16-byte quads -- lies in LDS; every wave reads the three 5x5
// patches of its plane (kd = 0, 1, 2: one ds_read_b128 + one ds_read_b32 per row, just in time: the MFMA groups of a patch
// use its operand rows one after the other) and transforms them in registers.  The weights never touch LDS: a B fragment
// is one coalesced 16-byte buffer load per lane from the packed image (L1 / L2 resident), issued three MFMA groups ahead
// into a register ring.
//
// Launch shape (how it got there: profiles/r05_s2pp_experiments.txt).  ONE persistent block per CU: eight MFMA waves (two
// tiles side by side, neighbours in x) + three LOADER waves, 168 registers each.  The loaders do nothing but copy bricks
// global -> LDS by DMA (`buffer_load_dwordx4 ... lds`; the range check writes the zero padding), double-buffered, one
// block barrier per chunk.  Why waves of their own: vector-memory operations of a wave complete in order on ONE counter,
// so with the brick fetched by the MFMA waves themselves every wait for a B fragment (L2, a few hundred cycles) also
// waited for the brick's HBM round trip issued before it -- 1.52 ms against 1.06 ms with the brick reads forced into L2,
// whatever the staging (dword or 16-byte register round trip, in-wave DMA, B ring 3 or 6 deep).  Why persistent: the
// software pipeline runs across tile boundaries, so a tile's first brick / first patch / first B fragments hide behind
// the previous tile's MFMAs (per-tile blocks lost 0.4 of 1.8 ms to their prologue and epilogue).  Tile order x-fastest:
// the 80-byte row pieces of a brick use 1-2 full 128-byte lines each, and the halo quad's line is the x-neighbour's main
// line -- walking x first makes it an L2 hit (1.55 -> 1.33 ms); for the same reason the wide patch shapes win even
// where they pad.  Measured (batch 8): 32 -> 64 at 48 x 128 x 240 1.21 ms (direct implicit GEMM 1.56), 64 -> 128 at
// 24 x 64 x 120 0.52 ms (0.67).

#include <type_traits>

#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// timing-only ablations (results wrong): 1 = no B loads in the loop, 2 = no brick copies after the first two, 4 = no patch
// reads / transforms, 8 = no step barriers (and no copies), 32 = every brick read inside one 256-KB window (L2 hits)
#ifndef DV_PP_ABL
#define DV_PP_ABL 0
#endif
#ifndef DV_PP_ORDER
#define DV_PP_ORDER 1
#endif
#ifndef DV_PP_LOADERS
#define DV_PP_LOADERS 3
#endif
#ifndef DV_PP_PD
#define DV_PP_PD 3
#endif

// SHAPE = how the 16 patches of a wave lie in its output plane: 0 -> 2 rows x 8 columns (4 x 16 outputs),
// 1 -> 4 x 4 (8 x 8 outputs); the host's choice is at the bottom of the file.
template <int SHAPE_>
struct PG {
  static constexpr int SHAPE = SHAPE_;
  static_assert(SHAPE == 0 || SHAPE == 1, "patch shapes");
  static constexpr int TR = SHAPE == 0 ? 2 : 4, TC = 16 / TR;
  static constexpr int KC = 4, NT = 2, TD = 2;
  static constexpr int IZ = 2 * TD + 1, IY = 4 * TR + 1;
  // quads of a brick row: input x in [4 TC bx - 4, 4 TC bx + 4 TC) (+ one surplus quad for 2 x 8, see RX)
  static constexpr int LQ = SHAPE == 0 ? TC + 2 : TC + 1;
  // row stride in floats (a multiple of 4).  Bank plan of the ds_read_b128 patch reads (16 lanes = the 16 patches of one
  // channel per pass): patch columns are one quad apart, patch rows RX quads; 4 x 4 patches tile the 16 quad slots with
  // RX = 20, 2 x 8 with RX = 40 (one surplus quad per row).
  static constexpr int RX = SHAPE == 0 ? 40 : 20;
  static constexpr int ROWS = IZ * IY;
  static constexpr int CS = ROWS * RX;                 // channel stride (floats)
  static constexpr int QPC = ROWS * LQ;                // loaded quads per channel
  static constexpr int NQ = KC * QPC;                  // per chunk
  static constexpr int NP = (NQ + 63) / 64;            // LDS-DMA pieces (64 lanes x 16 bytes) per chunk
  static constexpr int RAW_FLOATS = ((NP + 3) / 4 * 4) * 256;   // the lanes / pieces past the brick write zeros behind it
  static_assert(RX == 4 * LQ && CS == 4 * QPC, "the loaded quads of a chunk are contiguous in LDS: quad `it` at float 4 * it");
  static_assert(4 * RAW_FLOATS * 4 <= 160 * 1024, "one block per CU: two sub-tiles x two bricks");
};

namespace pp {
constexpr int U_CHUNK = 3 * 4 * 4 * 4 * 16 * 4;        // [kd 3][nt 4][by 4][k 4][n 16][bx 4] floats per (chunk, 64 couts)
__host__ __device__ constexpr int amap(int i) { return i < 2 ? 0 : (i == 2 ? 1 : 2); }    // accumulator row / column of operand row / column i: 0 0 1 2 2
__host__ __device__ constexpr int bmap(int i) { return i == 3 ? 1 : (i == 4 ? 3 : i); }    // its weight combination (w0, w1, w0+w2, w2): 0 1 2 1 3
}  // namespace pp

struct PPArgs {
  const float* in;
  const float* wpk;      // [Cin/4][Coutp/64] chunks of pp::U_CHUNK
  const float* ch_scale;
  const float* ch_bias;
  const float* residual;
  float* out;
  int B, Cin, D, H, W, Cout, Do, Ho, Wo;
  int ntx, nty, ntz, nco;
  int ntiles;            // B * ntx * nty * ntz * nco
  int act;
  int vec_ok;            // Wo % 4 == 0, 16-byte aligned out / residual: rows go out as 16-byte stores
  unsigned wpk_bytes;
};

struct Tile { int tcb, z0, y0, x0, b; bool valid; };

template <int SHAPE>
__global__ __launch_bounds__(512 + 64 * DV_PP_LOADERS, 1) void conv3d_s2pp_kernel(PPArgs a) {
  using G = PG<SHAPE>;
  constexpr int KC = G::KC, NT = G::NT, TR = G::TR, TC = G::TC, IY = G::IY, RX = G::RX, CS = G::CS, LQ = G::LQ;
  constexpr int NQ = G::NQ, QPC = G::QPC, NP = G::NP, RAW_FLOATS = G::RAW_FLOATS;
  // two bricks (separate arrays: the compiler has to see that an LDS-DMA into one cannot alias the patch reads of the other)
  // (x 2 sub-tiles: a block of eight MFMA waves works on two tiles side by side)
  __shared__ __attribute__((aligned(16))) float raw_a0[RAW_FLOATS];
  __shared__ __attribute__((aligned(16))) float raw_b0[RAW_FLOATS];
  __shared__ __attribute__((aligned(16))) float raw_a1[RAW_FLOATS];
  __shared__ __attribute__((aligned(16))) float raw_b1[RAW_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7: MFMA waves (sub-tile wave >> 2), 8: the loader

  // ---- this block's tile list: every XCD owns a contiguous slab of the tile order (output-channel blocks of a brick,
  // then z, y, x, batch: neighbours share halos in that XCD's L2), the blocks of an XCD walk it round-robin ----
  const unsigned nblk = gridDim.x, xcd = blockIdx.x & 7u, bidx = blockIdx.x >> 3;
  const unsigned nbx = (nblk >> 3) + (xcd < (nblk & 7u) ? 1u : 0u);                 // blocks on this XCD
  const unsigned npairs = ((unsigned)a.ntiles + 1u) >> 1;                          // the block works on tiles 2P, 2P + 1 at once
  const unsigned tq = npairs >> 3, trm = npairs & 7u;
  const unsigned slab_lo = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq;
  const unsigned slab_n = tq + (xcd < trm ? 1u : 0u);
  const int my_tiles = __builtin_amdgcn_readfirstlane(bidx < slab_n ? (int)((slab_n - bidx + nbx - 1) / nbx) : 0);  // pairs slab_lo + bidx + k * nbx
  if (my_tiles == 0) return;

  auto tile_at = [&](int k, int h) __attribute__((always_inline)) {               // sub-tile h of this block's k-th pair
    unsigned t = 2u * (slab_lo + bidx + (unsigned)k * nbx) + (unsigned)h;
    Tile r;
    r.valid = t < (unsigned)a.ntiles;
    if (!r.valid) t = (unsigned)a.ntiles - 1u;
    r.tcb = t % a.nco; t /= a.nco;
#if DV_PP_ORDER == 0
    r.z0 = (t % a.ntz) * G::TD; t /= a.ntz;
    r.y0 = (t % a.nty) * 2 * TR; t /= a.nty;
    r.x0 = (t % a.ntx) * 2 * TC;
    r.b = t / a.ntx;
#elif DV_PP_ORDER == 1
    r.x0 = (t % a.ntx) * 2 * TC; t /= a.ntx;
    r.y0 = (t % a.nty) * 2 * TR; t /= a.nty;
    r.z0 = (t % a.ntz) * G::TD;
    r.b = t / a.ntz;
#else
    r.x0 = (t % a.ntx) * 2 * TC; t /= a.ntx;
    r.z0 = (t % a.ntz) * G::TD; t /= a.ntz;
    r.y0 = (t % a.nty) * 2 * TR;
    r.b = t / a.nty;
#endif
    return r;
  };

  const size_t plane = (size_t)a.H * a.W;
  const size_t vol = (size_t)a.D * plane;
  const unsigned vol_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(vol * sizeof(float)));   // <= 2^30 (host)
  const int n_chunk = (a.Cin + KC - 1) / KC;
  const int NCH = (n_chunk + 1) & ~1;                  // chunks run in pairs; a surplus chunk reads zero records
  // an even number of output-channel blocks: the two tiles of a pair are the two channel blocks of ONE brick -- it is copied
  // once and both halves of the block read it (64 -> 128: half the copies)
  const bool share = (a.nco & 1) == 0;
  const int n_steps = my_tiles * NCH;                  // step s = chunk s % NCH of tile s / NCH; brick in buffer s & 1

  // Barrier protocol (all five waves, the same count on both paths): P before step 0, then ONE per step, B_s, which the
  // MFMA waves reach at their group 10.  After B_s nobody reads step s's brick any more (the last patch of it was read
  // in groups 6 / 7), so the loader overwrites that buffer with step s + 2's brick and waits for its own copies before it
  // goes to B_{s+1}; behind B_{s+1} the MFMA waves read the first patch of step s + 2.
  if (wave >= 8) {
    // =========================== loader wave: global -> LDS by DMA, nothing else ===========================
    // Its vector-memory counter is its own: the MFMA waves' waits for their B fragments (L2 latency, two groups ahead)
    // never stand behind an HBM round trip of the brick -- issued from the MFMA waves themselves, in order on one
    // counter, the brick's latency was exposed in every step (1.52 ms; with the brick reads forced into L2: 1.06 ms).
    // `buffer_load_dwordx4 ... lds`: lane l of piece p writes its 16 bytes at piece base + 16 l; a lane that fails the
    // range check writes zeros (the zero padding, the channel tail, the surplus chunk).
    constexpr int NL = DV_PP_LOADERS, NPL = (NP + NL - 1) / NL;   // loader li copies pieces li, li + NL, ...
    const int li = wave - 8;
    static_assert(NL * NPL * 256 <= RAW_FLOATS, "every piece a loader writes lies inside the buffer");
    unsigned sob0[NPL], sob1[NPL];                       // byte offsets of this lane's quad in every piece, per sub-tile
    auto plan = [&](const Tile& t, unsigned (&sob)[NPL]) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const int it = 64 * (li + NL * p) + lane;
        const int cl = it / QPC, r1 = it - cl * QPC;
        const int row = r1 / LQ, q = r1 - row * LQ;
        const int zz = row / IY, yy = row - zz * IY;
        const int z = 2 * t.z0 - 1 + zz, y = 2 * t.y0 - 1 + yy, x = 2 * t.x0 - 4 + 4 * q;
        const bool ok = t.valid && it < NQ && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H &&
                        (unsigned)x < (unsigned)a.W;                                // W % 4 == 0: a quad is inside or outside
        sob[p] = ok ? (unsigned)cl * vol_bytes + (unsigned)((z * a.H + y) * a.W + x) * 4u : 0xfffffff0u;
        if ((DV_PP_ABL & 32) && ok) sob[p] &= 0x3fff0u;     // timing only: every fetch inside one 256-KB window (L2 hits)
      }
    };
    auto dma_raw = [&](const Tile& t, const unsigned (&sob)[NPL], int c, float* dst) __attribute__((always_inline)) {
      const int c0 = c * KC;
      const int left = a.Cin - c0;
      const unsigned rec = left <= 0 ? 0u : (unsigned)(left < KC ? left : KC) * vol_bytes;
      const uint64_t ba = reinterpret_cast<uint64_t>(a.in + ((size_t)t.b * a.Cin + (left > 0 ? c0 : 0)) * vol);
      const uint64_t bu = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba) |
                          ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ba >> 32)) << 32);
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(bu), 0,
                                                        __builtin_amdgcn_readfirstlane((int)rec), 0x00020000);
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (li + NL * p) * 256), 16,
                                                 (int)sob[p], 0, 0, 0);
    };
    Tile ft0 = tile_at(0, 0), ft1 = tile_at(0, 1);
    int fk = 0, fc = 0;                                // fetch cursor (pair, chunk)
    auto advance = [&]() __attribute__((always_inline)) {
      if (++fc == NCH) {
        fc = 0;
        if (++fk < my_tiles) { ft0 = tile_at(fk, 0); ft1 = tile_at(fk, 1); plan(ft0, sob0); plan(ft1, sob1); }
      }
    };
    auto fetch = [&](float* d0, float* d1) __attribute__((always_inline)) {
      dma_raw(ft0, sob0, fc, d0);
      if (!share) dma_raw(ft1, sob1, fc, d1);
      advance();
    };
    plan(ft0, sob0);
    plan(ft1, sob1);
    fetch(raw_a0, raw_a1);
    fetch(raw_b0, raw_b1);                             // (NCH >= 2: step 1 is chunk 1 of the same pair)
    // (explicit waits: the compiler puts its own s_waitcnt vmcnt(0) in front of a barrier that follows an LDS-DMA in
    // straight-line code, but not in front of the loop-header barrier that follows one across the back edge)
    auto arrive = [&]() __attribute__((always_inline)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    };
    arrive();                                          // P
    if (DV_PP_ABL & 8) return;                         // (timing only: no step barriers, no further copies)
#pragma unroll 1
    for (int s = 0; s < n_steps; s += 2) {
      arrive();                                        // B_s
      if (s + 2 < n_steps && !(DV_PP_ABL & 2)) fetch(raw_a0, raw_a1);
      arrive();                                        // B_{s+1}
      if (s + 3 < n_steps && !(DV_PP_ABL & 2)) fetch(raw_b0, raw_b1);
    }
    return;
  }

  // =========================== MFMA waves ===========================
  const int sub = wave >> 2, pl = (wave >> 1) & 1, nh = wave & 1;
  const int j = lane & 15, kq = lane >> 4;
  const float* const raw_a = (sub && !share) ? raw_a1 : raw_a0;
  const float* const raw_b = (sub && !share) ? raw_b1 : raw_b0;

  // ---- weights: one descriptor over the packed image; lane part of the address = lane * 16 bytes, the (chunk, kd, nt, by)
  // part is scalar ----
  const uint64_t wb = reinterpret_cast<uint64_t>(a.wpk);
  const uint64_t wbs = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wb) |
                       ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wb >> 32)) << 32);
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(wbs), 0, (int)a.wpk_bytes, 0x00020000);
  const int wvoff = lane * 16;
  constexpr int PD = DV_PP_PD, RD = PD == 5 ? 6 : PD + 1;            // B fragments are fetched PD groups ahead into a ring of RD slots
  static_assert(12 % RD == 0, "ring slots repeat with the step");
  f32x4 bq[RD][NT];
  auto load_b = [&](int tcb, int chunk, int g, int slot) __attribute__((always_inline)) {
    const int ch = chunk < n_chunk ? chunk : n_chunk - 1;           // the surplus chunk: any finite weights (its A is zero)
    const int kd = g >> 2, by = g & 3;
    const int base = (ch * a.nco + tcb) * (pp::U_CHUNK * 4);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int so = base + (((kd * 4 + 2 * nh + n) * 4 + by) * 256) * 4;
      bq[slot][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, so, 0));
    }
  };

  // this lane's patch (A operand row j): origin inside a channel of the brick.  Patch columns 4 tc - 1 .. 4 tc + 3 of
  // the tile = brick floats 4 tc + 3 (the halo column, last float of the quad before) and the quad at 4 tc + 4
  const int a_tr = j / TC, a_tc = j % TC;
  const int patch_lo = kq * CS + ((2 * pl) * IY + 4 * a_tr) * RX + 4 * a_tc;

  // V[slot]: the 25 A operands of one patch.  Patches are consumed in the order (step, kd); patch p lives in slot p & 1,
  // so the slot of (step, kd) flips with the step parity: the step body is instantiated for both parities.  The MFMA
  // groups of a patch go through its operand rows in the order 0 | 1, 3 | 2 | 4, so the rows die one after the other and
  // the next patch is read just in time (rows 0, 2, 4 in group by = 2, rows 1, 3 in by = 3): about 30 live operand
  // registers instead of 50.
  float V[2][5][5];
  auto read_row = [&](const float* rb, int kd, int slot, int r) __attribute__((always_inline)) {
    const float* p = rb + patch_lo + kd * IY * RX;
    V[slot][r][0] = p[r * RX + 3];
    const f32x4 qd = *reinterpret_cast<const f32x4*>(p + r * RX + 4);
    V[slot][r][1] = qd[0]; V[slot][r][2] = qd[1]; V[slot][r][3] = qd[2]; V[slot][r][4] = qd[3];
  };
  auto col_tf = [&](int slot, int r) __attribute__((always_inline)) {      // operand 0 = c0 - c2, operand 4 = c4 - c2
    V[slot][r][0] -= V[slot][r][2];
    V[slot][r][4] -= V[slot][r][2];
  };
  auto row_tf = [&](int slot) __attribute__((always_inline)) {             // operand row 0 = r0 - r2, row 4 = r4 - r2
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      V[slot][0][c] -= V[slot][2][c];
      V[slot][4][c] -= V[slot][2][c];
    }
  };

  f32x4 acc[3][3][NT];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[i][jj][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // ---- epilogue of a finished tile: out(2t+a, 2t'+b) = sum of the 2x2 accumulators at (a, b); BN scale / bias, residual,
  // activation.  Accumulator element e of lane (kq, j) = patch 4 kq + e, output channel j: RL = min(TC, 4) patches of a lane
  // are neighbours in x, i.e. 2 RL consecutive outputs per output row ----
  const size_t oplane = (size_t)a.Ho * a.Wo;
  const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
    constexpr int RL = TC < 4 ? TC : 4, NRUN = 4 / RL;
    const int zo = t.z0 + pl;
    if (zo >= a.Do || !t.valid) return;
    const bool fast = a.vec_ok && t.x0 + 2 * TC <= a.Wo && t.y0 + 2 * TR <= a.Ho;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = t.tcb * 64 + nh * 32 + n * 16 + j;
      if (co >= a.Cout) continue;
      const float sc = a.ch_scale ? a.ch_scale[co] : 1.f;
      const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
      const size_t cbase = (((size_t)t.b * a.Cout + co) * a.Do + zo) * oplane;
#pragma unroll
      for (int run = 0; run < NRUN; ++run) {
        const int m0 = 4 * kq + run * RL;                           // first patch of the run
        const int yo = t.y0 + 2 * (m0 / TC), xo = t.x0 + 2 * (m0 % TC);
#pragma unroll
        for (int ya = 0; ya < 2; ++ya) {
          float v[2 * RL];
#pragma unroll
          for (int e = 0; e < RL; ++e)
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
              const int ee = run * RL + e;
              v[2 * e + xb] = (acc[ya][xb][n][ee] + acc[ya][xb + 1][n][ee]) + (acc[ya + 1][xb][n][ee] + acc[ya + 1][xb + 1][n][ee]);
            }
          const size_t o = cbase + (size_t)(yo + ya) * a.Wo + xo;
          if (fast) {
#pragma unroll
            for (int h = 0; h < RL / 2; ++h) {
              f32x4 u = (f32x4){v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]} * sc + bi;
              if (a.residual) u += *reinterpret_cast<const f32x4*>(a.residual + o + 4 * h);
              if (a.act == DV_ACT_MISH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = dv_act(u[e], DV_ACT_MISH);
              } else {
                u = __builtin_elementwise_max(u, u * slope);        // ReLU / LeakyReLU / identity: slope 0 / 0.01 / 1
              }
              *reinterpret_cast<f32x4*>(a.out + o + 4 * h) = u;
            }
          } else if (yo + ya < a.Ho) {
#pragma unroll
            for (int e = 0; e < 2 * RL; ++e)
              if (xo + e < a.Wo) {
                float u = fmaf(v[e], sc, bi);
                if (a.residual) u += a.residual[o + e];
                a.out[o + e] = dv_act(u, a.act);
              }
          }
        }
      }
    }
  };

  // ---- prologue: the first B fragments; behind P the first patch of step 0 ----
  Tile cur = tile_at(0, sub);
  Tile nxt = my_tiles > 1 ? tile_at(1, sub) : cur;
#pragma unroll
  for (int g = 0; g < PD; ++g) load_b(cur.tcb, 0, g, g);
  zero_acc();
  __syncthreads();                                      // P
#pragma unroll
  for (int r = 0; r < 5; ++r) read_row(raw_a, 0, 0, r);
  row_tf(0);
  col_tf(0, 0); col_tf(0, 2); col_tf(0, 4);             // (rows 1, 3: in group 0 of step 0, like every patch)

  // One step = one chunk of one tile = 12 MFMA groups (kd, by); by selects the operand rows (0 | 1, 3 | 2 | 4).  Everything
  // else is pinned to a group (sched_barrier between groups: left to itself the scheduler sinks the B loads and the patch
  // reads down to their first use and every group starts with an exposed s_waitcnt):  the B fragments of group g + PD;
  // the next patch -- (s, kd + 1), or (s + 1, 0) from the other buffer -- rows 0, 2, 4 read in group by = 2, rows 1, 3 and
  // the row transform in by = 3, the column transforms in by = 3 (rows 0, 2, 4) and the next by = 0 (rows 1, 3).
  auto step = [&](int c, auto parity) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parity)::value;                     // step parity: buffer and patch-slot phase
    const float* rb = PAR ? raw_b : raw_a;
    const float* nb = PAR ? raw_a : raw_b;
    const bool n1 = c + 1 >= NCH;                                    // step s + 1 belongs to the next tile
    const int c1 = n1 ? 0 : c + 1;
    const int tcb1 = n1 ? nxt.tcb : cur.tcb;
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      const int kd = g >> 2, by = g & 3;
      const int slot = (PAR + kd) & 1, nslot = slot ^ 1;             // patch p = 3 s + kd -> slot (s + kd) & 1
      if (g == 10 && !(DV_PP_ABL & 8)) __syncthreads();               // B_s
      if (!(DV_PP_ABL & 1)) {
        if (g + PD < 12) load_b(cur.tcb, c, g + PD, (g + PD) % RD);
        else load_b(tcb1, c1, g + PD - 12, (g + PD) % RD);
      }
      if (!(DV_PP_ABL & 4)) {
        const float* src = kd < 2 ? rb : nb;
        const int nkd = kd < 2 ? kd + 1 : 0;
        if (by == 0) { col_tf(slot, 1); col_tf(slot, 3); }
        if (by == 2) { read_row(src, nkd, nslot, 0); read_row(src, nkd, nslot, 2); read_row(src, nkd, nslot, 4); }
        if (by == 3) { read_row(src, nkd, nslot, 1); read_row(src, nkd, nslot, 3); }
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (pp::bmap(i) != by) continue;
#pragma unroll
        for (int jj = 0; jj < 5; ++jj)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[pp::amap(i)][pp::amap(jj)][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                V[slot][i][jj], bq[g % RD][n][pp::bmap(jj)], acc[pp::amap(i)][pp::amap(jj)][n], 0, 0, 0);
      }
      if (by == 3 && !(DV_PP_ABL & 4)) { row_tf(nslot); col_tf(nslot, 0); col_tf(nslot, 2); col_tf(nslot, 4); }
#ifndef DV_PP_NOPIN
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };

#pragma unroll 1
  for (int k = 0; k < my_tiles; ++k) {
#pragma unroll 1
    for (int c = 0; c < NCH; c += 2) {
      step(c, std::integral_constant<int, 0>{});
      step(c + 1, std::integral_constant<int, 1>{});
    }
    epilogue(cur);
    zero_acc();
    cur = nxt;
    if (k + 2 < my_tiles) nxt = tile_at(k + 2, sub);
  }
}

// U[by][bx] = sum of the taps combination (by, bx) stands for, formed in double and rounded once
__global__ void pack_s2pp_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout,
                                         int nchunk, int nco) {
  const size_t total = (size_t)nchunk * nco * 3 * 4 * 4 * 4 * 16;   // one thread per (chunk, cb, kd, nt, by, k, n)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int n = (int)(r % 16); r /= 16;
    const int k = (int)(r % 4); r /= 4;
    const int by = (int)(r % 4); r /= 4;
    const int nt = (int)(r % 4); r /= 4;
    const int kd = (int)(r % 3); r /= 3;
    const int cb = (int)(r % nco);
    const int ch = (int)(r / nco);
    const int co = cb * 64 + nt * 16 + n, ci = ch * 4 + k;
    double g[3][3];
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q)
        g[p][q] = (co < Cout && ci < Cin) ? (double)w[(((size_t)co * Cin + ci) * 3 + kd) * 9 + p * 3 + q] : 0.0;
    double row[3];
    for (int q = 0; q < 3; ++q) row[q] = by == 0 ? g[0][q] : (by == 1 ? g[1][q] : (by == 2 ? g[0][q] + g[2][q] : g[2][q]));
    float* dst = wpk + i * 4;
    dst[0] = (float)row[0];
    dst[1] = (float)row[1];
    dst[2] = (float)(row[0] + row[2]);
    dst[3] = (float)row[2];
  }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

int cu_count() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

}  // namespace

extern "C" int dv_conv3d_s2pp_supported(int Cin, int Cout, int D, int H, int W) {
  // 64 output channels per block (narrower layers would pay for empty N-tiles; the direct kernel has its own tilings);
  // rows travel as 16-byte quads (W % 4 == 0); 32-bit byte offsets across the four channels of a chunk
  return (Cin > 0 && Cout >= 64 && Cout % 64 == 0 && D > 0 && H > 0 && W > 0 && W % 4 == 0 &&
          (size_t)D * H * W * sizeof(float) <= 0x3fffffffull) ? 1 : 0;
}

extern "C" size_t dv_conv3d_s2pp_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)cdiv(Cin, 4) * cdiv(Cout, 64) * pp::U_CHUNK;
}

extern "C" int dv_conv3d_s2pp_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout, dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  const int nchunk = cdiv(Cin, 4), nco = cdiv(Cout, 64);
  const size_t total = (size_t)nchunk * nco * 3 * 4 * 4 * 4 * 16;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_s2pp_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpacked, Cin, Cout,
                     nchunk, nco);
  return dv_launch_status();
}

extern "C" int dv_conv3d_s2pp_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                                  const float* residual, float* out, int B, int Cin, int D, int H, int W, int Cout,
                                  int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(dv_conv3d_s2pp_supported(Cin, Cout, D, H, W), DV_ERR_UNSUPPORTED);
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_LEAKY, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(dv_aligned16(wpacked) && dv_aligned16(in), DV_ERR_ALIGN);
  const size_t wbytes = dv_conv3d_s2pp_packed_floats(Cin, Cout) * sizeof(float);
  DV_REQUIRE(wbytes <= 0x7fffffffull, DV_ERR_SHAPE);
  PPArgs a;
  a.in = in; a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.residual = residual; a.out = out;
  a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
  a.Do = (D - 1) / 2 + 1; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
  a.vec_ok = (a.Wo % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual));
  a.wpk_bytes = (unsigned)wbytes;
  hipStream_t s = (hipStream_t)stream;
  auto launch = [&](auto shape) {
    constexpr int SHAPE = decltype(shape)::value;
    using G = PG<SHAPE>;
    a.ntx = cdiv(a.Wo, 2 * G::TC); a.nty = cdiv(a.Ho, 2 * G::TR); a.ntz = cdiv(a.Do, G::TD); a.nco = cdiv(Cout, 64);
    const long long tiles = (long long)B * a.nco * a.ntz * a.nty * a.ntx;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return (int)DV_ERR_SHAPE;
    a.ntiles = (int)tiles;
    const long long slots = cu_count();                             // one resident block (two tiles at a time) per CU
    const long long pairs = (tiles + 1) / 2;
    const unsigned blocks = (unsigned)(pairs < slots ? pairs : slots);
    hipLaunchKernelGGL((conv3d_s2pp_kernel<SHAPE>), dim3(blocks), dim3(512 + 64 * DV_PP_LOADERS), 0, s, a);
    return dv_launch_status();
  };
  // shape of a wave's 16 patches: 8 x 8 outputs, or 4 x 16 where that pads the output plane less.  (16 x 4 outputs pad
  // the 60-wide planes least but lose to both: their 48-byte row pieces waste most of every 128-byte line -- 0.66 against
  // 0.52 ms on 64 -> 128.)  The choice looks at one batch item only (a shard of a batch reproduces the batch's bits) -- and
  // every shape sums an output in the same order anyway (chunk by chunk, kd, operand row, operand column).
  auto padded = [&](int tw, int th) { return (long long)cdiv(a.Wo, tw) * tw * cdiv(a.Ho, th) * th; };
  const int shape = padded(16, 4) < padded(8, 8) ? 0 : 1;
  if (shape == 1) return launch(std::integral_constant<int, 1>{});
  return launch(std::integral_constant<int, 0>{});
}
