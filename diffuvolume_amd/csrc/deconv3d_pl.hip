// K5p: ConvTranspose3d(k = 3, stride 2, padding 1, output_padding 1) + BN + skip + activation (hourglass conv5 / conv6,
// SceneFlow/models/acv_ddim.py:74-80, :91-92; KITTI12/models/pwcnet_ddim.py:131-205) in the launch shape of
// conv3d_s2pp.hip: ONE persistent block per CU, eight MFMA waves + four loader waves.  Same arithmetic as deconv3d.hip
// (parity decomposition: every tap feeds one of the 8 output parity classes, direct fp32 on v_mfma_f32_16x16x4_f32,
// the fused `redir` 1x1x1 convolution of the skip tensor as extra K-steps), same packed weights.
//
// What is different from deconv3d.hip's one-tile blocks (its in-kernel stamps: 21.5 % of a block's life is epilogue, 11 %
// first fetch + commit phases, and the CU's partner block fills those at the rate of one wave per SIMD):
//   * the MFMA waves only read LDS, issue MFMAs and store; bricks and weight images arrive by LDS-DMA from loader waves
//     (`buffer_load_dwordx4 ... lds`, the range check writes the zero padding), double-buffered, one block barrier per step
//     (a layer with ONE block of output channels keeps one ring of four weight images for both groups instead);
//   * a wave holds 64 accumulators (one input row x 16 positions x 32 output channels x 8 classes) instead of 128, so
//     twelve waves of <= 168 registers fit a CU;
//   * the eight MFMA waves are two GROUPS of four (a group = one tile of 1 x 2 x 32 input positions = 2 x 4 x 64 outputs
//     x 32 output channels) that walk their tile lists NE steps apart: a step is one chunk of 8 input channels for a group
//     that computes, or half an epilogue for a group that stores.  While one group stores, the other one computes -- the
//     epilogue's stores leave the CU beside the partner group's MFMA stream by construction instead of by the luck of two
//     independent blocks' phases, and no tile has a prologue: its first brick is copied during the previous tile's
//     epilogue;
//   * the accumulators are TRANSPOSED (the weights are the MFMA's A operand, the positions its B operand): element e of lane
//     (kq, j) is output channel 4 kq + e at position j, so the two x parities of a lane are neighbours in memory and sixteen
//     lanes write one whole 128-byte line with one 8-byte store each (the one-tile kernel's 16-byte stores touch sixteen
//     half-used lines per instruction).
// Measured, batch 8, same box, alternating (tools/ab_deconv.py): 64 -> 32 + redir at 24 x 64 x 120 1.875 -> 1.67 ms, 128 -> 64
// + redir at 12 x 32 x 60 0.860 -> 0.765 ms.  How it got there, the in-kernel stamps and what lost:
// profiles/r06_deconv_pl_experiments.txt (the instrumented source: profiles/attic/r06_deconv_pl/).
#include <atomic>
#include <type_traits>

#include "dv_common.h"

namespace {

// test hook (dv_deconv3d_pl_set_max_blocks): 0 = one block per CU; n > 0 = at most n blocks, so that small test volumes walk
// long tile lists (the result does not depend on the grid: every output is summed in the same order by whichever block)
std::atomic<int> g_pl_max_blocks{0};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace pl {
constexpr int TH = 2, TW = 32, KC = 8, NT = 2, COUT = 16 * NT;
constexpr int RQ = TW / 4 + 1;                 // quads of a brick row: x0 .. x0 + 35 (the +1 halo column is the first float of the last quad)
constexpr int RS = 4 * RQ;                     // row stride (floats)
constexpr int IY = TH + 1, ROWS = 2 * IY;      // two input planes x three rows
constexpr int QPC = 60;                        // quads per channel: 54 loaded rows' quads + 6 zero quads; channel stride 240 floats == 48 (mod 64):
constexpr int CS = 4 * QPC;                    //   the four k-lanes of an operand read fall on disjoint banks
static_assert(ROWS * RQ <= QPC && CS % 64 == 48, "brick image");
constexpr int BRICK_Q = KC * QPC;              // 480 quads per chunk
constexpr int BRICK_P = (BRICK_Q + 63) / 64;   // 8 DMA pieces (the last one half empty: its tail writes zeros behind the brick)
constexpr int BRICK_FLOATS = BRICK_P * 256;
constexpr int W_FLOATS = 27 * KC * COUT;       // the packed image of deconv3d.hip: [tap][kq][j][ks][n]
constexpr int W_P = W_FLOATS / 256;            // 27 pieces
static_assert(W_FLOATS % 256 == 0, "weight image in 1-KB pieces");
constexpr int SKC = 4;                         // skip channels that ride on one chunk
constexpr int SK_FLOATS = SKC * 4 * TW;        // a wave's skip tile: 4 channels x 4 output rows x 32 output columns
constexpr int NL = 4;                          // loader waves: one per SIMD (3 -> 4: -3 %; 5, 6, 8: no further gain)
constexpr int NE = 2;                          // steps of an epilogue = the distance between the two groups (1 and 4: +1 %)
constexpr int NB = (BRICK_P + NL - 1) / NL, NW = (W_P + NL - 1) / NL;
constexpr int MAXCO = 256;                     // output channels (scale / bias table in LDS)
static_assert((2 * 2 * (W_FLOATS + BRICK_FLOATS) + 8 * SK_FLOATS + 2 * MAXCO) * 4 <= 160 * 1024, "one block per CU");
}  // namespace pl

// One 1-KB LDS-DMA piece (lane l: 16 bytes from buffer offset voff + soff to LDS address lds + 16 l) as inline assembly, for
// the MFMA waves' own skip tiles: the compiler's wait-count pass does not see it, so it neither puts s_waitcnt vmcnt in front
// of the LDS reads that follow (it did, whenever the builtin was not inside a run-time branch: the copy's latency exposed at
// the top of every chunk) nor a draining fence at the barrier; the one wait this copy needs is written out where the tile is
// read.  (Untracked OLDER operations only make the compiler's own counted waits wait longer than it thinks: in-order counter.)
__device__ __forceinline__ void pl_dma16(i32x4 rsrc, unsigned lds, unsigned voff, unsigned soff) {
  unsigned m0_saved;             // M0 is reserved by the compiler: hand it back as found (as conv3d_wino.hip does)
  // (s_nop 0: one wait state between a scalar write of M0 and the LDS-DMA that reads it; tests/test_isa_lint.py)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved) : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__host__ __device__ constexpr int pl_par(int k) { return (k + 1) & 1; }
__host__ __device__ constexpr int pl_off(int k) { return k == 0 ? 1 : 0; }

struct PLArgs {
  const float* in;
  const float* wpk;
  const float* ch_scale;
  const float* ch_bias;
  const float* residual;
  const float* skip;
  const float* rw;
  float* out;
  int B, Cin, D, H, W, Cout, Cskip;
  int ntx, nty, nco, nsp;       // nsp: spatial tiles = B * D * nty * ntx
  int act;
  unsigned wpk_bytes;
};

struct PLTile { int tcb, xi, yi, z0, y0, x0, b; unsigned st; bool valid; };

// AM: 0 = ReLU, 1 = max(v, slope v) (identity / LeakyReLU), 2 = Mish
template <bool SKIP, bool RES, int AM>
__global__ __launch_bounds__(512 + 64 * pl::NL, 1) void deconv3d_pl_kernel(PLArgs a) {
  using namespace pl;
  // (separate arrays: the MFMA waves' own LDS-DMA goes to sk_s only)
  __shared__ __attribute__((aligned(16))) float w_s[4][W_FLOATS];           // [group][buffer], or ONE ring of four (shared)
  __shared__ __attribute__((aligned(16))) float in_s[2][2][BRICK_FLOATS];
  __shared__ __attribute__((aligned(16))) float sk_s[8][SK_FLOATS];
  // BN scale / bias of every output channel.  They are read per tile; as vector-memory loads they would share the counter
  // with the epilogue's stores (gfx9: stores count in vmcnt), and the compiler's conservative wait for them in front of an
  // epilogue row also waits for the previous row's stores to be acknowledged (measured: 6-9 k cycles per half epilogue).
  __shared__ __attribute__((aligned(16))) float sb_s[2][MAXCO];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < a.Cout; i += 512 + 64 * NL) {
    sb_s[0][i] = a.ch_scale ? a.ch_scale[i] : 1.f;
    sb_s[1][i] = a.ch_bias ? a.ch_bias[i] : 0.f;
  }

  // ---- tile list (as conv3d_s2pp.hip): tiles in the order output-channel block fastest, then x, y, z, batch; tiles 2 P and
  // 2 P + 1 are PAIR P (the two channel blocks of one brick where the layer has an even number of them, else x neighbours):
  // group g of a block computes tile 2 P + g.  Every XCD owns a contiguous slab of the pair order -- neighbours share
  // bricks, halos and skip rows in its L2 -- and its blocks walk it round-robin.  (A launch of fewer than 8 blocks, which
  // only the test hook produces, has that many slabs instead of 8.) ----
  const unsigned nblk = gridDim.x, nx = nblk < 8u ? nblk : 8u, xcd = blockIdx.x % nx, bidx = blockIdx.x / nx;
  const unsigned nbx = nblk / nx + (xcd < nblk % nx ? 1u : 0u);
  const unsigned ntiles = (unsigned)a.nsp * (unsigned)a.nco;
  const unsigned npairs = (ntiles + 1u) >> 1;
  const unsigned tq = npairs / nx, trm = npairs % nx;
  const unsigned slab_lo = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq;
  const unsigned slab_n = tq + (xcd < trm ? 1u : 0u);
  const int my_pairs = __builtin_amdgcn_readfirstlane(bidx < slab_n ? (int)((slab_n - bidx + nbx - 1) / nbx) : 0);
  if (my_pairs == 0) return;
  // a group's tiles are 2 * nbx apart in the linear order: the first one is decoded by division, the later ones by adding the
  // stride's digits with carries (scalar adds and compares instead of five divisions per tile)
  auto rfl = [](unsigned v) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  auto decode = [&](unsigned t, int (&d)[5]) __attribute__((always_inline)) {
    d[0] = (int)rfl(t % a.nco); t /= a.nco;
    d[1] = (int)rfl(t % a.ntx); t /= a.ntx;
    d[2] = (int)rfl(t % a.nty); t /= a.nty;
    d[3] = (int)rfl(t % a.D);
    d[4] = (int)rfl(t / a.D);
  };
  int sd[5];
  decode(2u * nbx, sd);
  auto tile_first = [&](int g) __attribute__((always_inline)) {
    PLTile r;
    r.st = 2u * (slab_lo + bidx) + (unsigned)g;
    r.valid = r.st < ntiles;
    int d[5];
    decode(r.valid ? r.st : ntiles - 1u, d);
    r.tcb = d[0]; r.xi = d[1]; r.yi = d[2]; r.z0 = d[3]; r.b = d[4];
    r.x0 = r.xi * TW; r.y0 = r.yi * TH;
    return r;
  };
  auto tile_next = [&](PLTile& r) __attribute__((always_inline)) {
    r.st += 2u * nbx;
    r.valid = r.st < ntiles;                             // (past the end the digits are never used)
    int c;
    r.tcb += sd[0];      c = r.tcb >= a.nco; r.tcb -= c ? a.nco : 0;
    r.xi += sd[1] + c;   c = r.xi >= a.ntx;  r.xi -= c ? a.ntx : 0;
    r.yi += sd[2] + c;   c = r.yi >= a.nty;  r.yi -= c ? a.nty : 0;
    r.z0 += sd[3] + c;   c = r.z0 >= a.D;    r.z0 -= c ? a.D : 0;
    r.b += sd[4] + c;
    r.x0 = r.xi * TW; r.y0 = r.yi * TH;
  };

  const size_t plane = (size_t)a.H * a.W, vol = (size_t)a.D * plane;
  const unsigned vol_bytes = (unsigned)__builtin_amdgcn_readfirstlane((int)(vol * sizeof(float)));
  const int nchunk = a.Cin / KC;
  const int PERIOD = nchunk + NE;                      // steps of a tile: its chunks, then the epilogue halves
  const int n_total = my_pairs * PERIOD + NE;          // group 1 runs NE steps behind group 0
  // one block of output channels: both groups walk the SAME weight chunks, NE steps apart -> one ring of four images, copied once
  // (-1 % on the 64 -> 32 layer: 27 instead of 54 weight pieces per step)
  const bool shared = a.nco == 1;

  // Barrier protocol (all twelve waves): P, then one per step.  In step s a computing group reads the buffers of its chunk
  // (its compute steps alternate between the two); the loaders copy what step s + 1 needs into the buffers step s does not
  // read and wait for their copies before they arrive at the step's barrier.  The barriers are bare s_barrier instructions:
  // __syncthreads() comes with a fence that drains vmcnt while an LDS-DMA is pending, i.e. an MFMA wave would wait at every
  // step for its epilogue's stores to be acknowledged.  What has to be ordered is ordered by hand: LDS reads are consumed
  // (lgkmcnt(0)) and the loaders' copies have landed (vmcnt(0)) before the barrier.
  if (wave >= 8) {
    // =========================== loader waves: global -> LDS by DMA, nothing else ===========================
    const int li = wave - 8;
    const uint64_t wb = reinterpret_cast<uint64_t>(a.wpk);
    const uint64_t wbs = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wb) |
                         ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wb >> 32)) << 32);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(wbs), 0, (int)a.wpk_bytes, 0x00020000);
    unsigned sob[2][NB];                                 // this lane's byte offset in each of the loader's brick pieces, per group
    auto plan = [&](const PLTile& t, unsigned (&so)[NB]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int it = 64 * (li + NL * i) + lane;
        const int cl = it / QPC, r1 = it - cl * QPC;
        const int row = r1 / RQ, q = r1 - row * RQ;
        const int zz = row / IY, yy = row - zz * IY;
        const int z = t.z0 + zz, y = t.y0 + yy, x = t.x0 + 4 * q;
        const bool ok = t.valid && it < BRICK_Q && row < ROWS && z < a.D && y < a.H && x < a.W;   // W % 4 == 0: a quad is inside or outside
        so[i] = ok ? (unsigned)cl * vol_bytes + (unsigned)((z * a.H + y) * a.W + x) * 4u : 0xfffffff0u;
      }
    };
    auto dma = [&](const PLTile& t, const unsigned (&so)[NB], int c, float* bdst, float* wdst, bool with_w) __attribute__((always_inline)) {
      const uint64_t ba = reinterpret_cast<uint64_t>(a.in + ((size_t)t.b * a.Cin + (size_t)c * KC) * vol);
      const uint64_t bu = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba) |
                          ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ba >> 32)) << 32);
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(bu), 0,
                                                        __builtin_amdgcn_readfirstlane((int)((unsigned)KC * vol_bytes)), 0x00020000);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (li + NL * i < BRICK_P)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(bdst + (li + NL * i) * 256), 16,
                                                   (int)so[i], 0, 0, 0);
      const int wbase = (c * a.nco + t.tcb) * (W_FLOATS * 4);
#pragma unroll
      for (int i = 0; i < NW; ++i)
        if (with_w && li + NL * i < W_P)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(wdst + (li + NL * i) * 256), 16,
                                                   lane * 16, wbase + (li + NL * i) * 1024, 0, 0);
    };
    // per group: the tile and phase of the NEXT step to prepare
    PLTile ft[2] = {tile_first(0), tile_first(1)};
    int fk[2] = {0, 0}, fp[2] = {0, 0}, fcc[2] = {0, 0};
    plan(ft[0], sob[0]);
    plan(ft[1], sob[1]);
    auto prepare = [&](int s1) __attribute__((always_inline)) {       // the copies step s1 reads
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (s1 < g * NE || fk[g] >= my_pairs) continue;
        if (fp[g] < nchunk) {
          if (ft[g].valid) dma(ft[g], sob[g], fp[g], in_s[g][fcc[g] & 1], w_s[shared ? (fcc[g] & 3) : 2 * g + (fcc[g] & 1)], !shared || g == 0);
          ++fcc[g];
        }
        if (++fp[g] == PERIOD) {
          fp[g] = 0;
          if (++fk[g] < my_pairs) { tile_next(ft[g]); plan(ft[g], sob[g]); }
        }
      }
    };
    prepare(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");          // P
#pragma unroll 1
    for (int s = 0; s < n_total; ++s) {
      if (s + 1 < n_total) prepare(s + 1);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                   // B_s
    }
    return;
  }

  // =========================== MFMA waves ===========================
  const int g = wave >> 2, yl = (wave >> 1) & 1, mw = wave & 1;
  const int j = lane & 15, kq = lane >> 4;
  float* const sk = sk_s[wave];
  const int a_off = kq * CS + yl * RS + mw * 16 + j;     // this lane's input element (k-step 0, shift 0) inside a brick
  const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
  const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
  const int nsk = SKIP ? a.Cskip / SKC : 0;

  // acc[class][n]: element e of lane (kq, j) = output channel n * 16 + 4 kq + e of the block, input position j of the wave's
  // sixteen, parity class (pz, py, px)
  f32x4 acc[8][NT];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[c][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // ---- per-tile state ----
  PLTile cur = tile_first(g);
  unsigned skv = 0;                                      // this lane's byte offset in a skip DMA piece (channel pair 0)
  f32x4 sc4[NT], bi4[NT];
  unsigned loff = 0;
  bool row_ok = false, x_ok = false;
  auto tile_setup = [&]() __attribute__((always_inline)) {
    const int yi = cur.y0 + yl, xi = cur.x0 + mw * 16;
    row_ok = cur.valid && yi < a.H;
    x_ok = row_ok && xi + j < a.W;
    loff = (unsigned)(((size_t)(cur.tcb * COUT + 4 * kq) * ovol + 2 * (xi + j)) * sizeof(float));
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      sc4[n] = *reinterpret_cast<const f32x4*>(&sb_s[0][cur.tcb * COUT + n * 16 + 4 * kq]);
      bi4[n] = *reinterpret_cast<const f32x4*>(&sb_s[1][cur.tcb * COUT + n * 16 + 4 * kq]);
    }
    if (SKIP) {
      // skip tile image: quad Q = (kq >> 1) * 64 + row * 16 + (kq & 1) * 8 + cq  (row = (pz, py), cq = column quad): the 8-byte
      // operand reads of a half wave (two k-lanes x 16 positions) then cover 32 distinct bank pairs.  DMA piece p = kq >> 1,
      // lane = Q % 64.
      const int row = lane >> 4, kl = (lane >> 3) & 1, cq = lane & 7;
      const int oz = 2 * cur.z0 + (row >> 1), oy = 2 * yi + (row & 1), ox = 2 * xi + 4 * cq;
      const bool ok = row_ok && ox < Wo;
      skv = ok ? (unsigned)(((size_t)kl * ovol + ((size_t)oz * Ho + oy) * Wo + ox) * sizeof(float)) : 0xfffffff0u;
    }
  };

  // ---- one chunk of 8 input channels: 27 taps x 2 k-steps x 2 channel halves, then the redir k-step of this chunk ----
  auto compute = [&](int c, int buf, int wslot) __attribute__((always_inline)) {
    // (the redir code is NOT wrapped in a run-time `c < nsk`: the load of `bw` and the wait for it must lie on ONE path, or
    // the compiler has to assume a pending load into those registers ever after and puts s_waitcnt vmcnt(0) -- which also
    // drains the stores -- in front of every later write to them.  A chunk beyond the skip channels copies zeros -- the
    // range check -- and multiplies them: for the hourglass layers every chunk carries skip channels.)
    const bool has_sk = c < nsk;
    const float* ib = in_s[g][buf] + a_off;
    const float* wbp = w_s[wslot] + lane * 4;
    float bw[NT];
    if (SKIP) {
      const uint64_t ba = reinterpret_cast<uint64_t>(a.skip + (size_t)cur.b * a.Cskip * ovol);
      i32x4 rs;
      rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
      rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ba >> 32));     // stride 0: raw buffer
      rs[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)a.Cskip * (unsigned)(ovol * sizeof(float))));
      rs[3] = 0x00020000;
      const unsigned skl = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)sk);
#pragma unroll
      for (int p = 0; p < 2; ++p)
        pl_dma16(rs, skl + p * 1024, has_sk ? skv : 0xfffffff0u, (unsigned)(c * SKC + 2 * p) * (unsigned)(ovol * sizeof(float)));
#pragma unroll
      for (int n = 0; n < NT; ++n)
        bw[n] = a.rw[(size_t)(cur.tcb * COUT + n * 16 + j) * a.Cskip + (has_sk ? c : 0) * SKC + kq];
    }
    float av[2][2][2][2];                               // [oz][oy][ox][ks], in the order the taps ask for them
#pragma unroll
    for (int oz = 1; oz >= 0; --oz)
#pragma unroll
      for (int oy = 1; oy >= 0; --oy)
#pragma unroll
        for (int ox = 1; ox >= 0; --ox)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) av[oz][oy][ox][ks] = ib[(oz * IY + oy) * RS + ox + ks * 4 * CS];
    f32x4 bq[3];
    bq[0] = *reinterpret_cast<const f32x4*>(wbp);
    bq[1] = *reinterpret_cast<const f32x4*>(wbp + 256);
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int tap = (kz * 3 + ky) * 3 + kx;
          const int cls = (pl_par(kz) << 2) | (pl_par(ky) << 1) | pl_par(kx);
          if (tap + 2 < 27) bq[(tap + 2) % 3] = *reinterpret_cast<const f32x4*>(wbp + (tap + 2) * 256);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[cls][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[tap % 3][ks * NT + n], av[pl_off(kz)][pl_off(ky)][pl_off(kx)][ks],
                                                                  acc[cls][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    if (SKIP) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's skip tile has landed (it was requested 108 MFMAs ago)
      const float* skr = sk + ((kq >> 1) * 64 + (kq & 1) * 8 + (j >> 1)) * 4 + (j & 1) * 2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x2 sa = *reinterpret_cast<const f32x2*>(skr + r * 64);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[(r << 1) | px][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[n], sa[px], acc[(r << 1) | px][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- one output row (pz, py) = (R >> 1, R & 1) of this wave's input row: the px = 0 / 1 classes are the two neighbours
  // 2 j, 2 j + 1 of the row: one 8-byte store per lane and channel, sixteen lanes = one whole 128-byte line ----
  const float slope = a.act == DV_ACT_LEAKY ? 0.01f : 1.f;
  auto epilogue = [&](auto rc) __attribute__((always_inline)) {
    constexpr int R = decltype(rc)::value;
    if (!row_ok) return;
    const size_t ro = (size_t)cur.b * a.Cout * ovol + (size_t)(2 * cur.z0 + (R >> 1)) * oplane +
                      (size_t)(2 * (cur.y0 + yl) + (R & 1)) * Wo;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const size_t so = ro + (size_t)(n * 16 + e) * ovol;                        // scalar part of the address
        char* orow = reinterpret_cast<char*>(a.out + so);
        f32x2 v;
        v[0] = fmaf(acc[R << 1][n][e], sc4[n][e], bi4[n][e]);
        v[1] = fmaf(acc[(R << 1) | 1][n][e], sc4[n][e], bi4[n][e]);
        if (RES) {
          f32x2 rr = (f32x2){0.f, 0.f};
          if (x_ok) rr = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(a.residual + so) + loff);
          v += rr;
        }
        if (AM == 0) {                                   // max(v, v * 0): NaN stays NaN
          v = __builtin_elementwise_max(v, v * 0.f);
        } else {
          v[0] = AM == 2 ? dv_act(v[0], DV_ACT_MISH) : fmaxf(v[0], v[0] * slope);
          v[1] = AM == 2 ? dv_act(v[1], DV_ACT_MISH) : fmaxf(v[1], v[1] * slope);
        }
        if (x_ok) *reinterpret_cast<f32x2*>(orow + loff) = v;
      }
  };

  zero_acc();
  int k = 0, p = 0, cc = 0;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // P
  tile_setup();
#pragma unroll 1
  for (int s = 0; s < n_total; ++s) {
    if (s >= g * NE && k < my_pairs) {
      if (p < nchunk) {
        if (cur.valid) compute(p, cc & 1, shared ? (cc & 3) : 2 * g + (cc & 1));
        ++cc;
      } else if (p == nchunk) {                          // (NE == 2: the output planes pz = 0, then pz = 1)
        epilogue(std::integral_constant<int, 0>{});
        epilogue(std::integral_constant<int, 1>{});
      } else {
        epilogue(std::integral_constant<int, 2>{});
        epilogue(std::integral_constant<int, 3>{});
      }
      if (++p == PERIOD) {
        p = 0;
        if (++k < my_pairs) { tile_next(cur); tile_setup(); }
        zero_acc();
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // B_s
  }
}

inline int pl_cdiv(int a, int b) { return (a + b - 1) / b; }

int pl_cu_count() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  static std::atomic<int> cache[64];                     // per device (a DataParallel-style caller has several)
  if (dev >= 0 && dev < 64 && (v = cache[dev].load(std::memory_order_relaxed)) > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
  if (dev >= 0 && dev < 64) cache[dev].store(v, std::memory_order_relaxed);
  return v;
}

template <bool SKIP, bool RES, int AM>
int pl_launch(const PLArgs& a, hipStream_t s) {
  const int cap = g_pl_max_blocks.load(std::memory_order_relaxed);
  const long long slots = cap > 0 ? cap : pl_cu_count();
  const long long pairs = ((long long)a.nsp * a.nco + 1) / 2;
  const unsigned blocks = (unsigned)(pairs < slots ? pairs : slots);
  hipLaunchKernelGGL((deconv3d_pl_kernel<SKIP, RES, AM>), dim3(blocks), dim3(512 + 64 * pl::NL), 0, s, a);
  return dv_launch_status();
}

}  // namespace

extern "C" int dv_deconv3d_pl_set_max_blocks(int n) {
  DV_REQUIRE(n >= 0, DV_ERR_SHAPE);
  g_pl_max_blocks.store(n, std::memory_order_relaxed);
  return DV_OK;
}

// shapes the persistent kernel takes (everything else stays on deconv3d_mfma_kernel): whole 8-channel chunks and 32-channel
// output blocks, rows that travel as 16-byte quads, 32-bit byte offsets inside a batch item, skip channels four per chunk
extern "C" int dv_deconv3d_pl_supported(int Cin, int Cout, int D, int H, int W, int Cskip) {
  if (Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0 || Cskip < 0) return 0;
  if (Cin % pl::KC || Cout % pl::COUT || Cout > pl::MAXCO || W % 4) return 0;
  if (Cskip % pl::SKC || Cskip / pl::SKC > Cin / pl::KC) return 0;
  const size_t vol = (size_t)D * H * W;
  if (vol * sizeof(float) * pl::KC > 0x7fffffffull) return 0;
  if ((size_t)Cout * 8 * vol * sizeof(float) > 0xffffffffull) return 0;
  if ((size_t)Cskip * 8 * vol * sizeof(float) > 0xffffffffull) return 0;
  return 1;
}

// internal entry (deconv3d.hip routes to it): same packed weights as dv_deconv3d_pack_weights_f32
int dv_deconv3d_pl_run(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias, const float* residual,
                       const float* skip, const float* rw, float* out, int B, int Cin, int D, int H, int W, int Cout, int Cskip,
                       int act, size_t wpk_floats, hipStream_t s) {
  PLArgs a;
  a.in = in; a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.residual = residual; a.skip = skip; a.rw = rw;
  a.out = out; a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.Cskip = skip ? Cskip : 0; a.act = act;
  a.ntx = pl_cdiv(W, pl::TW); a.nty = pl_cdiv(H, pl::TH); a.nco = Cout / pl::COUT;
  const long long tiles = (long long)B * D * a.nty * a.ntx;
  if (tiles <= 0 || tiles * a.nco > 0x3fffffffLL) return DV_ERR_SHAPE;
  a.nsp = (int)tiles;
  if (wpk_floats * sizeof(float) > 0x7fffffffull) return DV_ERR_SHAPE;
  a.wpk_bytes = (unsigned)(wpk_floats * sizeof(float));
  const int am = act == DV_ACT_RELU ? 0 : (act == DV_ACT_MISH ? 2 : 1);
  if (skip) {
    if (am == 0) return pl_launch<true, false, 0>(a, s);
    if (am == 2) return pl_launch<true, false, 2>(a, s);
    return pl_launch<true, false, 1>(a, s);
  }
  if (residual) {
    if (am == 0) return pl_launch<false, true, 0>(a, s);
    if (am == 2) return pl_launch<false, true, 2>(a, s);
    return pl_launch<false, true, 1>(a, s);
  }
  if (am == 0) return pl_launch<false, false, 0>(a, s);
  if (am == 2) return pl_launch<false, false, 2>(a, s);
  return pl_launch<false, false, 1>(a, s);
}
