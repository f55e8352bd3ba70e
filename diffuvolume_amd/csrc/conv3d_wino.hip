// K4w: the 3x3x3 stride-1 aggregation convolution (convbn_3d, SceneFlow/models/submodule.py:94-97; the
// dres / hourglass / classifier layers of acv_ddim.py:60-70, :200-222) with the two in-plane taps done by the
// Winograd minimal-filtering transform F(2x2, 3x3) and the depth taps kept direct:
//   per depth tap kd and input channel c:   M[p] += V[p](x) * U[p](w),  p = 16 transform positions
//   V = Bt d B  of every 4x4 input patch (stride 2),  U = G g Gt  (packed once),  Y = At M A  (2x2 outputs)
// so a 2x2 output tile costs 16 multiplies per (kd, c) instead of 36: 2.25x fewer MFMA flops than the direct
// implicit GEMM of conv3d.hip, still on the exact-fp32 instruction v_mfma_f32_16x16x4_f32.  The transforms
// only add and subtract (Bt, At entries are 0/+-1; G has the 1/2 folded into the packed weights), so the
// rounding is that of a few extra fp32 additions per product.
//
// Block = 4 waves = a 4(z) x 4(y) x 16(x) output brick x 32 output channels; wave w owns plane z0+w: its MFMA
// M index is the 16 tiles (2 tile rows x 8 tile columns) of that plane, N = 16 output channels, K = 4 input
// channels per step.  Per chunk of 4 input channels the haloed raw brick (6 x 6 x 18) goes through LDS, is
// transformed into V[c][plane 0..5][tile][16 positions] (each plane feeds the three waves that see it as
// kd = 0,1,2), and the weights of the chunk are copied next to it; a wave then runs 3 x 16 x 2 MFMAs whose
// A / B fragments are ds_read_b128 of four positions each.  The raw brick of chunk c+1 is written while chunk c
// computes, its global loads are issued a further chunk ahead.

#include <type_traits>

#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int round_up_mod64_32(int v) {   // smallest v' >= v with v' % 64 == 32
  const int r = v % 64;
  return r <= 32 ? v + (32 - r) : v + (64 - r) + 32;
}

// MTW = M-tiles (planes) per wave: 1 -> 4 planes per block, 128 accumulator registers, two blocks per CU;
// 2 -> 8 planes per block (wave w owns planes w and w+4), 256 accumulator registers (the AGPR half of the
// 512-register file), one block per CU: every B fragment feeds two MFMAs, half the weight DMA / barriers / blocks.
// SHAPE = how the 16 Winograd tiles of a wave's M index lie in its plane: 0 -> 2 tile rows x 8 tile columns
// (4 x 16 outputs), 1 -> 4 x 4 (8 x 8 outputs), 2 -> 8 x 2 (16 x 4 outputs).  The host picks the shape that pads
// the plane least: a 120-wide plane is 7.5 tiles of 16 but exactly 15 tiles of 8.
template <int MTW_, int SHAPE_ = 0>
struct WG {
  static constexpr int MTW = MTW_, SHAPE = SHAPE_;
  static constexpr int TR = SHAPE == 0 ? 2 : (SHAPE == 1 ? 4 : 8), TC = 16 / TR;     // tile rows / columns per wave
  static constexpr int KC = 4, NT = 2, TD = 4 * MTW, TH = 2 * TR, TW = 2 * TC;
  static constexpr int IZ = TD + 2, IY = TH + 2, IX = TW + 2;
  static constexpr int PRAW = IZ * IY * IX;          // raw positions per channel (648 / 600 / 648)
  // raw brick [c][z][y][RX].  Bank plan of the patch reads: a 32-lane half of a ds_read_b64 holds 16 tiles x 2
  // channels; with a channel stride = 32 mod 64 the two channels take disjoint halves of the 64 banks, and inside a
  // half the tile columns (2 floats apart) and tile rows (2*RX apart) have to tile 32 banks without overlap:
  //   2 x 8: columns cover 16 banks, 2*RX = 48 puts the second tile row on the other 16
  //   4 x 4: columns cover 8 banks, 2*RX = 24 -> rows at 0, 24, 48, 72 = 8 (mod 64)
  //   8 x 2: columns cover 4 banks, 2*RX = 12 -> rows at 0, 12, 24, 36, 48, 60, 8, 20 (mod 64)
  static constexpr int RX = SHAPE == 0 ? 24 : (SHAPE == 1 ? 12 : 6);
  static constexpr int RAWP = round_up_mod64_32(IZ * IY * RX);
  static constexpr int RAW_FLOATS = KC * RAWP;
  static constexpr int NS = (PRAW + 255) / 256;
  // where the staging lanes past the brick put their (zero) value: a padding column of row 0, else the channel tail
  static constexpr int DUMP = RX > IX ? IX : IZ * IY * RX;
  static_assert(RX >= IX && RX % 2 == 0, "row stride");
  static_assert(DUMP < RAWP, "the dump float lies inside the channel");
  static_assert(RAWP % 64 == 32, "bank plan of the patch reads");
};
namespace wg {
constexpr int U_CHUNK = 3 * 2 * 4 * 16 * 16;   // packed floats per (chunk, co block) = the LDS image, 24 KB
// Branch-free chunk body: the staging of the chunks to come is issued whether or not they exist (channels past the end
// are zero-record descriptors, a weight chunk past the end re-copies the last one), so the body is one scheduling region.
#ifndef DV_WINO_BF
#define DV_WINO_BF 1
#endif
constexpr bool BF = DV_WINO_BF;
#ifndef DV_WINO_DMA_G
#define DV_WINO_DMA_G 6
#endif
constexpr int DMA_G = DV_WINO_DMA_G;            // MFMA group of a chunk in which the next chunk's weight DMA is issued
}

struct WinoArgs {
  const float* in;
  const float* wpk;      // [Cin/4][Coutp/32][kd 3][nt 2][k 4][n 16][pos 16]
  const float* ch_scale;
  const float* ch_bias;
  const float* in_scale; // [B,D,H,W] or null
  const float* residual;
  float* out;
  int B, Cin, D, H, W, Cout;
  int ntx, nty, ntz, nco;
  int act;
  int fast_ok;           // W % 4 == 0, 16-byte aligned pointers
};

template <bool HAS_SCALE, int MTW, int SHAPE = 0>
__global__ __launch_bounds__(256, MTW == 1 ? 2 : 1) void conv3d_wino_kernel(WinoArgs a) {
  using G = WG<MTW, SHAPE>;
  constexpr int KC = G::KC, NT = G::NT, TD = G::TD, TH = G::TH, TW = G::TW, IY = G::IY, IX = G::IX, PRAW = G::PRAW;
  constexpr int RX = G::RX, RAWP = G::RAWP, RAW_FLOATS = G::RAW_FLOATS, NS = G::NS, U_CHUNK = wg::U_CHUNK;
  static_assert((2 * U_CHUNK + 2 * RAW_FLOATS) * 4 * (MTW == 1 ? 2 : 1) <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) float smem[2 * U_CHUNK + 2 * RAW_FLOATS];
  float* u_s = smem;
  float* raw_s = smem + 2 * U_CHUNK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;

  // the output-channel slices of a tile are neighbours in the linear order: they run at the same time on the same XCD
  // (dv_xcd_remap), so the input brick they all read comes from HBM once and from that XCD's L2 for the others
  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int tc = t % a.nco; t /= a.nco;
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty; t /= a.nty;
  const int tz = t % a.ntz;
  const int b = t / a.ntz;
  const int x0 = tx * TW, y0 = ty * TH, z0 = tz * TD, co0 = tc * 32;

  // (not zeroed: the first chunk's MFMAs take the inline constant 0 as their C operand -- 128 v_mov less per block, on
  // the issue pipe the MFMAs of the CU's other block need)
  f32x4 acc[MTW][16][NT];

  const size_t plane = (size_t)a.H * a.W;
  const size_t vol = (size_t)a.D * plane;
  const float* inb = a.in + (size_t)b * a.Cin * vol;
  const float* scb = (HAS_SCALE && a.in_scale) ? a.in_scale + (size_t)b * vol : nullptr;

  // ---- raw staging plan: NS positions of the haloed brick per thread, the same for every channel.  The loads
  // are buffer loads (one descriptor per channel, built on the scalar unit): the lane part of the address is a
  // 32-bit offset, and both the zero padding (offset 2^31 for positions outside the volume) and the channel tail
  // (zero records) come out of the hardware range check instead of vector selects ----
  unsigned sob[NS];                 // byte offset in a channel volume
  int lro[NS];                      // float offset in a channel of the LDS brick
  float scl[HAS_SCALE ? NS : 1];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int r = tid + 256 * i;
    const int zz = r / (IY * IX), r2 = r - zz * (IY * IX);
    const int yy = r2 / IX, xx = r2 - yy * IX;
    const int z = z0 - 1 + zz, y = y0 - 1 + yy, x = x0 - 1 + xx;
    const bool ok = r < PRAW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H &&
                    (unsigned)x < (unsigned)a.W;
    const unsigned sp = ok ? (unsigned)((z * a.H + y) * a.W + x) : 0u;
    sob[i] = ok ? sp * 4u : 0x80000000u;
    lro[i] = r < PRAW ? (zz * IY + yy) * RX + xx : G::DUMP;   // lanes past the brick write a float no patch reads
    if (HAS_SCALE) scl[i] = (ok && scb) ? scb[sp] : 1.f;
  }
  const int vol_bytes = __builtin_amdgcn_readfirstlane((int)(vol * sizeof(float)));   // < 2^31 (checked by the host)
  float vin[KC][NS];                  // raw brick of the next chunk, refilled for the one after as soon as it is in LDS
  // channels are fetched strictly in order (chunk after chunk), so the descriptor base is a running scalar pointer
  // (one 64-bit add per channel instead of a 64-bit multiply) and the tail test a running counter
  uint64_t fb = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uint64_t>(inb)) |
                ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(reinterpret_cast<uint64_t>(inb) >> 32)) << 32);
  int fc = 0;
  auto fetch_raw_cl = [&](int /*c0*/, int cl) __attribute__((always_inline)) {
    const bool cok = fc < a.Cin;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(fb), 0, cok ? vol_bytes : 0, 0x00020000);
    fb += (uint64_t)(unsigned)vol_bytes;
    ++fc;
#pragma unroll
    for (int i = 0; i < NS; ++i)
      vin[cl][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)sob[i], 0, 0));
  };
  auto fetch_raw = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int cl = 0; cl < KC; ++cl) fetch_raw_cl(c0, cl);
  };
  auto commit_raw_cl = [&](int cl, float* rb) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NS; ++i) rb[cl * RAWP + lro[i]] = HAS_SCALE ? vin[cl][i] * scl[i] : vin[cl][i];
  };
  auto commit_raw = [&](float* rb) __attribute__((always_inline)) {
#pragma unroll
    for (int cl = 0; cl < KC; ++cl) commit_raw_cl(cl, rb);
  };
  // ---- weights: the packed chunk is the LDS image; LDS-DMA copies it in 1-KB pieces (16 cout rows x 4 position
  // quads), six per wave.  Lane l of a piece lands in 16-byte slot l, so the source quad is XOR-swizzled with the
  // row (slot s of row n holds quad s ^ (n>>2)): the B-fragment ds_read_b128 of 16 rows then covers 16 distinct
  // slots of the 256-byte bank row without padding ----
  const int dma_lo = (lane >> 2) * 16 + (((lane & 3) ^ ((lane >> 4) & 3)) * 4);
  const int dma_voff = dma_lo * 4;                 // the lane part of a piece's source address: constant
  auto dma_u = [&](int c0, float* ub) __attribute__((always_inline)) {
    // piece base on the scalar unit (SGPR pair), lane offset in one loop-invariant VGPR: a 64-bit vector add per piece
    // (v_lshl_add_u64) is an isolated vector-ALU instruction in the MFMA stream -- the matrix pipe drains for it
    // (~60 cycles each, tools/probes/mfma_f32_neighbours.hip)
    const int n_chunk = (a.Cin + KC - 1) / KC;
    const int ch = (!wg::BF || (c0 >> 2) < n_chunk) ? (c0 >> 2) : n_chunk - 1;
    const float* src = a.wpk + ((size_t)ch * a.nco + tc) * U_CHUNK;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int piece = wave + 4 * q;
#ifdef DV_WINO_BUILTIN_DMA
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + dma_lo + piece * 256),
                                       (__attribute__((address_space(3))) void*)(ub + piece * 256), 16, 0, 0);
#else
      // Issued as inline asm rather than through the builtin: the compiler treats an LDS-DMA as an LDS store that any
      // later LDS store may alias and answers the next `ds_write` of the raw brick with s_waitcnt vmcnt(0) -- i.e. the
      // first commit group of every chunk waited for the DMA issued 16 MFMAs earlier.  Its completion is covered by
      // the manual s_waitcnt vmcnt + barrier at the top of the next chunk; not counting it makes the compiler's own
      // vmcnt waits for the raw registers stricter, never looser (vector memory operations complete in order).
      const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(ub + piece * 256);
      const uint64_t gb = reinterpret_cast<uint64_t>(src + piece * 256);
      const uint64_t gbs = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gb) |
                           ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(gb >> 32)) << 32);
      unsigned m0_saved;       // M0 is reserved by the compiler: hand it back as found
      // (s_nop 0: gfx9 needs one wait state between a scalar write of M0 and the LDS-DMA that reads it; the hazard
      // recogniser does not look inside inline asm -- tests/test_isa_lint.py checks the compiled stream)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(m0_saved) : "s"(lds_addr), "v"(dma_voff), "s"(gbs) : "memory");
#endif
    }
  };

  // this lane's 4x4 patch: M index j = tile (row 2*grow + (j&1), column 2*gcol + ((j>>1)&1)) with the 2x2-tile group
  // j>>2 laid out per SHAPE, of plane wave+kd, channel kq;  B rows (kq, j)
  constexpr int GC = G::TC / 2;                       // groups per row of groups: 4 / 2 / 1
  const int p_tr = 2 * ((j >> 2) / GC) + (j & 1), p_tc = 2 * ((j >> 2) % GC) + ((j >> 1) & 1);
  int patch_lo[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) patch_lo[mt] = kq * RAWP + ((wave + 4 * mt) * IY + 2 * p_tr) * RX + 2 * p_tc;
  int b_lo[4];
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) b_lo[p4] = (kq * 16 + j) * 16 + ((p4 ^ ((j >> 2) & 3)) * 4);

  // LDS offsets (bytes from smem) of this chunk's patch / B rows and of the next chunk's raw brick: loop-carried and
  // flipped between the two buffers with one add each per chunk (recomputing them from `cur` cost 28 vector
  // instructions per chunk)
  int po[MTW], bo[4], wo[NS];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) po[mt] = 4 * (2 * U_CHUNK + patch_lo[mt]);
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) bo[p4] = 4 * b_lo[p4];
#pragma unroll
  for (int i = 0; i < NS; ++i) wo[i] = 4 * (2 * U_CHUNK + RAW_FLOATS + lro[i]);
  char* const smem_b = reinterpret_cast<char*>(smem);

  auto commit_next_cl = [&](int cl) __attribute__((always_inline)) {   // into the other buffer, running offsets
#pragma unroll
    for (int i = 0; i < NS; ++i)
      *reinterpret_cast<float*>(smem_b + wo[i] + 4 * cl * RAWP) = HAS_SCALE ? vin[cl][i] * scl[i] : vin[cl][i];
  };

  fetch_raw(0);
  dma_u(0, u_s);
  commit_raw(raw_s);
  if (wg::BF || KC < a.Cin) fetch_raw(KC);
  // one chunk: `vin` holds the raw brick of chunk c0+KC on entry and that of chunk c0+2*KC on exit
  auto chunk = [&](int c0, int cur, auto first_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value;
    // this chunk's weights (DMA, issued at the start of the previous chunk) have to be in LDS; the raw loads issued
    // after them (KC*NS per thread, for chunk c0+KC) may stay in flight.  After the barrier every wave is done with
    // the other pair of buffers and this chunk's raw brick is complete.
    if (wg::BF || c0 + KC < a.Cin) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KC * NS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // The staging of the next chunk (weight DMA, LDS commit of the raw registers, their refill two chunks ahead) is
    // spread over the first MFMA groups below: between two MFMAs of a wave there are issue slots the matrix pipe does
    // not need, and instructions placed there cost nothing, while a staging phase in front of the stream delays the
    // first MFMA of every chunk.
    const bool nxt = wg::BF || c0 + KC < a.Cin, refill = wg::BF || c0 + 2 * KC < a.Cin;
    // MFMA stream: 12 groups (kd, position quad) of 8*MTW MFMAs.  The B fragments of group g+1, the raw patches of
    // the next plane(s) and their transform are written in the shadow of group g; the final order is the compiler's
    // (pinning it with sched_barrier around every group measured 1-2.5 % slower once the staging was spread out).
    f32x2 d[MTW][4][2];
    f32x4 bq[2][NT];
    f32x2 vp[MTW][2][4][2];      // V of the current / next plane: [row][column pair]
    auto load_patch = [&](int kd) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d[mt][r][0] = *reinterpret_cast<const f32x2*>(smem_b + po[mt] + 4 * ((kd * IY + r) * RX));
          d[mt][r][1] = *reinterpret_cast<const f32x2*>(smem_b + po[mt] + 4 * ((kd * IY + r) * RX + 2));
        }
    };
    auto load_b = [&](int g, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
        bq[slot][n] = *reinterpret_cast<const f32x4*>(smem_b + bo[g & 3] + 4 * (((g >> 2) * NT + n) * (KC * 256)));
    };
    // V = Bt d B on packed-fp32 adds: rows as register pairs (two columns at a time), then per row the column
    // combinations (t0-t2, t1+t2) and (t2-t1, t1-t3) as one v_pk_add_f32 each (op_sel picks the halves, neg_* the
    // signs) -- 16 vector instructions per patch instead of 32 adds plus the moves the compiler puts around them.
    // (Round 2 A/B: the same transform as 32 pinned single-register v_add_f32 / v_sub_f32, no moves, measured 1.9 %
    // SLOWER on the 32->32 layer -- 58.5 vs 57.35 ms per step -- so the packed form stays.)
    auto transform = [&](int slot) __attribute__((always_inline)) {
      // The 16 packed adds of a patch are ONE asm statement (one dense burst; measured the same speed as the scheduler's
      // own spreading of them over the MFMA stream, and one statement is easier to reason about).  Rows of Bt d:
      // r0 = d0 - d2, r1 = d1 + d2, r2 = d2 - d1, r3 = d1 - d3 (two column pairs each), then per row the column
      // combinations (t0-t2, t1+t2) and (t2-t1, t1-t3) as one v_pk_add_f32 each (op_sel picks the halves, neg_* the
      // signs) -- 16 vector instructions per patch instead of 32 adds plus the moves the compiler puts around them.
      // (Round 2 A/B: the same transform as 32 pinned single-register v_add_f32 / v_sub_f32, no moves, measured 1.9 %
      // SLOWER on the 32->32 layer -- 58.5 vs 57.35 ms per step -- so the packed form stays.)
      // The results are MFMA A operands.  gfx950 does not interlock a VALU write with an MFMA that reads the register
      // as SrcA/B within the next two issue slots (probed: v_pk_add_f32 / v_add_f32 -> v_mfma back to back or one
      // instruction apart reads the OLD value), and the compiler cannot see through inline asm to add the wait states
      // itself: the closing s_nop makes the burst safe wherever the scheduler puts the consuming MFMA.
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        f32x2 t0, t1, t2, t3;
        asm("v_pk_add_f32 %8, %12, %16 neg_lo:[0,1] neg_hi:[0,1]\n\t"
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
            "v_pk_add_f32 %7, %11, %10 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\t"
            "s_nop 1"
            : "=&v"(vp[mt][slot][0][0]), "=&v"(vp[mt][slot][0][1]), "=&v"(vp[mt][slot][1][0]), "=&v"(vp[mt][slot][1][1]),
              "=&v"(vp[mt][slot][2][0]), "=&v"(vp[mt][slot][2][1]), "=&v"(vp[mt][slot][3][0]), "=&v"(vp[mt][slot][3][1]),
              "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"(d[mt][0][0]), "v"(d[mt][0][1]), "v"(d[mt][1][0]), "v"(d[mt][1][1]), "v"(d[mt][2][0]), "v"(d[mt][2][1]),
              "v"(d[mt][3][0]), "v"(d[mt][3][1]));
      }
    };
    load_patch(0);
    load_b(0, 0);
    transform(0);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      const int kd = g >> 2, p4 = g & 3;
      if (g + 1 < 12) load_b(g + 1, (g + 1) & 1);
      if (p4 == 0 && kd < 2) load_patch(kd + 1);
      if (g == wg::DMA_G && wg::DMA_G < 2 + KC && nxt) dma_u(c0 + KC, u_s + (cur ^ 1) * U_CHUNK);
      if (g >= 2 && g < 2 + KC && nxt) commit_next_cl(g - 2);
      if (g == wg::DMA_G && wg::DMA_G >= 2 + KC && nxt) dma_u(c0 + KC, u_s + (cur ^ 1) * U_CHUNK);
      // (all KC*NS refill loads in ONE group would let the compiler count them exactly -- its waits before the commits
      // become vmcnt(11), (10), ... instead of (2), (1), (0) per group -- but bunching the loads costs more than the
      // coarser waits: 46.8 vs 47.4 pairs/s)
      if (g >= 2 + KC && g < 2 + 2 * KC && refill) fetch_raw_cl(c0 + 2 * KC, g - 2 - KC);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[mt][p4 * 4 + e][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                vp[mt][kd & 1][p4][e >> 1][e & 1], bq[g & 1][n][e],
                FIRST && kd == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mt][p4 * 4 + e][n], 0, 0, 0);
      if (p4 == 1 && kd < 2) transform((kd + 1) & 1);
    }
    const int dr = cur ? -4 * RAW_FLOATS : 4 * RAW_FLOATS, du = cur ? -4 * U_CHUNK : 4 * U_CHUNK;   // scalar
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) po[mt] += dr;
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) bo[p4] += du;
#pragma unroll
    for (int i = 0; i < NS; ++i) wo[i] -= dr;
  };
  {
    chunk(0, 0, std::true_type{});
    int cur = 1;
#pragma unroll 1
    for (int c0 = KC; c0 < a.Cin; c0 += KC, cur ^= 1) chunk(c0, cur, std::false_type{});
  }

  if (wg::BF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus weight DMA of the last chunk

  // ---- epilogue: Y = At M A per tile, BN scale/bias, residual, activation.  M index m = tile (row m&1, column
  // m>>1), so a lane (cout j, tiles 4*kq .. 4*kq+3) holds tile columns 2kq, 2kq+1 of both tile rows: 4 consecutive x
  // of four output rows, and the four kq lanes of a channel write 64 contiguous bytes per row ----
  // accumulator rows 4kq..4kq+3 = the 2x2 tiles of group kq = a 4 x 4 output patch at (4*(kq/GC), 4*(kq%GC))
  const int xb = x0 + 4 * (kq % GC), yq = 4 * (kq / GC);
  const bool fast = a.fast_ok && x0 + TW <= a.W && y0 + TH <= a.H;
  const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
  const bool mish = a.act == DV_ACT_MISH;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
  const int zo = z0 + wave + 4 * mt;
  if (zo >= a.D) continue;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = co0 + n * 16 + j;
    if (co >= a.Cout) continue;
    const float sc = a.ch_scale ? a.ch_scale[co] : 1.f;
    const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
    const size_t cbase = (((size_t)b * a.Cout + co) * a.D + zo) * plane + (size_t)(y0 + yq) * a.W + xb;
    f32x4 rv[4];
    if (fast && a.residual) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rv[r] = *reinterpret_cast<const f32x4*>(a.residual + cbase + (size_t)r * a.W);
    }
    // Output transform on packed fp32: the two tile rows of a tile column are elements (2h, 2h+1) of every accumulator,
    // i.e. an aligned register pair, and At M A is the same arithmetic for both -- 12 v_pk_add_f32 per (column pair, n)
    // instead of 24 + 24 scalar adds (the epilogue is pure vector-ALU work on the pipe the MFMAs of the CU's other block
    // need: skipping it altogether measured -10.7 % on the 32->32 layer, profiles/r03_wino3d_epilogue.txt).
    f32x2 yq2[2][2][2];                    // [tile column h][output row of the tile rr][output column of the tile] over (tr 0, tr 1)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x2 s0[4], s1[4];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const f32x2 m0 = {acc[mt][px][n][2 * h], acc[mt][px][n][2 * h + 1]};
        const f32x2 m1 = {acc[mt][4 + px][n][2 * h], acc[mt][4 + px][n][2 * h + 1]};
        const f32x2 m2 = {acc[mt][8 + px][n][2 * h], acc[mt][8 + px][n][2 * h + 1]};
        const f32x2 m3 = {acc[mt][12 + px][n][2 * h], acc[mt][12 + px][n][2 * h + 1]};
        s0[px] = m0 + m1 + m2;
        s1[px] = m1 - m2 - m3;
      }
      yq2[h][0][0] = s0[0] + s0[1] + s0[2];
      yq2[h][0][1] = s0[1] - s0[2] - s0[3];
      yq2[h][1][0] = s1[0] + s1[1] + s1[2];
      yq2[h][1][1] = s1[1] - s1[2] - s1[3];
    }
    // The uniform decisions (fast path, activation kind, residual) are taken ONCE per (plane, n): taken per element they
    // were a branch and four v_cndmask per stored row.
    auto rows = [&](auto act_c, auto res_c) __attribute__((always_inline)) {
      constexpr int ACTK = decltype(act_c)::value;          // 0 = identity, 1 = ReLU, 2 = whatever a.act says
      constexpr bool RELU = ACTK == 1, RES = decltype(res_c)::value;
#pragma unroll
      for (int tr = 0; tr < 2; ++tr) {     // tile row
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int yr = 2 * tr + r;
          const float y4[4] = {yq2[0][r][0][tr], yq2[0][r][1][tr], yq2[1][r][0][tr], yq2[1][r][1][tr]};
          f32x4 v = (f32x4){y4[0], y4[1], y4[2], y4[3]} * sc + bi;          // (contracted to fma; packed by the compiler)
          if (RES) v += rv[yr];
          if (RELU) {
            // max(v, v*0): NaN stays NaN as in torch.relu; two instructions per element and no VCC round trip (a compare
            // + select costs two wait states per element on gfx950)
            v = __builtin_elementwise_max(v, v * 0.f);
          } else if (ACTK == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mish ? dv_act(v[e], DV_ACT_MISH) : fmaxf(v[e], v[e] * slope);
          }
          *reinterpret_cast<f32x4*>(a.out + cbase + (size_t)yr * a.W) = v;
        }
      }
    };
    if (fast) {
      using K0 = std::integral_constant<int, 0>;
      using K1 = std::integral_constant<int, 1>;
      using K2 = std::integral_constant<int, 2>;
      if (a.act == DV_ACT_RELU) {
        if (a.residual) rows(K1{}, std::true_type{});
        else rows(K1{}, std::false_type{});
      } else if (a.act == DV_ACT_NONE) {          // (the residual layer of dres1, acv_ddim.py:262)
        if (a.residual) rows(K0{}, std::true_type{});
        else rows(K0{}, std::false_type{});
      } else {
        if (a.residual) rows(K2{}, std::true_type{});
        else rows(K2{}, std::false_type{});
      }
    } else {
#pragma unroll
      for (int tr = 0; tr < 2; ++tr) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int yr = 2 * tr + r;
          if (y0 + yq + yr >= a.H) continue;
          const size_t o = cbase + (size_t)yr * a.W;
          const float y4[4] = {yq2[0][r][0][tr], yq2[0][r][1][tr], yq2[1][r][0][tr], yq2[1][r][1][tr]};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (xb + e < a.W) {
              float u = fmaf(y4[e], sc, bi);
              if (a.residual) u += a.residual[o + e];
              a.out[o + e] = dv_act(u, a.act);
            }
        }
      }
    }
  }
}
}

// U = G g Gt per (cout, cin, kd);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void pack_wino_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout,
                                         int nchunk, int nco) {
  const size_t total = (size_t)nchunk * nco * 3 * 2 * 4 * 16;   // one thread per (chunk, cb, kd, nt, k, n)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int n = (int)(r % 16); r /= 16;
    const int k = (int)(r % 4); r /= 4;
    const int nt = (int)(r % 2); r /= 2;
    const int kd = (int)(r % 3); r /= 3;
    const int cb = (int)(r % nco);
    const int ch = (int)(r / nco);
    const int co = cb * 32 + nt * 16 + n, ci = ch * 4 + k;
    float g[3][3];
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q)
        g[p][q] = (co < Cout && ci < Cin) ? w[(((size_t)co * Cin + ci) * 3 + kd) * 9 + p * 3 + q] : 0.f;
    float gg[4][3];   // G g
    for (int q = 0; q < 3; ++q) {
      gg[0][q] = g[0][q];
      gg[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
      gg[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
      gg[3][q] = g[2][q];
    }
    float* dst = wpk + i * 16;
    for (int p = 0; p < 4; ++p) {
      dst[p * 4 + 0] = gg[p][0];
      dst[p * 4 + 1] = 0.5f * (gg[p][0] + gg[p][1] + gg[p][2]);
      dst[p * 4 + 2] = 0.5f * (gg[p][0] - gg[p][1] + gg[p][2]);
      dst[p * 4 + 3] = gg[p][2];
    }
  }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

extern "C" size_t dv_conv3d_wino_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)cdiv(Cin, 4) * cdiv(Cout, 32) * wg::U_CHUNK;
}

extern "C" int dv_conv3d_wino_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout,
                                               dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  const int nchunk = cdiv(Cin, 4), nco = cdiv(Cout, 32);
  const size_t total = (size_t)nchunk * nco * 3 * 2 * 4 * 16;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_wino_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpacked, Cin,
                     Cout, nchunk, nco);
  return dv_launch_status();
}

extern "C" int dv_conv3d_wino_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                                  const float* in_scale, const float* residual, float* out, int B, int Cin, int D,
                                  int H, int W, int Cout, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE((size_t)D * H * W * sizeof(float) <= 0x7fffffffull, DV_ERR_SHAPE);   // 31-bit byte offsets in a channel
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_LEAKY, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(dv_aligned16(wpacked), DV_ERR_ALIGN);
  WinoArgs a;
  a.in = in; a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.in_scale = in_scale;
  a.residual = residual; a.out = out;
  a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
  a.fast_ok = (W % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual));
  hipStream_t s = (hipStream_t)stream;
  auto launch = [&](auto mtw, auto shape) {
    constexpr int MTW = decltype(mtw)::value, SHAPE = decltype(shape)::value;
    using G = WG<MTW, SHAPE>;
    a.ntx = cdiv(W, G::TW); a.nty = cdiv(H, G::TH); a.ntz = cdiv(D, G::TD); a.nco = cdiv(Cout, 32);
    const long long blocks = (long long)B * a.nco * a.ntz * a.nty * a.ntx;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)DV_ERR_SHAPE;
    if (in_scale)
      hipLaunchKernelGGL((conv3d_wino_kernel<true, MTW, SHAPE>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((conv3d_wino_kernel<false, MTW, SHAPE>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    return dv_launch_status();
  };
  // tile shape of a wave's 16 Winograd tiles: the one that pads the plane least (16 x 4, 8 x 8 or 4 x 16 outputs);
  // ties go to the widest, whose rows are stored in 64-byte runs
  auto padded = [&](int tw, int th) { return (long long)cdiv(W, tw) * tw * cdiv(H, th) * th; };
  const long long p0 = padded(16, 4), p1 = padded(8, 8), p2 = padded(4, 16);
  int shape = 0;
  if (p1 < p0 && p1 <= p2) shape = 1;
  else if (p2 < p0 && p2 < p1) shape = 2;
#ifdef DV_WINO_FORCE_SHAPE
  shape = DV_WINO_FORCE_SHAPE;
#endif
  // MTW = 2 (one wave per SIMD on the 512-register file, two planes per wave) is correct but measured 4.25 vs 3.09 ms
  // on the 32->32 layer: with a single wave per SIMD the LDS / barrier latencies of every chunk are exposed.
  using one = std::integral_constant<int, 1>;
  if (shape == 1) return launch(one{}, std::integral_constant<int, 1>{});
  if (shape == 2) return launch(one{}, std::integral_constant<int, 2>{});
  return launch(one{}, std::integral_constant<int, 0>{});
}
