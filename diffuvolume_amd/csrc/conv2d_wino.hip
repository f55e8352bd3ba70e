// K11w: the 3x3, dilation-1, stride-1 2-D convolutions (IGEV's ConvGRU / motion encoder / heads, KITTI15/core/update.py;
// the residual blocks of the 2-D feature CNNs, SceneFlow/models/submodule.py:21-24,:192-215) in the Winograd
// F(2x2, 3x3) form on v_mfma_f32_16x16x4_f32 -- the 2-D sibling of conv3d_wino.hip (same transforms in registers,
// same XOR-swizzled LDS-DMA weight image, same staging spread into the MFMA groups), with the epilogue of
// conv2d.hip: per-channel scale/bias, residual, activation (incl. sigmoid / tanh), `mul` and the GRU blend, and up
// to four input tensors read as one virtual channel concatenation.
//
// Block = 4 waves = 16 x 16 outputs x 32 output channels; wave w owns rows 4w..4w+3: its MFMA tile is M = 16 Winograd
// tiles (2 tile rows x 8 tile columns), N = 16 output channels, K = 4 input channels.  A chunk is 8 input channels =
// two k-steps x 16 positions x 2 N-tiles = 64 MFMAs per wave; raw brick (18 x 18 per channel) and weights are
// double-buffered in LDS (62 KB, two blocks per CU).

#include <type_traits>

#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Timing-only ablations (results are WRONG; tools/build_variant.sh ... -DDV_W2_ABL=n): 1 no raw loads, 2 no LDS commits,
// 4 no weight DMA, 8 no block barrier, 16 no input transform, 32 no epilogue.
#ifndef DV_W2_ABL
#define DV_W2_ABL 0
#endif

#ifndef DV_W2_PIN
#define DV_W2_PIN 0
#endif

namespace w2 {
constexpr int KC = 8, NKS = 2, NT = 2, TH = 16, TW = 16;
constexpr int IY = TH + 2, IX = TW + 2;
constexpr int PRAW = IY * IX;                 // 324 raw positions per channel
constexpr int RX = 24;                        // row stride; with a channel stride = 32 mod 64 the 16 tiles x 2 channels of
constexpr int RAWP = 480;                     // a 32-lane ds_read_b64 cover the 64 banks once (see conv3d_wino.hip)
constexpr int RAW_FLOATS = KC * RAWP;
constexpr int U_CHUNK = NKS * NT * 4 * 16 * 16;   // packed floats per (chunk, co block) = the LDS image, 16 KB
constexpr int NS = (PRAW + 255) / 256;
static_assert(RAWP >= IY * RX && RAWP % 64 == 32, "bank plan of the patch reads");
static_assert((RAW_FLOATS + U_CHUNK) * 2 * 4 * 2 <= 160 * 1024, "two blocks per CU, both stages double-buffered");
}  // namespace w2

struct Wino2dArgs {
  const float* src[4];    // sources of the virtual channel concatenation ([B,c_k,H,W] each)
  int cend[4];            // cumulative channel count after each source (cend[3] == Cin)
  const float* wpk;       // [Cin/8][Coutp/32][ks 2][nt 2][k 4][n 16][pos 16]
  const float* ch_scale;
  const float* ch_bias;
  const float* residual;
  const float* mul;
  const float* blend_z;
  const float* blend_h;
  float* out;
  int B, Cin, H, W, Cout;
  int ntx, nty, nco;
  int act, fast_ok;
  int dil;                // dilation: the block works on one of the dil*dil sub-sampled images (a dilation-1 problem)
  // two convolutions that read the same input in one launch (ConvGRU's z and r gates): output channels >= gsplit (a
  // multiple of 32, 0 = off) belong to the second one, whose result / residual / mul tensors are [B, Cout - gsplit, H, W]
  int gsplit;
  const float* residual2;
  const float* mul2;
  float* out2;
  // store de-interleaved by 2 (dv_conv2d_wino_s2b_f32): out is [4B, Cout, H/2, W/2], pixel (y, x) of item b goes to item
  // 4b + 2(y & 1) + (x & 1) at (y >> 1, x >> 1) -- the layout the NEXT layer of a dilation-doubling stack reads densely
  int s2b;
  // K-split (KS kernels): `kslices` blocks share an output tile, each sums a contiguous range of the 8-channel chunks and
  // stores its raw tile sums to scratch[slice][B, Cout, H, W]; wino2d_ksplit_epilogue_kernel adds the slices in slice order
  // (deterministic) and applies the epilogue.  For the launches that leave most of the chip empty (IGEV's 1/16 scale).
  int kslices;
  float* scratch;
};

// DEEP: the low-occupancy variant for launches that do not fill the chip (IGEV's 1/8 and 1/16 scales at batch 1):
// with one block or less per CU nothing hides a chunk's memory latency, so the weight DMA runs two chunks ahead
// (ring of three LDS images) and the raw loads three (two register sets).
// SRC: how the position in the virtual concatenation advances -- 0: one source (the next plane follows; the feature CNNs,
// the KITTI12 refinement stack, the motion encoder's single-input layers), 1: several sources whose channel counts
// (all but the last) are multiples of the 8-channel chunk, so the source queue moves once per chunk (ConvGRU's
// [h | x...]), 2: any split, the queue is looked at per channel.  The scalar unit issues in the wave's instruction
// stream: the per-channel selects were ~130 of the 200 scalar instructions of a chunk body, and a wave that spends its
// issue slots on them cannot keep the matrix pipe fed when its SIMD partner stalls (round 5: one source -7 %).
template <bool DEEP, int SRC, bool KS = false>
__global__ __launch_bounds__(256, 2) void conv2d_wino_kernel(Wino2dArgs a) {
  using namespace w2;
  constexpr int NU = DEEP ? 3 : 2;
  __shared__ __attribute__((aligned(1024))) float smem[NU * U_CHUNK + 2 * RAW_FLOATS];
  float* u_s = smem;
  float* raw_s = smem + NU * U_CHUNK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;

  // A 3x3 convolution with dilation d is d*d independent dilation-1 convolutions on the images sub-sampled at
  // (ry + d*Y, rx + d*X): a block owns a 16x16 tile of ONE sub-image, and everything between the raw loads and the
  // stores is the dilation-1 kernel.  The sub-image index is the fastest tile index, so the d*d blocks that share
  // the same cache lines of input and output run side by side on one XCD.
  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  const int slice = KS ? __builtin_amdgcn_readfirstlane((int)(t % (unsigned)a.kslices)) : 0;
  if (KS) t /= (unsigned)a.kslices;
  const int dil = a.dil;
  const int sg = t % (unsigned)(dil * dil); t /= (unsigned)(dil * dil);
  const int ry = sg / dil, rx = sg - ry * dil;
  const int tc = t % a.nco; t /= a.nco;       // the output-channel slices of a tile side by side too: one HBM read of the brick
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty;
  const int b = t / a.nty;
  const int x0 = tx * TW, y0 = ty * TH, co0 = tc * 32;          // in sub-image coordinates
  const int Hs = (a.H - ry + dil - 1) / dil, Ws = (a.W - rx + dil - 1) / dil;   // size of this sub-image
  if (y0 >= Hs || x0 >= Ws) return;                              // (the tile grid is that of the largest sub-image)

  // (not zeroed: the first chunk's first k-step takes the inline constant 0 as its C operand, conv3d_wino.hip)
  f32x4 acc[16][NT];

  const size_t plane = (size_t)a.H * a.W;
  const int plane_bytes = (int)(plane * sizeof(float));      // < 2^31 (checked by the host)
  const int n_in = a.Cin;
  const int n_chunk = (n_in + w2::KC - 1) / w2::KC;
  // this block's chunks: all of them, or its slice of a K-split launch
  const int cbeg = KS ? (n_chunk * slice) / a.kslices : 0;
  const int cend = KS ? (n_chunk * (slice + 1)) / a.kslices : n_chunk;
  const int c_first = cbeg * w2::KC, c_lim = KS ? min(n_in, cend * w2::KC) : n_in;

  // ---- raw staging plan (as conv3d_wino.hip): buffer loads, zero padding and channel tail from the range check ----
  // A chunk's brick is staged in NP "pieces" (one buffer load + one LDS write per thread each).
  //  * CC (SRC 0 / 1: the 8 channels of a chunk are 8 consecutive planes of ONE tensor): the chunk is one buffer -- ONE
  //    descriptor per chunk (base = first plane, records = the planes that exist), the channel is part of the lane's
  //    offset, and the 8 x 324 positions are dealt to the 256 threads as a whole: 11 pieces.  Positions outside the image
  //    carry the offset 0x80000000 (out of range: zero), channels past the end fall outside the records (zero, no traffic).
  //  * otherwise (SRC 2: a chunk may straddle two tensors): per channel a descriptor of one plane and 2 pieces (the second
  //    27 % full): 16 pieces and ~6 scalar instructions per channel.
  constexpr bool CC = SRC != 2;
  constexpr int NP = CC ? (KC * PRAW + 255) / 256 : KC * NS;
  unsigned sob[CC ? NP : NS];
  int lro[CC ? NP : NS];
#pragma unroll
  for (int i = 0; i < (CC ? NP : NS); ++i) {
    const int e = tid + 256 * i;
    const int cl = CC ? e / PRAW : 0;
    const int r = e - cl * PRAW;
    const int yy = r / IX, xx = r - yy * IX;
    const int y = ry + dil * (y0 - 1 + yy), x = rx + dil * (x0 - 1 + xx);
    const bool in_brick = CC ? e < KC * PRAW : r < PRAW;
    const bool ok = in_brick && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    sob[i] = ok ? (unsigned)cl * (unsigned)plane_bytes + (unsigned)(y * a.W + x) * 4u : 0x80000000u;
    lro[i] = in_brick ? cl * RAWP + yy * RX + xx : IX;       // lanes past the brick write a column no patch reads
  }
  typedef float RawSet[NP];
  RawSet vinA, vinB;      // (vinB is only used by the deep variant)
  // Channels are fetched strictly in order (chunk after chunk, also past the end: a channel >= Cin gets a descriptor with
  // zero records and costs no memory traffic), so the position in the virtual concatenation is running scalar state:
  // the byte address of the next channel plane, the channels left in its source, and a queue of the sources to come.
  // Everything is selects on the scalar unit -- no kernarg loads, no 64-bit multiplies and no branch per channel (the
  // first version looked the source up per channel: two dependent s_load round trips, each closed by an
  // s_waitcnt lgkmcnt(0) that also drained the LDS queue, and ~40 scalar instructions in four basic blocks between
  // every two MFMA groups).
  auto sgpr64 = [](uint64_t v) __attribute__((always_inline)) {
    return (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
           ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
  };
  const int cw0 = a.cend[0], cw1 = a.cend[1] - a.cend[0], cw2 = a.cend[2] - a.cend[1], cw3 = a.cend[3] - a.cend[2];
  uint64_t fb = sgpr64(reinterpret_cast<uint64_t>(a.src[0] + (size_t)b * cw0 * plane));
  uint64_t nb1 = sgpr64(reinterpret_cast<uint64_t>(a.src[1] + (size_t)b * cw1 * plane));
  uint64_t nb2 = sgpr64(reinterpret_cast<uint64_t>(a.src[2] + (size_t)b * cw2 * plane));
  uint64_t nb3 = sgpr64(reinterpret_cast<uint64_t>(a.src[3] + (size_t)b * cw3 * plane));
  constexpr int NEVER = 0x7fffffff;                 // an unused source slot: its counter never reaches zero
  // (CC: the counters are compared with <= 0 once per chunk and an unused slot holds 0 channels -- once the last source is
  // used up every later descriptor has zero records)
  int left = cw0, nl1 = cw1 > 0 ? cw1 : (CC ? 0 : NEVER), nl2 = cw2 > 0 ? cw2 : (CC ? 0 : NEVER),
      nl3 = cw3 > 0 ? cw3 : (CC ? 0 : NEVER);
  int fc = 0;
  auto queue_up = [&]() __attribute__((always_inline)) {
    const bool sw = left <= 0;                      // source exhausted: the queue moves up
    fb = sw ? nb1 : fb;   left = sw ? nl1 : left;
    nb1 = sw ? nb2 : nb1; nl1 = sw ? nl2 : nl1;
    nb2 = sw ? nb3 : nb2; nl2 = sw ? nl3 : nl2;
    nl3 = sw ? (CC ? 0 : NEVER) : nl3;
  };
  // records of a chunk's descriptor = the planes that exist, clamp(left, 0, 8) * plane bytes -- on the scalar unit by hand
  // (the compiler matches the clamp to v_med3_i32 and then multiplies on the vector ALU too: three vector instructions
  // alone in the MFMA stream of every chunk)
  auto records = [&](int l) __attribute__((always_inline)) {
    int r;
    asm("s_min_i32 %0, %1, 8\n\ts_max_i32 %0, %0, 0\n\ts_mul_i32 %0, %0, %2" : "=&s"(r) : "s"(l), "s"(plane_bytes) : "scc");
    return r;
  };
  if (KS) {                                         // walk the source queue to this slice's first chunk
    for (int i = 0; i < cbeg * (CC ? 1 : KC); ++i) {
      fb += (uint64_t)(unsigned)((CC ? KC : 1) * plane_bytes);
      left -= CC ? KC : 1;
      if (!CC) ++fc;
      if (SRC != 0) queue_up();
    }
  }
  int nrec = CC ? records(left) : 0;
  // piece k of the chunk that is next in the stream (pieces are requested strictly in order, chunk after chunk)
  auto fetch_piece = [&](int k, RawSet& vin) __attribute__((always_inline)) {
    if (CC) {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(fb), 0, nrec, 0x00020000);
      vin[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)sob[k], 0, 0));
      if (k == NP - 1) {
        fb += (uint64_t)(unsigned)(KC * plane_bytes);
        left -= KC;
        if (SRC == 1) queue_up();
        nrec = records(left);
      }
    } else {
      const int i = k % NS;
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(fb), 0, fc < n_in ? plane_bytes : 0,
                                                        0x00020000);
      vin[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)sob[i], 0, 0));
      if (i == NS - 1) {
        ++fc;
        fb += (uint64_t)(unsigned)plane_bytes;
        --left;
        queue_up();
      }
    }
  };
  auto fetch_raw = [&](RawSet& vin) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NP; ++k) fetch_piece(k, vin);
  };
  auto commit_piece = [&](int k, float* rb, RawSet& vin) __attribute__((always_inline)) {
    if (CC) rb[lro[k]] = vin[k];
    else rb[(k / NS) * RAWP + lro[k % NS]] = vin[k];
  };
  // weights by LDS-DMA, source quad XOR-swizzled with the row (conv3d_wino.hip); 16 pieces of 1 KB, four per wave
  const int dma_lo = (lane >> 2) * 16 + (((lane & 3) ^ ((lane >> 4) & 3)) * 4);
  const int dma_voff = dma_lo * 4;                 // the lane part of a piece's source address: constant
  // the weight chunks are copied strictly in order too: a running scalar pointer to this wave's first piece of the next
  // chunk (a chunk past the end re-copies the last one: the staging in `chunk` is issued unconditionally)
  uint64_t ud = sgpr64(reinterpret_cast<uint64_t>(a.wpk + ((size_t)cbeg * a.nco + tc) * U_CHUNK + wave * 256));
  const uint64_t ud_step = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane(a.nco * U_CHUNK * (int)sizeof(float));
  int ud_left = cend - cbeg - 1;                         // advances left before the pointer stays on the last chunk
  auto dma_u = [&](int /*c0*/, float* ub) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = wave + 4 * q;
      // inline asm, not __builtin_amdgcn_global_load_lds: the compiler treats the builtin as an LDS store that any later
      // LDS access may alias and answers the next ds_write with s_waitcnt vmcnt(0) -- every chunk then waited for its
      // own DMA (config 5: 593 -> 584 ms per 128 GRU iterations).  Completion is covered by the manual s_waitcnt vmcnt + barrier
      // at the top of the next chunk.  Scalar piece base + constant lane offset, M0 handed back as found, one wait state
      // between the M0 write and the DMA (conv3d_wino.hip; tests/test_isa_lint.py checks the compiled stream).
      const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(ub + piece * 256);
      const uint64_t gbs = ud + (uint64_t)(q * 4 * 256 * sizeof(float));
      unsigned m0_saved;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(m0_saved) : "s"(lds_addr), "v"(dma_voff), "s"(gbs) : "memory");
    }
    ud += ud_left > 0 ? ud_step : 0;
    --ud_left;
  };

  // this lane's 4x4 patch: tile (row j&1, column j>>1) of the wave's four rows, channel ks*4 + kq;  B rows (kq, j)
  const int patch_lo = kq * RAWP + (4 * wave + 2 * (j & 1)) * RX + 2 * (j >> 1);
  int b_lo[4];
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) b_lo[p4] = (kq * 16 + j) * 16 + ((p4 ^ ((j >> 2) & 3)) * 4);

  // ---- prologue.  Chunk k >= 1 travels in register set A (shallow) or set (k & 1 ? B : A) (deep) ----
  fetch_raw(vinA);
  dma_u(0, u_s);
#pragma unroll
  for (int k = 0; k < NP; ++k) commit_piece(k, raw_s, vinA);
  if (DEEP) {
    dma_u(KC, u_s + U_CHUNK);
    fetch_raw(vinB);
    fetch_raw(vinA);
  } else {
    fetch_raw(vinA);
  }

  // one chunk.  `vin` holds the raw brick of chunk c0+KC on entry and is refilled for chunk c0+2*KC (deep: c0+3*KC);
  // ub = LDS weight image of this chunk, unxt = target of the DMA issued in this chunk (chunk c0+KC, deep: c0+2*KC)
  auto chunk = [&](int c0, int cur, const float* ub, float* unxt, RawSet& vin, auto first_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value;
    // this chunk's weights (DMA) have to be in LDS; the loads issued after that DMA may stay in flight: the raw loads
    // of the next chunk (shallow), or raw + DMA + raw of the next two (deep)
    // The chunk body has NO branch: the staging of the chunks to come is issued whether or not they exist (a channel
    // past the end is a zero-record descriptor, a weight chunk past the end re-copies the last one into a buffer nobody
    // reads), so the wait count is the same in every chunk and the whole body is one scheduling region -- with a uniform
    // branch per staging step the MFMA stream was cut into 8-instruction basic blocks.
    if (DV_W2_ABL & 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEEP ? 2 * NP + 4 : NP) : "memory");
    if (!(DV_W2_ABL & 8)) __syncthreads();
    constexpr bool nxt = true, dma_ok = true, refill = true;
    const int c_dma = c0 + (DEEP ? 2 : 1) * KC;
    float* rbn = raw_s + (cur ^ 1) * RAW_FLOATS;
    const float* rb = raw_s + cur * RAW_FLOATS + patch_lo;
    f32x2 d[4][2];
    f32x4 bq[2][NT];
    f32x2 vp[2][4][2];
    auto load_patch = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[r][0] = *reinterpret_cast<const f32x2*>(rb + ks * 4 * RAWP + r * RX);
        d[r][1] = *reinterpret_cast<const f32x2*>(rb + ks * 4 * RAWP + r * RX + 2);
      }
    };
    auto load_b = [&](int g, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
        bq[slot][n] = *reinterpret_cast<const f32x4*>(ub + ((g >> 2) * NT + n) * (4 * 256) + b_lo[g & 3]);
    };
    auto transform = [&](int slot) __attribute__((always_inline)) {   // V = Bt d B, packed fp32 (conv3d_wino.hip)
      // one asm statement = one dense burst of 16 packed adds (a vector instruction that arrives alone between two fp32
      // MFMAs makes the shared pipe drain, ~60 cycles; ~5 inside a burst); the closing s_nop covers the VALU -> MFMA
      // read hazard that gfx950 does not interlock (conv3d_wino.hip)
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
          : "=&v"(vp[slot][0][0]), "=&v"(vp[slot][0][1]), "=&v"(vp[slot][1][0]), "=&v"(vp[slot][1][1]),
            "=&v"(vp[slot][2][0]), "=&v"(vp[slot][2][1]), "=&v"(vp[slot][3][0]), "=&v"(vp[slot][3][1]),
            "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
          : "v"(d[0][0]), "v"(d[0][1]), "v"(d[1][0]), "v"(d[1][1]), "v"(d[2][0]), "v"(d[2][1]), "v"(d[3][0]), "v"(d[3][1]));
    };
    load_patch(0);
    load_b(0, 0);
    if (DV_W2_ABL & 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { vp[0][q][0] = d[q][0]; vp[0][q][1] = d[q][1]; }
    } else {
      transform(0);
    }
#pragma unroll
    for (int g = 0; g < 4 * NKS; ++g) {
      const int ks = g >> 2, p4 = g & 3;
      if (g + 1 < 4 * NKS) load_b(g + 1, (g + 1) & 1);
      if (p4 == 0 && ks + 1 < NKS) load_patch(ks + 1);
      // staging in the shadow of the MFMAs: 2 channels of LDS commit per group, then the weight DMA (after the commits:
      // the waits the compiler puts in front of them are vmcnt counts that would otherwise take the DMA along), then
      // 2 channels of refill per group
      if (g == 4 && dma_ok && !(DV_W2_ABL & 4)) dma_u(c_dma, unxt);
      if (g < 4 && nxt && !(DV_W2_ABL & 2)) {
#pragma unroll
        for (int k = NP * g / 4; k < NP * (g + 1) / 4; ++k) commit_piece(k, rbn, vin);
      }
      if (g >= 4 && refill && !(DV_W2_ABL & 1)) {
#pragma unroll
        for (int k = NP * (g - 4) / 4; k < NP * (g - 3) / 4; ++k) fetch_piece(k, vin);
      }
      if (DV_W2_PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[p4 * 4 + e][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
              vp[ks & 1][p4][e >> 1][e & 1], bq[g & 1][n][e],
              FIRST && ks == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[p4 * 4 + e][n], 0, 0, 0);
      if (DV_W2_PIN) __builtin_amdgcn_sched_barrier(0);
      if (p4 == 1 && ks + 1 < NKS) {
        if (DV_W2_ABL & 16) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { vp[(ks + 1) & 1][q][0] = d[q][0]; vp[(ks + 1) & 1][q][1] = d[q][1]; }
        } else {
          transform((ks + 1) & 1);
        }
      }
    }
  };
  if (DEEP) {
    // weight ring slot of chunk k = k % 3; raw LDS buffer k & 1; register set of chunk k+1 alternates B, A, B, ...
    int iu = 0;
    auto pair = [&](int c0, auto first_c) __attribute__((always_inline)) {
      const int iu1 = iu == 2 ? 0 : iu + 1, iu2 = iu1 == 2 ? 0 : iu1 + 1;
      chunk(c0, 0, u_s + iu * U_CHUNK, u_s + iu2 * U_CHUNK, vinB, first_c);
      if (c0 + KC < c_lim) chunk(c0 + KC, 1, u_s + iu1 * U_CHUNK, u_s + iu * U_CHUNK, vinA, std::false_type{});
      iu = iu2;
    };
    pair(c_first, std::true_type{});
#pragma unroll 1
    for (int c0 = c_first + 2 * KC; c0 < c_lim; c0 += 2 * KC) pair(c0, std::false_type{});
  } else {
    // two chunks per trip: the buffer index is a compile-time constant in each copy of the body, so every LDS address is
    // a loop-invariant register + an immediate.  With a run-time index the body carried 17 address instructions on the
    // vector ALU, alone between MFMAs -- the pipe the fp32 MFMAs issue on (round 5: -3 % per launch).
    auto pair = [&](int c0, auto first_c) __attribute__((always_inline)) {
      chunk(c0, 0, u_s, u_s + U_CHUNK, vinA, first_c);
      if (c0 + KC < c_lim) chunk(c0 + KC, 1, u_s + U_CHUNK, u_s, vinA, std::false_type{});
    };
    pair(c_first, std::true_type{});
#pragma unroll 1
    for (int c0 = c_first + 2 * KC; c0 < c_lim; c0 += 2 * KC) pair(c0, std::false_type{});
  }

  // the last chunks' surplus weight DMA must have landed before this block's LDS can be given to another one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: Y = At M A per tile; a lane (cout j, tiles 4kq..4kq+3) holds 4 consecutive x of four rows ----
  const int yb = y0 + 4 * wave, xb = x0 + 4 * kq;
  if (yb >= Hs) return;
  if ((DV_W2_ABL & 32) && a.Cin > 0) {          // keep the accumulators alive: one store that depends on all of them
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) t += acc[i][n][0] + acc[i][n][1] + acc[i][n][2] + acc[i][n][3];
    if (t == 123.456f) a.out[0] = t;
    return;
  }
  const bool fast = a.fast_ok && dil == 1 && x0 + TW <= a.W && yb + 4 <= a.H;
  const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
  const bool gen = a.act == DV_ACT_MISH || a.act == DV_ACT_SIGMOID || a.act == DV_ACT_TANH;
  // a block's 32 output channels lie in one group (gsplit % 32 == 0): the group's tensors are chosen per block
  const bool g2 = a.gsplit > 0 && co0 >= a.gsplit;
  const int cog0 = g2 ? a.gsplit : 0, coutg = g2 ? a.Cout - a.gsplit : (a.gsplit > 0 ? a.gsplit : a.Cout);
  float* const outp = g2 ? a.out2 : a.out;
  const float* const resp = g2 ? a.residual2 : a.residual;
  const float* const mulp = g2 ? a.mul2 : a.mul;
  typedef f32x2 TileSums[2][2][2];       // [tile column h][row of the tile][column of the tile] over (tile row 0, 1)
  // At M A on packed fp32 over the two tile rows (elements 2h, 2h+1 of every accumulator are an aligned register pair;
  // conv3d_wino.hip)
  auto tile_sums = [&](int n, TileSums& yq2) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x2 s0[4], s1[4];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const f32x2 m0 = {acc[px][n][2 * h], acc[px][n][2 * h + 1]};
        const f32x2 m1 = {acc[4 + px][n][2 * h], acc[4 + px][n][2 * h + 1]};
        const f32x2 m2 = {acc[8 + px][n][2 * h], acc[8 + px][n][2 * h + 1]};
        const f32x2 m3 = {acc[12 + px][n][2 * h], acc[12 + px][n][2 * h + 1]};
        s0[px] = m0 + m1 + m2;
        s1[px] = m1 - m2 - m3;
      }
      yq2[h][0][0] = s0[0] + s0[1] + s0[2];
      yq2[h][0][1] = s0[1] - s0[2] - s0[3];
      yq2[h][1][0] = s1[0] + s1[1] + s1[2];
      yq2[h][1][1] = s1[1] - s1[2] - s1[3];
    }
  };
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = co0 + n * 16 + j;
    if (co >= a.Cout) continue;
    const float sc = a.ch_scale ? a.ch_scale[co] : 1.f;
    const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
    const size_t cbase = (((size_t)b * coutg + (co - cog0)) * a.H + (ry + dil * yb)) * a.W + (rx + dil * xb);
    // the rows: the plain ReLU / identity layers (feature CNNs, refinement stack, motion encoder) take a path whose uniform
    // decisions are made once, everything else the general one.  An epilogue instruction is not hidden by the other
    // block's MFMAs -- they share the issue pipe.
    TileSums yq2;
    tile_sums(n, yq2);
    if (KS) {      // raw tile sums of this slice -> scratch[slice][B, Cout, H, W]; the epilogue runs in the reduction kernel
      float* sp = a.scratch + (size_t)slice * ((size_t)a.B * a.Cout * plane) +
                  (((size_t)b * a.Cout + co) * a.H + (ry + dil * yb)) * a.W + (rx + dil * xb);
#pragma unroll
      for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int yr = 2 * tr + r;
          const float y4[4] = {yq2[0][r][0][tr], yq2[0][r][1][tr], yq2[1][r][0][tr], yq2[1][r][1][tr]};
          if (yb + yr < Hs) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (xb + e < Ws) sp[(size_t)(dil * yr) * a.W + (size_t)(dil * e)] = y4[e];
          }
        }
      continue;
    }
    auto plain_rows = [&](auto relu_c, auto res_c) __attribute__((always_inline)) {
      constexpr bool RELU = decltype(relu_c)::value, RES = decltype(res_c)::value;
#pragma unroll
      for (int tr = 0; tr < 2; ++tr) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const size_t o = cbase + (size_t)(2 * tr + r) * a.W;
          f32x4 v = (f32x4){yq2[0][r][0][tr], yq2[0][r][1][tr], yq2[1][r][0][tr], yq2[1][r][1][tr]} * sc + bi;
          if (RES) v += *reinterpret_cast<const f32x4*>(resp + o);
          if (RELU) v = __builtin_elementwise_max(v, v * 0.f);      // NaN stays NaN, as torch.relu
          *reinterpret_cast<f32x4*>(outp + o) = v;
        }
      }
    };
    // the same for a dilated layer (a tile of one sub-sampled image: outputs `dil` apart, scalar stores) when the tile
    // lies inside its sub-image -- the general path below tests every element
    auto plain_rows_dil = [&](auto relu_c, auto res_c) __attribute__((always_inline)) {
      constexpr bool RELU = decltype(relu_c)::value, RES = decltype(res_c)::value;
#pragma unroll
      for (int tr = 0; tr < 2; ++tr) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const size_t o = cbase + (size_t)(dil * (2 * tr + r)) * a.W;
          f32x4 v = (f32x4){yq2[0][r][0][tr], yq2[0][r][1][tr], yq2[1][r][0][tr], yq2[1][r][1][tr]} * sc + bi;
          if (RES) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += resp[o + (size_t)(dil * e)];
          }
          if (RELU) v = __builtin_elementwise_max(v, v * 0.f);
#pragma unroll
          for (int e = 0; e < 4; ++e) outp[o + (size_t)(dil * e)] = v[e];
        }
      }
    };
    const bool plain_act = !mulp && !a.blend_z && (a.act == DV_ACT_RELU || a.act == DV_ACT_NONE);
    if (plain_act && dil > 1 && x0 + TW <= Ws && yb + 4 <= Hs) {
      if (a.act == DV_ACT_RELU) {
        if (resp) plain_rows_dil(std::true_type{}, std::true_type{});
        else plain_rows_dil(std::true_type{}, std::false_type{});
      } else {
        if (resp) plain_rows_dil(std::false_type{}, std::true_type{});
        else plain_rows_dil(std::false_type{}, std::false_type{});
      }
      continue;
    }
    const bool plain = fast && plain_act && !a.s2b;
    if (plain) {
      if (a.act == DV_ACT_RELU) {
        if (resp) plain_rows(std::true_type{}, std::true_type{});
        else plain_rows(std::true_type{}, std::false_type{});
      } else {
        if (resp) plain_rows(std::false_type{}, std::true_type{});
        else plain_rows(std::false_type{}, std::false_type{});
      }
      continue;
    }
    // the gate epilogues of ConvGRU (sigmoid [* h], tanh + blend) on full aligned tiles: the non-linearity is a compile-time
    // choice per block (hardware exponential / reciprocal, dv_common.h), the operand decisions are made per row.  Measured
    // and NOT kept: every operand row of a tile (or of both tiles) requested in front of the output transform -- slower
    // (0.895 / 0.493 ms against 0.871 / 0.475 for the z|r and q launches of gru04), the rows as they come are the best order.
    auto gate_rows = [&](auto act_c) __attribute__((always_inline)) {
      constexpr int ACT = decltype(act_c)::value;
#pragma unroll
      for (int tr = 0; tr < 2; ++tr) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const size_t o = cbase + (size_t)(2 * tr + r) * a.W;
          f32x4 v = (f32x4){yq2[0][r][0][tr], yq2[0][r][1][tr], yq2[1][r][0][tr], yq2[1][r][1][tr]} * sc + bi;
          if (resp) v += *reinterpret_cast<const f32x4*>(resp + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ACT == DV_ACT_SIGMOID ? dv_sigmoid(v[e]) : dv_tanh(v[e]);
          if (mulp) v *= *reinterpret_cast<const f32x4*>(mulp + o);
          if (a.blend_z) {
            const f32x4 z = *reinterpret_cast<const f32x4*>(a.blend_z + o);
            const f32x4 h = *reinterpret_cast<const f32x4*>(a.blend_h + o);
            v = h + z * (v - h);
          }
          *reinterpret_cast<f32x4*>(outp + o) = v;
        }
      }
    };
    if (fast && !a.s2b && (a.act == DV_ACT_SIGMOID || a.act == DV_ACT_TANH)) {
      if (a.act == DV_ACT_SIGMOID) gate_rows(std::integral_constant<int, DV_ACT_SIGMOID>{});
      else gate_rows(std::integral_constant<int, DV_ACT_TANH>{});
      continue;
    }
#pragma unroll
    for (int tr = 0; tr < 2; ++tr) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int yr = 2 * tr + r;
        const size_t o = cbase + (size_t)(dil * yr) * a.W;
        const float y4[4] = {yq2[0][r][0][tr], yq2[0][r][1][tr], yq2[1][r][0][tr], yq2[1][r][1][tr]};
        if (fast) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(y4[e], sc, bi);
          if (resp) v += *reinterpret_cast<const f32x4*>(resp + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gen ? dv_act(v[e], a.act) : fmaxf(v[e], v[e] * slope);
          if (mulp) v *= *reinterpret_cast<const f32x4*>(mulp + o);
          if (a.blend_z) {
            const f32x4 z = *reinterpret_cast<const f32x4*>(a.blend_z + o);
            const f32x4 h = *reinterpret_cast<const f32x4*>(a.blend_h + o);
            v = h + z * (v - h);
          }
          if (a.s2b) {     // (no residual / mul / blend on this path: the host checks) x = xb + e with xb a multiple of 4
            const int y = yb + yr;
            const size_t pl = (size_t)(a.H >> 1) * (a.W >> 1);
            const size_t ob = (((size_t)b * 4 + 2 * (y & 1)) * coutg + (co - cog0)) * pl + (size_t)(y >> 1) * (a.W >> 1) + (xb >> 1);
            *reinterpret_cast<f32x2*>(outp + ob) = (f32x2){v[0], v[2]};
            *reinterpret_cast<f32x2*>(outp + ob + (size_t)coutg * pl) = (f32x2){v[1], v[3]};
          } else {
            *reinterpret_cast<f32x4*>(outp + o) = v;
          }
        } else if (yb + yr < Hs) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (xb + e < Ws) {
              const size_t oe = o + (size_t)(dil * e);
              float u = fmaf(y4[e], sc, bi);
              if (resp) u += resp[oe];
              u = dv_act(u, a.act);
              if (mulp) u *= mulp[oe];
              if (a.blend_z) u = a.blend_h[oe] + a.blend_z[oe] * (u - a.blend_h[oe]);
              if (a.s2b) {
                const int y = yb + yr, x = xb + e;
                const size_t pl = (size_t)(a.H >> 1) * (a.W >> 1);
                outp[(((size_t)b * 4 + 2 * (y & 1) + (x & 1)) * coutg + (co - cog0)) * pl + (size_t)(y >> 1) * (a.W >> 1) + (x >> 1)] = u;
              } else {
                outp[oe] = u;
              }
            }
        }
      }
    }
  }
}

// U = G g Gt per (cout, cin);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ void pack_wino2d_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout,
                                           int nchunk, int nco) {
  const size_t total = (size_t)nchunk * nco * 2 * 2 * 4 * 16;   // one thread per (chunk, cb, ks, nt, k, n)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int n = (int)(r % 16); r /= 16;
    const int k = (int)(r % 4); r /= 4;
    const int nt = (int)(r % 2); r /= 2;
    const int ks = (int)(r % 2); r /= 2;
    const int cb = (int)(r % nco);
    const int ch = (int)(r / nco);
    const int co = cb * 32 + nt * 16 + n, ci = ch * 8 + ks * 4 + k;
    float g[3][3];
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) g[p][q] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + p * 3 + q] : 0.f;
    float gg[4][3];
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

inline int cdiv2(int a, int b) { return (a + b - 1) / b; }

// sum of the K-split slices (fixed order) + the epilogue of conv2d_wino_kernel, incl. the two-group form of the gate pair
__global__ __launch_bounds__(256) void wino2d_ksplit_epilogue_kernel(Wino2dArgs a, size_t total, size_t oplane) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float v = a.scratch[i];
  for (int sl = 1; sl < a.kslices; ++sl) v += a.scratch[(size_t)sl * total + i];
  const size_t pc = i / oplane, p = i - pc * oplane;
  const int co = (int)(pc % (size_t)a.Cout);
  const size_t b = pc / (size_t)a.Cout;
  const bool g2 = a.gsplit > 0 && co >= a.gsplit;
  const int cog0 = g2 ? a.gsplit : 0, coutg = g2 ? a.Cout - a.gsplit : (a.gsplit > 0 ? a.gsplit : a.Cout);
  const size_t o = (b * coutg + (co - cog0)) * oplane + p;
  const float* resp = g2 ? a.residual2 : a.residual;
  const float* mulp = g2 ? a.mul2 : a.mul;
  v = fmaf(v, a.ch_scale ? a.ch_scale[co] : 1.f, a.ch_bias ? a.ch_bias[co] : 0.f);
  if (resp) v += resp[o];
  v = dv_act(v, a.act);
  if (mulp) v *= mulp[o];
  if (a.blend_z) v = a.blend_h[o] + a.blend_z[o] * (v - a.blend_h[o]);
  (g2 ? a.out2 : a.out)[o] = v;
}

// (measured at batch 4, 24 x 78: gate pair 104 us unsplit, 92 / 86 / 86 us at up to 4 / 2 / 3 slices; candidate 75 -> 58 / 58 / 48)
// Slices of a Winograd launch: 1 unless ONE batch item has at most 96 blocks (the rule may only look at one item: a shard of
// a batch has to sum in the batch's order) and the slices keep >= 8 chunks each.  At batch 4 this is IGEV's 1/16 scale
// (80 / 40 blocks per item): 320 blocks of 32 chunks leave 3/4 of the SIMDs with one wave or none.
inline int wino2d_kslices(int Cin, int H, int W, int Cout, int dilation) {
  const long long blocks = (long long)cdiv2(cdiv2(H, dilation), w2::TH) * cdiv2(cdiv2(W, dilation), w2::TW) * cdiv2(Cout, 32) *
                           dilation * dilation;
  const int nchunk = cdiv2(Cin, w2::KC);
  if (blocks > 96 || nchunk < 16) return 1;
#ifndef DV_W2_KS_MAX
#define DV_W2_KS_MAX 3
#endif
  const int ks = nchunk / 8;
  return ks > DV_W2_KS_MAX ? DV_W2_KS_MAX : ks;
}

}  // namespace

extern "C" size_t dv_conv2d_wino_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)cdiv2(Cin, 8) * cdiv2(Cout, 32) * w2::U_CHUNK;
}

extern "C" int dv_conv2d_wino_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout,
                                               dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  const int nchunk = cdiv2(Cin, 8), nco = cdiv2(Cout, 32);
  const size_t total = (size_t)nchunk * nco * 2 * 2 * 4 * 16;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_wino2d_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpacked, Cin,
                     Cout, nchunk, nco);
  return dv_launch_status();
}

namespace {
int wino2d_launch(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                  const float* ch_scale, const float* ch_bias, const float* residual, const float* mul,
                  const float* blend_z, const float* blend_h, float* out, int B, int H, int W, int Cout, int dilation,
                  int act, int gsplit, const float* residual2, const float* mul2, float* out2, dv_stream_t stream,
                  int s2b = 0, float* scratch = nullptr, int kslices = 1);
}

extern "C" int dv_conv2d_wino_auto_kslices(int Cin, int H, int W, int Cout, int dilation) {
  if (Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || dilation < 1) return 1;
  return wino2d_kslices(Cin, H, W, Cout, dilation);
}

// The K-split forms of dv_conv2d_wino_dil_cat_f32 / dv_conv2d_wino_cat_pair_f32: `kslices` must be
// dv_conv2d_wino_auto_kslices(Cin, H, W, Cout [both convolutions], dilation) > 1, scratch holds kslices * B * Cout * H * W floats.
extern "C" int dv_conv2d_wino_cat_ksplit_f32(const float* const* inputs, const int* channels, int n_inputs,
                                             const float* wpacked, const float* ch_scale, const float* ch_bias,
                                             const float* residual, const float* mul, const float* blend_z,
                                             const float* blend_h, float* out, float* scratch, int kslices, int B, int H,
                                             int W, int Cout, int dilation, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(scratch);
  return wino2d_launch(inputs, channels, n_inputs, wpacked, ch_scale, ch_bias, residual, mul, blend_z, blend_h, out, B, H,
                       W, Cout, dilation, act, 0, nullptr, nullptr, nullptr, stream, 0, scratch, kslices);
}

extern "C" int dv_conv2d_wino_cat_pair_ksplit_f32(const float* const* inputs, const int* channels, int n_inputs,
                                                  const float* wpacked, const float* ch_scale, const float* ch_bias,
                                                  const float* residual1, const float* mul1, float* out1,
                                                  const float* residual2, const float* mul2, float* out2, float* scratch,
                                                  int kslices, int B, int H, int W, int Cout1, int Cout2, int act,
                                                  dv_stream_t stream) {
  DV_REQUIRE_PTR(out2);
  DV_REQUIRE_PTR(scratch);
  DV_REQUIRE(Cout1 > 0 && Cout2 > 0 && Cout1 % 32 == 0, DV_ERR_SHAPE);
  return wino2d_launch(inputs, channels, n_inputs, wpacked, ch_scale, ch_bias, residual1, mul1, nullptr, nullptr, out1, B,
                       H, W, Cout1 + Cout2, 1, act, Cout1, residual2, mul2, out2, stream, 0, scratch, kslices);
}

extern "C" int dv_conv2d_wino_dil_cat_f32(const float* const* inputs, const int* channels, int n_inputs,
                                          const float* wpacked, const float* ch_scale, const float* ch_bias,
                                          const float* residual, const float* mul, const float* blend_z,
                                          const float* blend_h, float* out, int B, int H, int W, int Cout,
                                          int dilation, int act, dv_stream_t stream) {
  return wino2d_launch(inputs, channels, n_inputs, wpacked, ch_scale, ch_bias, residual, mul, blend_z, blend_h, out, B, H,
                       W, Cout, dilation, act, 0, nullptr, nullptr, nullptr, stream);
}

extern "C" int dv_conv2d_wino_cat_pair_f32(const float* const* inputs, const int* channels, int n_inputs,
                                           const float* wpacked, const float* ch_scale, const float* ch_bias,
                                           const float* residual1, const float* mul1, float* out1,
                                           const float* residual2, const float* mul2, float* out2, int B, int H, int W,
                                           int Cout1, int Cout2, int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(out2);
  DV_REQUIRE(Cout1 > 0 && Cout2 > 0 && Cout1 % 32 == 0, DV_ERR_SHAPE);
  DV_REQUIRE((!residual2 || dv_aligned16(residual2)) && (!mul2 || dv_aligned16(mul2)) && dv_aligned16(out2), DV_ERR_ALIGN);
  return wino2d_launch(inputs, channels, n_inputs, wpacked, ch_scale, ch_bias, residual1, mul1, nullptr, nullptr, out1, B,
                       H, W, Cout1 + Cout2, 1, act, Cout1, residual2, mul2, out2, stream);
}

namespace {
int wino2d_launch(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                  const float* ch_scale, const float* ch_bias, const float* residual, const float* mul,
                  const float* blend_z, const float* blend_h, float* out, int B, int H, int W, int Cout, int dilation,
                  int act, int gsplit, const float* residual2, const float* mul2, float* out2, dv_stream_t stream,
                  int s2b, float* scratch, int kslices) {
  DV_REQUIRE_PTR(inputs);
  DV_REQUIRE(dilation >= 1 && dilation <= 16, DV_ERR_UNSUPPORTED);
  DV_REQUIRE_PTR(channels);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(n_inputs >= 1 && n_inputs <= 4, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_TANH, DV_ERR_UNSUPPORTED);
  DV_REQUIRE((blend_z == nullptr) == (blend_h == nullptr), DV_ERR_NULL);
  DV_REQUIRE(dv_aligned16(wpacked), DV_ERR_ALIGN);
  DV_REQUIRE((size_t)H * W * sizeof(float) <= 0x7fffffffull, DV_ERR_SHAPE);   // 31-bit byte offsets in a channel
  Wino2dArgs a;
  int cin = 0;
  for (int i = 0; i < 4; ++i) {
    if (i < n_inputs) {
      DV_REQUIRE_PTR(inputs[i]);
      DV_REQUIRE(channels[i] > 0, DV_ERR_SHAPE);
      cin += channels[i];
      a.src[i] = inputs[i];
    } else {
      a.src[i] = inputs[0];
    }
    a.cend[i] = cin;
  }
  a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.residual = residual; a.mul = mul;
  a.blend_z = blend_z; a.blend_h = blend_h; a.out = out;
  a.B = B; a.Cin = cin; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
  a.fast_ok = (W % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual)) &&
              (!mul || dv_aligned16(mul)) && (!blend_z || (dv_aligned16(blend_z) && dv_aligned16(blend_h)));
  a.dil = dilation;
  a.gsplit = gsplit; a.residual2 = residual2; a.mul2 = mul2; a.out2 = out2;
  a.s2b = s2b;
  if (s2b) {
    DV_REQUIRE(dilation == 1 && gsplit == 0 && !residual && !mul && !blend_z && H % 2 == 0 && W % 2 == 0, DV_ERR_UNSUPPORTED);
    a.fast_ok = (W % 4 == 0) && dv_aligned16(out);
  }
  a.ntx = cdiv2(cdiv2(W, dilation), w2::TW); a.nty = cdiv2(cdiv2(H, dilation), w2::TH); a.nco = cdiv2(Cout, 32);
  a.kslices = kslices; a.scratch = scratch;
  if (kslices != 1) {
    DV_REQUIRE(scratch != nullptr && !s2b && kslices == wino2d_kslices(cin, H, W, Cout, dilation), DV_ERR_UNSUPPORTED);
  }
  const long long blocks = (long long)B * a.nco * a.nty * a.ntx * dilation * dilation * kslices;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return DV_ERR_SHAPE;
  // launches that leave most of the chip empty run the deep-prefetch variant
#ifndef DV_W2_DEEP_BELOW
#define DV_W2_DEEP_BELOW 512
#endif
  const bool deep = blocks < DV_W2_DEEP_BELOW;
  int src_mode = n_inputs == 1 ? 0 : 1;
  for (int i = 0; i + 1 < n_inputs; ++i)
    if (channels[i] % w2::KC) src_mode = 2;
  if ((size_t)H * W * sizeof(float) * w2::KC > 0x7fffffffull) src_mode = 2;     // a chunk as ONE buffer: 31-bit lane offsets
  void (*kern)(Wino2dArgs) =
      deep ? (src_mode == 0 ? conv2d_wino_kernel<true, 0> : src_mode == 1 ? conv2d_wino_kernel<true, 1> : conv2d_wino_kernel<true, 2>)
           : (src_mode == 0 ? conv2d_wino_kernel<false, 0> : src_mode == 1 ? conv2d_wino_kernel<false, 1> : conv2d_wino_kernel<false, 2>);
  if (kslices > 1) {
    void (*kk)(Wino2dArgs) = src_mode == 0 ? conv2d_wino_kernel<true, 0, true>
                             : src_mode == 1 ? conv2d_wino_kernel<true, 1, true> : conv2d_wino_kernel<true, 2, true>;
    hipLaunchKernelGGL(kk, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    const int rc = dv_launch_status();
    if (rc != DV_OK) return rc;
    const size_t oplane = (size_t)H * W, total = (size_t)B * Cout * oplane;
    hipLaunchKernelGGL(wino2d_ksplit_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a, total, oplane);
    return dv_launch_status();
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return dv_launch_status();
}
}  // namespace

extern "C" int dv_conv2d_wino_cat_f32(const float* const* inputs, const int* channels, int n_inputs,
                                      const float* wpacked, const float* ch_scale, const float* ch_bias,
                                      const float* residual, const float* mul, const float* blend_z,
                                      const float* blend_h, float* out, int B, int H, int W, int Cout, int act,
                                      dv_stream_t stream) {
  return dv_conv2d_wino_dil_cat_f32(inputs, channels, n_inputs, wpacked, ch_scale, ch_bias, residual, mul, blend_z,
                                    blend_h, out, B, H, W, Cout, 1, act, stream);
}

extern "C" int dv_conv2d_wino_s2b_f32(const float* const* inputs, const int* channels, int n_inputs, const float* wpacked,
                                      const float* ch_scale, const float* ch_bias, float* out, int B, int H, int W, int Cout,
                                      int act, dv_stream_t stream) {
  return wino2d_launch(inputs, channels, n_inputs, wpacked, ch_scale, ch_bias, nullptr, nullptr, nullptr, nullptr, out, B, H,
                       W, Cout, 1, act, 0, nullptr, nullptr, nullptr, stream, 1);
}
