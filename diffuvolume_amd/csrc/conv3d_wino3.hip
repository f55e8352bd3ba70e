// K4w3: the 3x3x3 stride-1 aggregation convolution (convbn_3d, SceneFlow/models/submodule.py:94-97; the dres / hourglass /
// classifier layers of acv_ddim.py:60-70, :200-222) with ALL THREE taps done by the Winograd minimal-filtering transform
// F(2x2x2, 3x3x3):
//   V = (Bt x Bt x Bt) d  of every 4x4x4 input patch (stride 2),  U = (G x G x G) g  (packed once),
//   M[p] += V[p] * U[p] over the input channels, p = 64 transform positions,  Y = (At x At x At) M  (2x2x2 outputs)
// -- 8 multiplies per output and input channel instead of 12 for conv3d_wino.hip (F(2x2,3x3) in-plane, depth taps direct)
// and 27 for the direct sum: 1.5x fewer MFMA flops than the kernel it replaces, still on the exact-fp32 instruction
// v_mfma_f32_16x16x4_f32.  The transforms only add and subtract (the 1/2 of G is folded into the packed weights); measured
// fp32 error against float64 (tools/probes/wino_f222_numerics.py, 32 channels): 5.3e-8 rms of the output scale, the
// in-plane form 6.1e-8, the direct fp32 sum 5.0e-8.
//
// 64 positions x (16 tiles x 16 cout) x 4 registers would be 256 accumulators per wave, so the DEPTH position is the wave:
// block = 4 waves = a 2(z) x 16-tile output brick x 32 output channels; wave a owns depth position a of the transform --
// the input planes (zA, zB, sign) = (0,2,-), (1,2,+), (2,1,-), (1,3,-) of the brick's four -- and all 16 in-plane
// positions: M = the 16 in-plane tiles, N = 16 output channels (two N-tiles), K = 4 input channels, 128 accumulators.
// After the channel loop every wave applies the in-plane output transform to its own sums (16 positions -> 2x2 per tile),
// the four depth positions meet in LDS (32 floats per lane and wave) and wave w = (plane, N-tile) adds the three that
// make its output plane (Z0 + Z1 + Z2 / Z1 - Z2 - Z3), applies BN / residual / activation and stores.
//
// Pipeline (one chunk = 4 input channels = one k-step = 4 groups of 8 MFMAs per wave; two blocks per CU):
//  * weights never touch LDS: wave a reads only depth position a of the packed image, a B fragment (the 4 positions of one
//    transform row x 16 cout x 4 channels = 1 KB per wave) is one coalesced 16-byte buffer load per lane from the packed
//    image (L2 resident), issued three MFMA groups ahead into a ring of four register slots (conv3d_s2pp.hip).  (A first
//    version copied them by LDS-DMA into a wave-private ring and the raw brick by dword LDS-DMA: 18 DMA instructions per
//    chunk and wave at ~60 cycles of issue each -- 3.05 ms on the 32 -> 32 layer against 2.13 without the raw copies and
//    2.55 without the weight copies; profiles/r06_wino3_experiments.txt.)
//  * the raw brick (4 planes x (TH+2) x (TW+2) x 4 channels) goes global -> registers -> LDS: 16-byte loads of the
//    aligned quads x0-4 .. (whole quads are inside or outside the volume: the range check is the zero padding), three
//    chunks ahead into two register sets, committed one float to the right (b32 + b64 + b32) so that a patch's column
//    pairs stay 8-byte aligned; double-buffered in LDS with ONE block barrier per chunk, placed mid-chunk: transform rows
//    0,1 of this chunk's patch and the commits of the next brick come before it, transform rows 2,3 of the next chunk's
//    patch after it, so no MFMA group waits for a transform (the MFMA groups run in the row order 2, 3, 0, 1 and the patch
//    registers are updated in place).

#include <atomic>
#include <type_traits>

#include "dv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int w3_round_up_mod64_32(int v) {   // smallest v' >= v with v' % 64 == 32
  const int r = v % 64;
  return r <= 32 ? v + (32 - r) : v + (64 - r) + 32;
}

// SHAPE = how the 16 in-plane Winograd tiles of a wave's M index lie in the plane (conv3d_wino.hip): 0 -> 2 tile rows x 8
// tile columns (4 x 16 outputs), 1 -> 4 x 4 (8 x 8), 2 -> 8 x 2 (16 x 4); the host picks the one that pads the plane least.
template <int SHAPE_>
struct W3G {
  static constexpr int SHAPE = SHAPE_;
  static constexpr int TR = SHAPE == 0 ? 2 : (SHAPE == 1 ? 4 : 8), TC = 16 / TR;
  static constexpr int KC = 4, NT = 2, TD = 2, TH = 2 * TR, TW = 2 * TC;
  static constexpr int IZ = TD + 2, IY = TH + 2, IX = TW + 2;
  static constexpr int PRAW = IZ * IY * IX;          // raw positions per channel (432 / 400 / 432)
  // row stride: room for the staged quads (columns 0 .. 4 QR - 4) and the bank plan of the patch reads (conv3d_wino.hip: a
  // 32-lane half of a ds_read_b64 = 16 tiles x 2 channels; channel stride = 32 mod 64; tile columns 2 floats apart, tile rows
  // 2 RX: 48 -> 2 x 8 tiles on 32 banks; 40 -> rows at 0, 40, 16, 56; 20 -> rows at 0, 20, 40, 60, 16, 36, 56, 12)
  static constexpr int RX = SHAPE == 0 ? 24 : (SHAPE == 1 ? 20 : 10);
  static constexpr int RAWP = w3_round_up_mod64_32(IZ * IY * RX);
  static constexpr int RAW_FLOATS = KC * RAWP + 8;   // one raw buffer (+ a guard quad in front: see the commits)
  // staging: the brick's rows as aligned 16-byte quads from x0 - 4: QR quads per row cover x0 - 1 .. x0 + TW
  static constexpr int QR = (IX + 3 + 3) / 4;        // quads x0-4+4q, q < QR  (6 / 4 / 3)
  static constexpr int NQ = KC * IZ * IY * QR;       // quads per chunk (576 / 640 / 864)
  static constexpr int NS = (NQ + 255) / 256;        // 16-byte loads per thread and chunk
  static_assert(RX >= IX && RX % 2 == 0 && 4 * (QR - 1) <= RX - 1, "row stride");
  static_assert(RAWP % 64 == 32, "bank plan of the patch reads");
};

namespace w3 {
constexpr int PIECE = 256;                 // floats of a B fragment of a wave: [k 4][n 16][4 positions of a transform row]
constexpr int U_PART = 8 * PIECE;          // one depth position of a (chunk, co block): [group 4][nt 2] pieces
constexpr int U_CHUNK = 4 * U_PART;        // packed floats per (chunk, co block): 32 KB
constexpr int PD = 3;                      // MFMA groups a B fragment is fetched ahead (ring of 4 = one chunk)
__host__ __device__ constexpr int row_of_group(int s) { return (s + 2) & 3; }   // MFMA groups run rows 2, 3, 0, 1
}  // namespace w3

struct W3Args {
  const float* in;
  const float* wpk;      // [Cin/4][Coutp/32][a 4][group 4][nt 2][k 4][n 16][col 4]
  const float* ch_scale;
  const float* ch_bias;
  const float* residual;
  float* out;
  int B, Cin, D, H, W, Cout;
  int ntx, nty, ntz, nco;
  int tc_slow;           // tile order: output-channel block slowest (see the kernel)
  int act;
  int fast_ok;           // W % 4 == 0, 16-byte aligned pointers
};

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void conv3d_wino3_kernel(W3Args a) {
  using G = W3G<SHAPE>;
  constexpr int KC = G::KC, NT = G::NT, TD = G::TD, TH = G::TH, TW = G::TW, IY = G::IY;
  constexpr int RX = G::RX, RAWP = G::RAWP, RAW_FLOATS = G::RAW_FLOATS, NS = G::NS, NQ = G::NQ, QR = G::QR;
  constexpr int PD = w3::PD, RD = PD + 1;
  static_assert(RD == 4, "the B ring is one chunk long: slot = group");
  constexpr int EX_FLOATS = 4 * NT * 4 * 256;            // the depth exchange of the epilogue (its own region: no barrier
  constexpr int SMEM_FLOATS = 2 * RAW_FLOATS + EX_FLOATS;  // between the last patch reads and its writes)
  static_assert(SMEM_FLOATS * 4 * 2 <= 160 * 1024, "two blocks per CU");
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = depth position a of the transform
  const int j = lane & 15, kq = lane >> 4;

  // tile order.  tc_slow (the host sets it when the layer's whole input fits the Infinity Cache): the output-channel block
  // is the SLOWEST index, so an XCD's slab of the tile order (dv_xcd_remap) works with one block's weights (1 MB of the
  // 128 -> 128 layer's 4-MB image, which does not stay in a 4-MB L2 beside the bricks: 1.23 GB of HBM traffic per launch
  // for 0.19 GB of tensors, 0.78 GB and -2.4 % with this order) and the bricks the blocks share come from the Infinity
  // Cache.  Otherwise the output-channel blocks of a tile are neighbours and share its bricks in one L2.
  unsigned t = dv_xcd_remap(blockIdx.x, gridDim.x);
  int tc = 0;
  if (!a.tc_slow) { tc = t % a.nco; t /= a.nco; }
  const int tx = t % a.ntx; t /= a.ntx;
  const int ty = t % a.nty; t /= a.nty;
  const int tz = t % a.ntz; t /= a.ntz;
  const int b = t % a.B;
  if (a.tc_slow) tc = t / a.B;
  const int x0 = tx * TW, y0 = ty * TH, z0 = tz * TD, co0 = tc * 32;

  // (not zeroed: the first chunk's MFMAs take the inline constant 0 as their C operand)
  f32x4 acc[16][NT];

  const size_t plane = (size_t)a.H * a.W;
  const size_t vol = (size_t)a.D * plane;
  const int vol_bytes = __builtin_amdgcn_readfirstlane((int)(vol * sizeof(float)));   // 4 * vol_bytes < 2^31 (host)
  auto sgpr64 = [](uint64_t v) __attribute__((always_inline)) {
    return (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
           ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
  };

  // ---- raw staging plan: the 4 channels x IZ x IY rows x QR aligned quads of a chunk's brick dealt to the 256 threads as a
  // whole; one descriptor per chunk (base = first channel, records = the channels that exist), the channel is part of the
  // lane's offset; quads outside the volume carry the offset 2^31 (out of range: zero).  Quad q of a row holds
  // x0 - 4 + 4 q .. + 3 and lands in the row's columns 4 q - 3 .. 4 q (column c = x0 - 1 + c): one float (b32), an aligned
  // pair (b64), one float (b32); the three floats in front of a row's column 0 fall into the padding of the row before
  // (of the guard quad for the very first row), the ones past column IX - 1 into the row's own padding ----
  unsigned sob[NS];
  int wo[NS];                                                     // byte offset of column 4 q of the quad's row, raw buffer 0
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int e = tid + 256 * i;
    const int cl = e / (G::IZ * IY * QR), r = e - cl * (G::IZ * IY * QR);
    const int rw = r / QR, cq = r - rw * QR;
    const int zz = rw / IY, yy = rw - zz * IY;
    const int z = z0 - 1 + zz, y = y0 - 1 + yy, x = x0 - 4 + 4 * cq;
    const bool in_brick = e < NQ;
    const bool ok = in_brick && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    sob[i] = ok ? (unsigned)cl * (unsigned)vol_bytes + (unsigned)((z * a.H + y) * a.W + x) * 4u : 0x80000000u;
    // (threads past the brick write their zeros into the guard quad)
    wo[i] = in_brick ? 4 * (8 + cl * RAWP + rw * RX + 4 * cq) : 4 * 4;
  }
  f32x4 vin[2][NS];                                               // bricks in flight: set = brick & 1
  uint64_t fb = sgpr64(reinterpret_cast<uint64_t>(a.in + (size_t)b * a.Cin * vol));     // first channel of the next brick
  int left = a.Cin;                                                                      // channels from there on
  auto records = [&](int l) __attribute__((always_inline)) {     // clamp(l, 0, 4) * vol_bytes on the scalar unit
    int r;
    asm("s_min_i32 %0, %1, 4\n\ts_max_i32 %0, %0, 0\n\ts_mul_i32 %0, %0, %2" : "=&s"(r) : "s"(l), "s"(vol_bytes) : "scc");
    return r;
  };
  int nrec = records(left);
  // piece k of the brick that is next in the stream (pieces are requested strictly in order, brick after brick)
  auto fetch_piece = [&](int k, int set) __attribute__((always_inline)) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(fb), 0, nrec, 0x00020000);
    vin[set][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)sob[k], 0, 0));
    if (k == NS - 1) {
      fb += (uint64_t)(unsigned)(KC * vol_bytes);
      left -= KC;
      nrec = records(left);
    }
  };
  char* const smem_b = reinterpret_cast<char*>(smem);
  auto commit_piece = [&](int k, int set, int buf) __attribute__((always_inline)) {
    char* const p = smem_b + wo[k] + 4 * buf * RAW_FLOATS;
    const f32x4 v = vin[set][k];
    *reinterpret_cast<float*>(p - 12) = v[0];
    *reinterpret_cast<f32x2*>(p - 8) = (f32x2){v[1], v[2]};
    *reinterpret_cast<float*>(p) = v[3];
  };

  // ---- weights: this wave's depth position of the packed image; one descriptor over the image, lane part of the address
  // = lane * 16 bytes, the (chunk, piece) part is scalar ----
  const int n_chunk = (a.Cin + KC - 1) / KC;
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(sgpr64(reinterpret_cast<uint64_t>(a.wpk))), 0,
                                                     (int)(unsigned)((size_t)n_chunk * a.nco * w3::U_CHUNK * sizeof(float)),
                                                     0x00020000);
  const int wvoff = lane * 16;
  const int wpart0 = __builtin_amdgcn_readfirstlane((tc * 4 + wave) * w3::U_PART * (int)sizeof(float));
  const int wstep = __builtin_amdgcn_readfirstlane(a.nco * w3::U_CHUNK * (int)sizeof(float));
  f32x4 bq[RD][NT];
  // B fragments of group s of chunk ch into ring slot s (a chunk past the end reads the last one: finite weights, unused)
  auto load_b = [&](int ch, int s) __attribute__((always_inline)) {
    const int c = ch < n_chunk ? ch : n_chunk - 1;
    const int base = wpart0 + c * wstep + s * (NT * w3::PIECE * 4);
#pragma unroll
    for (int n = 0; n < NT; ++n)
      bq[s][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, base + n * w3::PIECE * 4, 0));
  };

  // ---- this lane's patch: in-plane tile j of the wave's two input planes, channel kq ----
  constexpr int GC = G::TC / 2;
  const int p_tr = 2 * ((j >> 2) / GC) + (j & 1), p_tc = 2 * ((j >> 2) % GC) + ((j >> 1) & 1);
  const int zA = wave == 0 ? 0 : (wave == 2 ? 2 : 1), zB = wave == 2 ? 1 : (wave == 3 ? 3 : 2);
  const float sgn = wave == 1 ? 1.f : -1.f;
  const int patch_lo = 8 + kq * RAWP + 2 * p_tr * RX + 2 * p_tc;
  const int pao = 4 * (patch_lo + zA * IY * RX), pbo = 4 * (patch_lo + zB * IY * RX);      // bytes, raw buffer 0
  f32x2 sgn2;
  {
    const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sgn)));
    sgn2 = (f32x2){sg, sg};
  }

  f32x2 vp[4][2];          // V of the patch in flight: [transform row][column pair], updated in place
  // rows r0, r0+1, r0+2 of the two planes of a patch (raw buffer `buf`), as the depth combination needs them
  struct Rows3 { f32x2 pa[3][2], pb[3][2]; };
  auto read_rows = [&](int buf, int r0, Rows3& q) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        q.pa[r][h] = *reinterpret_cast<const f32x2*>(smem_b + pao + 4 * (buf * RAW_FLOATS + (r0 + r) * RX + 2 * h));
        q.pb[r][h] = *reinterpret_cast<const f32x2*>(smem_b + pbo + 4 * (buf * RAW_FLOATS + (r0 + r) * RX + 2 * h));
      }
  };
  // Depth combination c_r = A_r + sgn B_r (a packed fma with the wave's sign: exact), then V rows (0, 1) from c0, c1, c2
  // [HI = false] or rows (2, 3) from c1, c2, c3 [HI = true] (`q` holds the three rows in that order): row combinations
  // r0 = c0 - c2, r1 = c1 + c2 / r2 = c2 - c1, r3 = c1 - c3 and per row the column combinations (t0-t2, t1+t2),
  // (t2-t1, t1-t3) as one v_pk_add_f32 each (conv3d_wino.hip).  ONE asm statement = one dense burst of 14 packed
  // instructions (a vector instruction alone between two fp32 MFMAs makes the shared pipe drain); the closing s_nop covers
  // the VALU -> MFMA read hazard that gfx950 does not interlock.
  auto transform_lo = [&](const Rows3& q) __attribute__((always_inline)) {
    f32x2 c0a, c0b, c1a, c1b, c2a, c2b, t0, t1, t2, t3;
    asm("v_pk_fma_f32 %4, %20, %26, %14 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %5, %21, %26, %15 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %6, %22, %26, %16 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %7, %23, %26, %17 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %8, %24, %26, %18 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %9, %25, %26, %19 op_sel_hi:[1,0,1]\n\t"
        "v_pk_add_f32 %10, %4, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %11, %5, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %12, %6, %8\n\t"
        "v_pk_add_f32 %13, %7, %9\n\t"
        "v_pk_add_f32 %0, %10, %11 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %1, %11, %10 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\t"
        "v_pk_add_f32 %2, %12, %13 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %3, %13, %12 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\t"
        "s_nop 1"
        : "=&v"(vp[0][0]), "=&v"(vp[0][1]), "=&v"(vp[1][0]), "=&v"(vp[1][1]),
          "=&v"(c0a), "=&v"(c0b), "=&v"(c1a), "=&v"(c1b), "=&v"(c2a), "=&v"(c2b), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(q.pa[0][0]), "v"(q.pa[0][1]), "v"(q.pa[1][0]), "v"(q.pa[1][1]), "v"(q.pa[2][0]), "v"(q.pa[2][1]),
          "v"(q.pb[0][0]), "v"(q.pb[0][1]), "v"(q.pb[1][0]), "v"(q.pb[1][1]), "v"(q.pb[2][0]), "v"(q.pb[2][1]), "s"(sgn2));
  };
  auto transform_hi = [&](const Rows3& q) __attribute__((always_inline)) {   // q = rows 1, 2, 3
    f32x2 c1a, c1b, c2a, c2b, c3a, c3b, t0, t1, t2, t3;
    asm("v_pk_fma_f32 %4, %20, %26, %14 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %5, %21, %26, %15 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %6, %22, %26, %16 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %7, %23, %26, %17 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %8, %24, %26, %18 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %9, %25, %26, %19 op_sel_hi:[1,0,1]\n\t"
        "v_pk_add_f32 %10, %6, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %11, %7, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %12, %4, %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %13, %5, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %10, %11 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %1, %11, %10 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\t"
        "v_pk_add_f32 %2, %12, %13 op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %3, %13, %12 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\t"
        "s_nop 1"
        : "=&v"(vp[2][0]), "=&v"(vp[2][1]), "=&v"(vp[3][0]), "=&v"(vp[3][1]),
          "=&v"(c1a), "=&v"(c1b), "=&v"(c2a), "=&v"(c2b), "=&v"(c3a), "=&v"(c3b), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(q.pa[0][0]), "v"(q.pa[0][1]), "v"(q.pa[1][0]), "v"(q.pa[1][1]), "v"(q.pa[2][0]), "v"(q.pa[2][1]),
          "v"(q.pb[0][0]), "v"(q.pb[0][1]), "v"(q.pb[1][0]), "v"(q.pb[1][1]), "v"(q.pb[2][0]), "v"(q.pb[2][1]), "s"(sgn2));
  };

  // ---- prologue: brick 0 into LDS, bricks 1 and 2 into the register sets, rows 2,3 of patch 0, the first B fragments ----
#pragma unroll
  for (int k = 0; k < NS; ++k) fetch_piece(k, 0);
#pragma unroll
  for (int s = 0; s < PD; ++s) load_b(0, s);
#pragma unroll
  for (int k = 0; k < NS; ++k) fetch_piece(k, 1);
#pragma unroll
  for (int k = 0; k < NS; ++k) commit_piece(k, 0, 0);
#pragma unroll
  for (int k = 0; k < NS; ++k) fetch_piece(k, 0);
  __syncthreads();
  {
    Rows3 q;
    read_rows(0, 1, q);
    transform_hi(q);
  }
  int cidx = 0;                                          // the chunk in flight

  // One chunk.  PH = c & 1: chunk c's raw brick is buffer PH, brick c + 1 (register set PH ^ 1) is committed to buffer
  // PH ^ 1 before the barrier, brick c + 3 is requested into that register set after it.  On entry: vp rows 2,3 = patch c,
  // the B fragments of groups 0 .. PD - 1 requested.
  auto chunk = [&](auto ph_c, auto first_c) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value, PN = PH ^ 1;
    constexpr bool FIRST = decltype(first_c)::value;
    constexpr int H0 = NS / 2;                          // staging split of the chunk body
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = w3::row_of_group(s);
      // B fragments PD groups ahead: group 3 of this chunk (s = 0), groups s - 1 of the next
      load_b(s == 0 ? cidx : cidx + 1, (s + PD) & 3);
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[row * 4 + e][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(
              vp[row][e >> 1][e & 1], bq[s][n][e], FIRST ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[row * 4 + e][n], 0, 0, 0);
      if (s == 0) {                                     // rows 0,1 of this chunk's patch; the next brick goes to LDS
        Rows3 q;
        read_rows(PH, 0, q);
        transform_lo(q);
#pragma unroll
        for (int k = 0; k < H0; ++k) commit_piece(k, PN, PN);
      }
      if (s == 1) {
#pragma unroll
        for (int k = H0; k < NS; ++k) commit_piece(k, PN, PN);
        __syncthreads();                                // brick c + 1 is complete; every wave is done with brick c - 1
      }
      if (s == 2) {                                     // rows 2,3 of the next chunk's patch; brick c + 3 into the free set
        Rows3 q;
        read_rows(PN, 1, q);
        transform_hi(q);
#pragma unroll
        for (int k = 0; k < H0; ++k) fetch_piece(k, PN);
      }
      if (s == 3) {
#pragma unroll
        for (int k = H0; k < NS; ++k) fetch_piece(k, PN);
      }
    }
    ++cidx;
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  chunk(P0{}, std::true_type{});
  if (n_chunk > 1) chunk(P1{}, std::false_type{});
#pragma unroll 1
  for (int c = 2; c < n_chunk; c += 2) {
    chunk(P0{}, std::false_type{});
    if (c + 1 < n_chunk) chunk(P1{}, std::false_type{});
  }

  // ---- epilogue 1: in-plane output transform of this wave's depth position (conv3d_wino.hip: At M A on packed fp32 over
  // the two tile rows), 16 positions -> a 4 x 4 output patch per lane and N-tile; into LDS: [a][n][row 4][lane][4] ----
  float* const ex = smem + 2 * RAW_FLOATS;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    f32x2 yq2[2][2][2];                    // [tile column h][output row of the tile][output column of the tile] over (tile row 0, 1)
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
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int yr = 2 * tr + r;
        const f32x4 v = {yq2[0][r][0][tr], yq2[0][r][1][tr], yq2[1][r][0][tr], yq2[1][r][1][tr]};
        *reinterpret_cast<f32x4*>(ex + (((wave * NT + n) * 4 + yr) * 64 + lane) * 4) = v;
      }
  }
  __syncthreads();

  // ---- epilogue 2: wave w = (output plane w >> 1, N-tile w & 1): plane 0 = Z0 + Z1 + Z2, plane 1 = Z1 - Z2 - Z3; BN scale /
  // bias, residual, activation.  A lane (cout j, tiles 4kq..4kq+3) holds a 4 x 4 output patch; the four kq lanes of a
  // channel write 64 contiguous bytes per row ----
  const int pz = wave >> 1, n = wave & 1;
  const int zo = z0 + pz;
  const int co = co0 + n * 16 + j;
  if (zo >= a.D || co >= a.Cout) return;
  const float es = pz ? -1.f : 1.f;
  const int xb = x0 + 4 * (kq % GC), yq = 4 * (kq / GC);
  const bool fast = a.fast_ok && x0 + TW <= a.W && y0 + TH <= a.H;
  const float slope = a.act == DV_ACT_RELU ? 0.f : (a.act == DV_ACT_LEAKY ? 0.01f : 1.f);
  const float sc = a.ch_scale ? a.ch_scale[co] : 1.f;
  const float bi = a.ch_bias ? a.ch_bias[co] : 0.f;
  const size_t cbase = (((size_t)b * a.Cout + co) * a.D + zo) * plane + (size_t)(y0 + yq) * a.W + xb;
  f32x4 rv[4];
  if (fast && a.residual) {
#pragma unroll
    for (int r = 0; r < 4; ++r) rv[r] = *reinterpret_cast<const f32x4*>(a.residual + cbase + (size_t)r * a.W);
  }
  f32x4 y[4];
#pragma unroll
  for (int yr = 0; yr < 4; ++yr) {
    const f32x4 za = *reinterpret_cast<const f32x4*>(ex + ((((pz + 0) * NT + n) * 4 + yr) * 64 + lane) * 4);
    const f32x4 zb = *reinterpret_cast<const f32x4*>(ex + ((((pz + 1) * NT + n) * 4 + yr) * 64 + lane) * 4);
    const f32x4 zc = *reinterpret_cast<const f32x4*>(ex + ((((pz + 2) * NT + n) * 4 + yr) * 64 + lane) * 4);
    y[yr] = (za + es * zb) + es * zc;                  // (es = +-1: the products are exact)
  }
  if (fast) {
#pragma unroll
    for (int yr = 0; yr < 4; ++yr) {
      f32x4 v = y[yr] * sc + bi;
      if (a.residual) v += rv[yr];
      if (a.act == DV_ACT_RELU) {
        v = __builtin_elementwise_max(v, v * 0.f);     // NaN stays NaN as in torch.relu
      } else if (a.act != DV_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a.act == DV_ACT_MISH ? dv_act(v[e], DV_ACT_MISH) : fmaxf(v[e], v[e] * slope);
      }
      *reinterpret_cast<f32x4*>(a.out + cbase + (size_t)yr * a.W) = v;
    }
  } else {
#pragma unroll
    for (int yr = 0; yr < 4; ++yr) {
      if (y0 + yq + yr >= a.H) continue;
      const size_t o = cbase + (size_t)yr * a.W;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (xb + e < a.W) {
          float u = fmaf(y[yr][e], sc, bi);
          if (a.residual) u += a.residual[o + e];
          a.out[o + e] = dv_act(u, a.act);
        }
    }
  }
}

// U = (G x G x G) g per (cout, cin);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; one thread per (chunk, cb, a, group, nt, k, n)
__global__ void pack_wino3_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout, int nchunk,
                                          int nco) {
  const size_t total = (size_t)nchunk * nco * 4 * 4 * 2 * 4 * 16;
  const double Gm[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int n = (int)(r % 16); r /= 16;
    const int k = (int)(r % 4); r /= 4;
    const int nt = (int)(r % 2); r /= 2;
    const int s = (int)(r % 4); r /= 4;
    const int ap = (int)(r % 4); r /= 4;
    const int cb = (int)(r % nco);
    const int ch = (int)(r / nco);
    const int co = cb * 32 + nt * 16 + n, ci = ch * 4 + k;
    const int row = w3::row_of_group(s);
    double g[3][3];                          // depth already combined: sum_kd G[a][kd] w[kd]
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) {
        double v = 0.0;
        if (co < Cout && ci < Cin)
          for (int kd = 0; kd < 3; ++kd) v += Gm[ap][kd] * (double)w[(((size_t)co * Cin + ci) * 3 + kd) * 9 + p * 3 + q];
        g[p][q] = v;
      }
    float* dst = wpk + i * 4;
    for (int e = 0; e < 4; ++e) {
      double v = 0.0;
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) v += Gm[row][p] * Gm[e][q] * g[p][q];
      dst[e] = (float)v;
    }
  }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

extern "C" size_t dv_conv3d_wino3_packed_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)cdiv(Cin, 4) * cdiv(Cout, 32) * w3::U_CHUNK;
}

extern "C" int dv_conv3d_wino3_supported(int Cin, int Cout, int D, int H, int W) {
  if (Cin <= 0 || Cout <= 1 || D <= 0 || H <= 0 || W <= 0) return 0;
  if (W % 4) return 0;                       // the brick is staged as aligned 16-byte quads: whole quads inside / outside a row
  // a chunk's four channel volumes are one buffer: 31-bit byte offsets
  return (size_t)D * H * W * sizeof(float) * 4 <= 0x7fffffffull ? 1 : 0;
}

extern "C" int dv_conv3d_wino3_pack_weights_f32(const float* w, float* wpacked, int Cin, int Cout, dv_stream_t stream) {
  DV_REQUIRE_PTR(w);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE(Cin > 0 && Cout > 0, DV_ERR_SHAPE);
  const int nchunk = cdiv(Cin, 4), nco = cdiv(Cout, 32);
  const size_t total = (size_t)nchunk * nco * 4 * 4 * 2 * 4 * 16;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_wino3_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpacked, Cin, Cout,
                     nchunk, nco);
  return dv_launch_status();
}

extern "C" int dv_conv3d_wino3_f32(const float* in, const float* wpacked, const float* ch_scale, const float* ch_bias,
                                   const float* residual, float* out, int B, int Cin, int D, int H, int W, int Cout,
                                   int act, dv_stream_t stream) {
  DV_REQUIRE_PTR(in);
  DV_REQUIRE_PTR(wpacked);
  DV_REQUIRE_PTR(out);
  DV_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, DV_ERR_SHAPE);
  DV_REQUIRE(dv_conv3d_wino3_supported(Cin, Cout, D, H, W), DV_ERR_UNSUPPORTED);
  DV_REQUIRE(act >= DV_ACT_NONE && act <= DV_ACT_LEAKY, DV_ERR_UNSUPPORTED);
  DV_REQUIRE(dv_aligned16(wpacked) && dv_aligned16(in), DV_ERR_ALIGN);
  W3Args a;
  a.in = in; a.wpk = wpacked; a.ch_scale = ch_scale; a.ch_bias = ch_bias; a.residual = residual; a.out = out;
  a.B = B; a.Cin = Cin; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
  a.fast_ok = (W % 4 == 0) && dv_aligned16(out) && (!residual || dv_aligned16(residual));
  hipStream_t s = (hipStream_t)stream;
  auto launch = [&](auto shape) {
    constexpr int SHAPE = decltype(shape)::value;
    using G = W3G<SHAPE>;
    a.ntx = cdiv(W, G::TW); a.nty = cdiv(H, G::TH); a.ntz = cdiv(D, G::TD); a.nco = cdiv(Cout, 32);
    const long long blocks = (long long)B * a.nco * a.ntz * a.nty * a.ntx;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return (int)DV_ERR_SHAPE;
    a.tc_slow = a.nco > 1 && (size_t)B * Cin * D * H * W * sizeof(float) <= ((size_t)128 << 20);   // (order only: same bits)
    hipLaunchKernelGGL((conv3d_wino3_kernel<SHAPE>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    return dv_launch_status();
  };
  // tile shape of a wave's 16 in-plane tiles: padded outputs x the measured cost of a tile of that shape relative to the
  // widest (narrow tiles stage more surplus columns: 24 / 16 / 12 floats per row for 18 / 10 / 6 needed; batch 8, three
  // layer sizes: 1 : 1.03 : 1.17) -- the 60-wide 128-channel plane takes 4 x 16 tiles padded to 64 (-5.5 % against 16 x 4)
  auto padded = [&](int tw, int th) { return (double)cdiv(W, tw) * tw * cdiv(H, th) * th; };
  const double p0 = padded(16, 4), p1 = padded(8, 8) * 1.03, p2 = padded(4, 16) * 1.17;
  int shape = 0;
  if (p1 < p0 && p1 <= p2) shape = 1;
  else if (p2 < p0 && p2 < p1) shape = 2;
  if (shape == 1) return launch(std::integral_constant<int, 1>{});
  if (shape == 2) return launch(std::integral_constant<int, 2>{});
  return launch(std::integral_constant<int, 0>{});
}
