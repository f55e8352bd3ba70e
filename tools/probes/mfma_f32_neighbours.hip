// Probe: what does an instruction issued by the OTHER wave of a SIMD cost a wave that streams fp32 MFMAs
// (v_mfma_f32_16x16x4_f32 back to back, independent accumulators)?  8 waves per block, one block per CU: waves 0-3 run
// N MFMAs each; waves 4-7 run R filler instructions of one kind each (R chosen so that they finish first).  The slope
// of the kernel time over R is the price of one filler instruction in matrix-pipe cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f32_neighbours.hip -o gpurun_tmp/mfma_neigh && gpurun_tmp/mfma_neigh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { K_NONE, K_VADD, K_VPKADD, K_SADD, K_SNOP, K_DSREAD64, K_DSREAD128, K_DSWRITE32, K_DSWRITE128, K_BUFLOAD, K_DMA,
            K_VMOV, K_VFMA, K_COUNT };
static const char* kind_name[] = {"none", "v_add_f32", "v_pk_add_f32", "s_add_u32", "s_nop 0", "ds_read_b64", "ds_read_b128",
                                  "ds_write_b32", "ds_write_b128", "buffer_load_dword", "global_load_lds_dwordx4", "v_mov_b32",
                                  "v_fma_f32"};

template <int KIND>
__global__ __launch_bounds__(512, 1) void probe(const float* src, float* out, int n_mfma, int n_fill, int prio) {
  __shared__ __attribute__((aligned(1024))) float lds[16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = src[lane], b = src[64 + lane];
    for (int it = 0; it < n_mfma / 16; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    out[(blockIdx.x * 4 + wave) * 64 + lane] = s[0] + s[1] + s[2] + s[3];
    return;
  }
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  if (prio == 2) __builtin_amdgcn_s_setprio(3);
  float x0 = src[lane], x1 = src[lane + 64], x2 = src[lane + 128], x3 = src[lane + 192];
  f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x1, x2}, p3 = {x3, x0};
  f32x4 q = {x0, x1, x2, x3};
  unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + (wave - 4) * 2048) + lane * 16;
  unsigned s0 = 1, s1 = 2;
  const float* gp = src + lane * 4;
  for (int it = 0; it < n_fill / 8; ++it) {
    __builtin_amdgcn_s_sleep(8);          // ~512 cycles: 8 fillers per ~600 cycles, the density of a staging wave
    if (KIND == K_VADD)
      asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
                   "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(q[0]));
    else if (KIND == K_VFMA)
      asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n"
                   "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(q[0]));
    else if (KIND == K_VMOV)
      asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n"
                   "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(q[0]));
    else if (KIND == K_VPKADD)
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));
    else if (KIND == K_SADD)
      asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n"
                   "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %0"
                   : "+s"(s0), "+s"(s1));
    else if (KIND == K_SNOP)
      asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
    else if (KIND == K_DSREAD64) {
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n"
                   "ds_read_b64 %0, %4 offset:2048\n ds_read_b64 %1, %4 offset:2560\n ds_read_b64 %2, %4 offset:3072\n ds_read_b64 %3, %4 offset:3584"
                   : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(laddr / 2 + (laddr & ~0x3ffu) / 2));
    } else if (KIND == K_DSREAD128) {
      f32x4 r0, r1, r2, r3;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                   "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(laddr));
      asm volatile("" :: "v"(r0), "v"(r1), "v"(r2), "v"(r3));
    } else if (KIND == K_DSWRITE32) {
      asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n"
                   "ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1 offset:1280\n ds_write_b32 %0, %1 offset:1536\n ds_write_b32 %0, %1 offset:1792" : : "v"(laddr / 4 + (laddr & ~0x3ffu) * 3 / 4), "v"(x0) : "memory");
    } else if (KIND == K_DSWRITE128) {
      asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n"
                   "ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:5120\n ds_write_b128 %0, %1 offset:6144\n ds_write_b128 %0, %1 offset:7168" : : "v"(laddr), "v"(q) : "memory");
    } else if (KIND == K_BUFLOAD) {
      float r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("global_load_dword %0, %8, off\n global_load_dword %1, %8, off offset:256\n global_load_dword %2, %8, off offset:512\n"
                   "global_load_dword %3, %8, off offset:768\n global_load_dword %4, %8, off offset:1024\n global_load_dword %5, %8, off offset:1280\n"
                   "global_load_dword %6, %8, off offset:1536\n global_load_dword %7, %8, off offset:1792"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(gp) : "memory");
      asm volatile("" :: "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7));
    } else if (KIND == K_DMA) {
      const unsigned la = __builtin_amdgcn_readfirstlane(laddr & ~0x3ffu);
      asm volatile("s_mov_b32 m0, %0\n s_nop 0\n"
                   "global_load_lds_dwordx4 %1, off\n global_load_lds_dwordx4 %1, off offset:1024\n"
                   "global_load_lds_dwordx4 %1, off offset:2048\n global_load_lds_dwordx4 %1, off offset:3072\n"
                   "global_load_lds_dwordx4 %1, off\n global_load_lds_dwordx4 %1, off offset:1024\n"
                   "global_load_lds_dwordx4 %1, off offset:2048\n global_load_lds_dwordx4 %1, off offset:3072"
                   : : "s"(la), "v"(gp) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  out[4096 * 64 + (blockIdx.x * 4 + wave - 4) * 64 + lane] = x0 + x1 + x2 + x3 + p0[0] + p1[1] + p2[0] + p3[1] + q[0] + (float)(s0 + s1);
}

template <int KIND>
float run(const float* src, float* out, int n_mfma, int n_fill, int prio) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<KIND><<<256, 512>>>(src, out, n_mfma, n_fill, prio);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) probe<KIND><<<256, 512>>>(src, out, n_mfma, n_fill, prio);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

template <int KIND>
void sweep(const float* src, float* out) {
  const int N = 65536;                       // MFMAs per wave: 2.1 M matrix-pipe cycles ~ 1 ms
  const float t0 = run<K_NONE>(src, out, N, 0, 0);
  printf("%-26s", kind_name[KIND]);
  for (int prio = 0; prio <= 1; ++prio)
    for (int R : {8192, 16384, 24576}) {
      const float t = run<KIND>(src, out, N, R, prio);
      // cycles the MFMA stream lost per filler instruction, in units of the baseline's cycles (N * 32 per t0)
      const double cyc = (t - t0) / t0 * (double)N * 32.0 / R;
      printf("  prio%d R=%5d: %6.3f ms (+%5.1f cyc/instr)", prio, R, t, cyc);
    }
  printf("   [baseline %.3f ms]\n", t0);
}

int main() {
  float *src, *out;
  hipMalloc(&src, 1 << 20); hipMalloc(&out, 8192 * 64 * 4);
  std::vector<float> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
  sweep<K_VADD>(src, out); sweep<K_VFMA>(src, out); sweep<K_VMOV>(src, out); sweep<K_VPKADD>(src, out);
  sweep<K_SADD>(src, out); sweep<K_SNOP>(src, out);
  sweep<K_DSREAD64>(src, out); sweep<K_DSREAD128>(src, out); sweep<K_DSWRITE32>(src, out); sweep<K_DSWRITE128>(src, out);
  sweep<K_BUFLOAD>(src, out); sweep<K_DMA>(src, out);
  return 0;
}
