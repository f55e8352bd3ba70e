// Probe: does gfx950 interlock a VALU write with an MFMA that reads the register as SrcA?  (It does not: two wait
// states are needed.  Result of a run: profiles/r02_valu_mfma_hazard_probe.txt; consequence: the s_nop inside the
// input-transform asm of csrc/conv3d_wino.hip and csrc/conv2d_wino.hip.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_mfma_hazard.hip -o /tmp/hz && /tmp/hz
// One wave; the producer, the gap and the consuming MFMA sit in ONE asm block so nothing can be scheduled between
// them; v[10:11] are poisoned with NaN first so that a stale read shows.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERN(NAME, PROD, GAP)                                                                                         \
  __global__ void NAME(const float* in, float* out) {                                                                 \
    const int lane = threadIdx.x;                                                                                     \
    f32x2 a = {in[lane], in[lane + 64]}, b = {in[lane + 128], in[lane + 192]};                                        \
    float bv = in[lane + 256];                                                                                        \
    f32x4 d;                                                                                                          \
    asm volatile("v_mov_b32 v10, 0x7fc00000\n\tv_mov_b32 v11, 0x7fc00000\n\ts_nop 7\n\ts_nop 7\n\t" PROD GAP        \
                 "v_mfma_f32_16x16x4_f32 %0, v10, %3, 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"                             \
                 : "=&v"(d)                                                                                           \
                 : "v"(a), "v"(b), "v"(bv)                                                                            \
                 : "v10", "v11", "v12");                                                                              \
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = d[i];                                                             \
  }
#define PK "v_pk_add_f32 v[10:11], %1, %2\n\t"
#define AD "v_add_f32 v10, %3, %3\n\t"
KERN(pk_ref, PK, "s_nop 7\n\ts_nop 7\n\t")
KERN(pk_0, PK, "")
KERN(pk_1, PK, "s_nop 0\n\t")
KERN(pk_2, PK, "s_nop 1\n\t")
KERN(pk_3, PK, "s_nop 2\n\t")
KERN(pk_v1, PK, "v_mov_b32 v12, 0\n\t")
KERN(pk_v2, PK, "v_mov_b32 v12, 0\n\tv_mov_b32 v12, 1\n\t")
KERN(ad_ref, AD, "s_nop 7\n\ts_nop 7\n\t")
KERN(ad_0, AD, "")
KERN(ad_2, AD, "s_nop 1\n\t")

int main() {
  float *in, *o;
  if (hipMalloc(&in, 4096) != hipSuccess || hipMalloc(&o, 1024) != hipSuccess) return 1;
  float h[320];
  for (int i = 0; i < 320; ++i) h[i] = (float)((i * 7919) % 31) * 0.25f - 3.f;
  (void)hipMemcpy(in, h, 1280, hipMemcpyHostToDevice);
  float ref[256], r[256];
  auto get = [&](void (*kk)(const float*, float*), float* dst) {
    hipLaunchKernelGGL(kk, dim3(1), dim3(64), 0, 0, in, o);
    (void)hipMemcpy(dst, o, 1024, hipMemcpyDeviceToHost);
  };
  auto run = [&](const char* name, void (*kk)(const float*, float*)) {
    get(kk, r);
    int nd = 0;
    for (int i = 0; i < 256; ++i) nd += !(r[i] == ref[i]);
    printf("%-46s %3d of 256 results wrong\n", name, nd);
  };
  get(pk_ref, ref);
  run("v_pk_add_f32 -> v_mfma, 0 wait states:", pk_0);
  run("1 wait state (s_nop 0):", pk_1);
  run("2 wait states (s_nop 1):", pk_2);
  run("3 wait states (s_nop 2):", pk_3);
  run("1 independent v_mov in between:", pk_v1);
  run("2 independent v_mov in between:", pk_v2);
  get(ad_ref, ref);
  run("v_add_f32 -> v_mfma, 0 wait states:", ad_0);
  run("v_add_f32 -> v_mfma, 2 wait states:", ad_2);
  return 0;
}
