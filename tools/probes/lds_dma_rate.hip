// Probe: how fast can ONE wave issue `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KB per instruction) against plain
// buffer_load_dwordx4 into registers?  L2-hot source (second repetition), 16 instructions back to back (inline asm, so that
// nothing is inserted between them), s_memtime around issue and around completion; scattered like a stride-2 brick
// (80-byte row pieces 960 bytes apart) or fully coalesced.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_rate.hip -o /tmp/lds_dma_rate && /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 16;
__global__ void probe(const float* src, int bytes, long long* out, float* sink, int mode, int scatter) {
  __shared__ __attribute__((aligned(16))) float lds[N * 256];
  const int lane = threadIdx.x & 63;
  auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  int voff = scatter ? (lane / 5) * 960 + (lane % 5) * 16 : lane * 16;
  for (int rep = 0; rep < 3; ++rep) {                      // rep 0 warms L2 / TLB
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 v[N];
    if (mode == 0) {                                        // LDS-DMA, M0 per piece
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + i * 256);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(la), "v"(voff + i * 16384), "s"(rs) : "memory");
      }
    } else if (mode == 1) {                                 // LDS-DMA, one M0, immediate offsets (these shift the source too)
      const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(la) : "memory");
#pragma unroll
      for (int i = 0; i < N; ++i)
        asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff + i * 16384), "s"(rs) : "memory");
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[i]) : "v"(voff + i * 16384), "s"(rs) : "memory");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (mode == 2) {
      f32x4 s = v[0];
#pragma unroll
      for (int i = 1; i < N; ++i) s += v[i];
      sink[lane] = s[0] + s[1] + s[2] + s[3];
    } else {
      sink[lane] = lds[lane * 4];
    }
    if (lane == 0) { out[rep * 2] = t1 - t0; out[rep * 2 + 1] = t2 - t0; }
  }
}
int main() {
  const int bytes = 64 << 20;
  float *src, *sink; long long* out;
  hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&sink, 1024); hipMalloc(&out, 64);
  const char* names[3] = {"LDS-DMA x4, M0 per piece", "LDS-DMA x4, one M0", "buffer_load x4 -> VGPR"};
  for (int scatter = 0; scatter < 2; ++scatter)
    for (int mode = 0; mode < 3; ++mode) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, bytes, out, sink, mode, scatter);
      long long h[6];
      if (hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
      printf("%-10s %-26s %d instructions: issued after %5lld, complete after %5lld ticks (L2-warm; cold: %lld / %lld)\n",
             scatter ? "scattered" : "coalesced", names[mode], N, h[4], h[5], h[0], h[1]);
    }
  return 0;
}
