// Probe: what does `buffer_load_dword ... offen lds` (LDS-DMA through a buffer descriptor) write for lanes whose offset fails
// the range check, and for lanes switched off in EXEC?     hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_lds_dma.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out, int n) {
  __shared__ float s[256];
  const int tid = threadIdx.x;
  s[tid] = -7.f;                        // sentinel
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long b = (unsigned long long)in;
  i32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  rs.z = n * 4;
  rs.w = 0x00020000;
  unsigned voff = (unsigned)((tid * 7) % n) * 4u;
  if ((tid & 15) == 3) voff = 0x80000000u;      // range check fails
  const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s + 64 * wave);
  if ((tid & 15) != 5) {                        // lanes 5, 21, ... are off
    unsigned m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[tid] = s[tid];
}
int main() {
  const int n = 1000;
  float *in, *out, h[1000], o[256];
  for (int i = 0; i < n; ++i) h[i] = 100.f + i;
  hipMalloc(&in, n * 4); hipMalloc(&out, 256 * 4);
  hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, in, out, n);
  hipMemcpy(o, out, 256 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    const float want = (t & 15) == 5 ? -7.f : ((t & 15) == 3 ? 0.f : 100.f + (t * 7) % n);
    if (o[t] != want) { if (bad < 8) printf("lane %d: got %g want %g\n", t, o[t], want); ++bad; }
  }
  printf("out-of-range lanes -> %g, masked lanes -> %g, mismatches %d of 256\n", o[3], o[5], bad);
  return bad != 0;
}
