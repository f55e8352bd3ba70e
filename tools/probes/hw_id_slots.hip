// Which HW_ID bits tell the two co-resident work-groups of a CU apart?  Launches 512 blocks of 256 threads with 72 KB of LDS
// (two per CU), each block records HW_REG_HW_ID of its first wave and a timestamp; the host groups them by XCC / SE / CU.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hw_id_slots tools/probes/hw_id_slots.hip && /tmp/hw_id_slots
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out) {
  __shared__ float pad[18 * 1024];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);        // HW_REG_HW_ID, all 32 bits
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID (gfx940+)
  }
  for (volatile int i = 0; i < 200000; ++i) {}
  if (pad[(threadIdx.x + 1) & 255] < 0) out[0] = 0;
}
int main() {
  const int n = 512;
  unsigned* d;
  hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;   // key = (xcc, se, sh, cu)
  for (int b = 0; b < n; ++b) {
    const unsigned v = h[2 * b], x = h[2 * b + 1] & 15u;
    const unsigned key = (x << 16) | (((v >> 13) & 7u) << 8) | (((v >> 12) & 1u) << 4) | ((v >> 8) & 15u);
    cu[key].push_back(b);
  }
  printf("%zu distinct (xcc, se, sh, cu) keys for %d blocks\n", cu.size(), n);
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 12) break;
    printf("key %06x:", kv.first);
    for (int b : kv.second) {
      const unsigned v = h[2 * b];
      printf("  [blk %d wave_id %u simd %u tg_id %u queue %u]", b, v & 15u, (v >> 4) & 3u, (v >> 16) & 15u, (v >> 24) & 7u);
    }
    printf("\n");
  }
  return 0;
}
