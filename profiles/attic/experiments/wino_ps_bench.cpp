// Stand-alone A/B of the producer / consumer Winograd kernel (tools/experiments/conv3d_wino_ps.hip) against the all-in-one-wave
// kernel (csrc/conv3d_wino.hip): bitwise comparison of the whole output and timing, no Python, no torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Idiffuvolume_amd/csrc tools/experiments/wino_ps_bench.cpp \
//         diffuvolume_amd/csrc/conv3d_wino.hip tools/experiments/conv3d_wino_ps.hip -o gpurun_tmp/wino_ps_bench
//   gpurun_tmp/wino_ps_bench [Cin Cout [D H W [B [scale residual]]]]      (default 32 32 48 128 240, batch 8)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/diffuvolume_hip.h"
#ifdef DV_PS_STAMPS
extern "C" int dv_ps_read_stamps(unsigned long long* host);
#endif
extern "C" int dv_conv3d_wino_ps_f32(const float*, const float*, const float*, const float*, const float*, const float*,
                                     float*, int, int, int, int, int, int, int, dv_stream_t);

__global__ void compare_kernel(const float* a, const float* b, size_t n, unsigned long long* nbad, float* maxd) {
  unsigned long long bad = 0;
  float md = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = a[i], y = b[i];
    if (__float_as_uint(x) != __float_as_uint(y) && !(x == 0.f && y == 0.f)) ++bad;
    md = fmaxf(md, fabsf(x - y));
    if (x != x || y != y) md = 1e30f;
  }
  if (bad) atomicAdd(nbad, bad);
  if (md > 0.f) atomicMax(reinterpret_cast<unsigned*>(maxd), __float_as_uint(md));
}

int main(int argc, char** argv) {
  int B = 8, Cin = 32, Cout = 32, D = 48, H = 128, W = 240, use_scale = 0, use_res = 0;
  if (argc > 2) { Cin = atoi(argv[1]); Cout = atoi(argv[2]); }
  if (argc > 5) { D = atoi(argv[3]); H = atoi(argv[4]); W = atoi(argv[5]); }
  if (argc > 6) B = atoi(argv[6]);
  if (argc > 8) { use_scale = atoi(argv[7]); use_res = atoi(argv[8]); }
  const size_t nin = (size_t)B * Cin * D * H * W, nout = (size_t)B * Cout * D * H * W, nsc = (size_t)B * D * H * W;
  float *in, *o1, *o2, *w, *wp, *sc, *bi, *isc = nullptr, *res = nullptr;
  hipMalloc(&in, nin * 4); hipMalloc(&o1, nout * 4); hipMalloc(&o2, nout * 4); hipMalloc(&w, (size_t)Cin * Cout * 27 * 4);
  hipMalloc(&sc, Cout * 4); hipMalloc(&bi, Cout * 4);
  {
    std::vector<float> h(nin);
    const bool zero = getenv("DV_ZERO_INPUT") != nullptr;     // clock probe: all-zero operands draw less power
    for (size_t i = 0; i < nin; ++i) h[i] = zero ? 0.f : (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice);
  }
  std::vector<float> hw((size_t)Cin * Cout * 27);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = getenv("DV_ZERO_WEIGHTS") ? 0.f : (float)((i * 40503u) % 977) / 977.f * 0.1f - 0.05f;
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hs(Cout), hb(Cout);
  for (int i = 0; i < Cout; ++i) { hs[i] = 0.8f + 0.01f * i; hb[i] = 0.05f * (i % 7) - 0.1f; }
  hipMemcpy(sc, hs.data(), Cout * 4, hipMemcpyHostToDevice);
  hipMemcpy(bi, hb.data(), Cout * 4, hipMemcpyHostToDevice);
  if (use_scale) {
    hipMalloc(&isc, nsc * 4);
    std::vector<float> h(nsc);
    for (size_t i = 0; i < nsc; ++i) h[i] = (float)((i * 2246822519u) % 1013) / 1013.f;
    hipMemcpy(isc, h.data(), nsc * 4, hipMemcpyHostToDevice);
  }
  if (use_res) {
    hipMalloc(&res, nout * 4);
    std::vector<float> h(nout);
    for (size_t i = 0; i < nout; ++i) h[i] = (float)((i * 3266489917u) % 911) / 911.f - 0.5f;
    hipMemcpy(res, h.data(), nout * 4, hipMemcpyHostToDevice);
  }
  hipMalloc(&wp, dv_conv3d_wino_packed_floats(Cin, Cout) * 4);
  dv_conv3d_wino_pack_weights_f32(w, wp, Cin, Cout, 0);
  hipMemset(o1, 0xff, nout * 4); hipMemset(o2, 0xff, nout * 4);
  int r1 = dv_conv3d_wino_f32(in, wp, sc, bi, isc, res, o1, B, Cin, D, H, W, Cout, 1, 0);
  int r2 = dv_conv3d_wino_ps_f32(in, wp, sc, bi, isc, res, o2, B, Cin, D, H, W, Cout, 1, 0);
  hipError_t e = hipDeviceSynchronize();
  printf("launch codes %d %d sync %s\n", r1, r2, hipGetErrorString(e));
  unsigned long long* nbad; float* maxd;
  hipMalloc(&nbad, 8); hipMalloc(&maxd, 4); hipMemset(nbad, 0, 8); hipMemset(maxd, 0, 4);
  compare_kernel<<<2048, 256>>>(o1, o2, nout, nbad, maxd);
  unsigned long long hbad; float hmax;
  hipMemcpy(&hbad, nbad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hmax, maxd, 4, hipMemcpyDeviceToHost);
  printf("compare: %llu of %zu elements differ bitwise, max |d| = %g\n", hbad, nout, hmax);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 6;
  for (int which = 0; which < 4; ++which) {     // A B A B
    auto run = [&]() {
      return (which & 1) ? dv_conv3d_wino_ps_f32(in, wp, sc, bi, isc, res, o2, B, Cin, D, H, W, Cout, 1, 0)
                         : dv_conv3d_wino_f32(in, wp, sc, bi, isc, res, o1, B, Cin, D, H, W, Cout, 1, 0);
    };
    run(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < N; ++i) run();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= N;
    printf("%s %d->%d %dx%dx%d B%d scale%d res%d: %.3f ms  %.1f algorithmic TF  %.1f issued TF (%.3f of 157.3)\n",
           (which & 1) ? "producer/consumer" : "all-in-one-wave  ", Cin, Cout, D, H, W, B, use_scale, use_res, ms,
           2.0 * nout * Cin * 27 / ms / 1e9, 2.0 * nout * Cin * 27 / 2.25 / ms / 1e9, 2.0 * nout * Cin * 27 / 2.25 / ms / 1e9 / 157.3);
  }
#ifdef DV_PS_STAMPS
  {
    static unsigned long long st[2][48][8];
    dv_conv3d_wino_ps_f32(in, wp, sc, bi, isc, res, o2, B, Cin, D, H, W, Cout, 1, 0);
    hipDeviceSynchronize();
    dv_ps_read_stamps(&st[0][0][0]);
    const unsigned long long t0 = st[0][0][0];
    printf("consumer wave 0 of block 0 (cycles since its first stamp): item: start  compute_end(+d)  barrier_end(+d)  [epilogue_end]\n");
    for (int m = 0; m < 20; ++m)
      printf("  C %2d: %8lld  +%6lld  +%6lld  %s%lld\n", m, (long long)(st[0][m][0] - t0), (long long)(st[0][m][1] - st[0][m][0]),
             (long long)(st[0][m][2] - st[0][m][1]), st[0][m][3] ? "epi +" : "", st[0][m][3] ? (long long)(st[0][m][3] - st[0][m][2]) : 0LL);
    printf("block 0 consumer wave 0: %llu items in %lld s_memtime ticks = %.1f ticks per item; s_memrealtime (100 MHz) %lld -> %.3f ms, %.3f GHz tick rate\n",
           st[0][47][6], (long long)(st[0][47][7] - t0), (double)(st[0][47][7] - t0) / (double)st[0][47][6],
           (long long)(st[0][47][4] - st[0][47][5]), (double)(st[0][47][4] - st[0][47][5]) / 1e5,
           (double)(st[0][47][7] - st[0][7][3]) / ((double)(st[0][47][4] - st[0][47][5]) * 10.0));
    printf("producer wave 4, slot m: start  burst  W+SU  C+LU+F  R  -  barrier\n");
    for (int m = 0; m < 20; ++m)
      printf("  P %2d: %8lld  +%5lld +%5lld +%5lld +%5lld +%5lld +%5lld\n", m, (long long)(st[1][m][0] - t0),
             (long long)(st[1][m][1] - st[1][m][0]), (long long)(st[1][m][2] - st[1][m][1]), (long long)(st[1][m][3] - st[1][m][2]),
             (long long)(st[1][m][4] - st[1][m][3]), (long long)(st[1][m][5] - st[1][m][4]), (long long)(st[1][m][6] - st[1][m][5]));
  }
#endif
  return hbad ? 1 : 0;
}
