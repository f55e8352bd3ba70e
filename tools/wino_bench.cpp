// Stand-alone timing harness of the Winograd 3-D conv (no Python, no torch): includes the kernel source so that
// timing-only ablation builds can be made with -D flags / edited copies.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_bench.cpp -o /tmp/wino_bench
//   /tmp/wino_bench [Cin Cout [D H W [tag]]]      (batch 8; default 32 32 48 128 240)
#include "../diffuvolume_amd/csrc/conv3d_wino.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  int B = 8, Cin = 32, Cout = 32, D = 48, H = 128, W = 240;
  if (argc > 2) { Cin = atoi(argv[1]); Cout = atoi(argv[2]); }
  if (argc > 5) { D = atoi(argv[3]); H = atoi(argv[4]); W = atoi(argv[5]); }
  size_t nin = (size_t)B * Cin * D * H * W, nout = (size_t)B * Cout * D * H * W;
  float *in, *out, *w, *wp, *sc, *bi;
  hipMalloc(&in, nin * 4); hipMalloc(&out, nout * 4); hipMalloc(&w, (size_t)Cin * Cout * 27 * 4);
  hipMalloc(&sc, Cout * 4); hipMalloc(&bi, Cout * 4);
  std::vector<float> h(nin);
  for (size_t i = 0; i < nin; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice);
  std::vector<float> hw((size_t)Cin * Cout * 27);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) % 977) / 977.f * 0.1f - 0.05f;
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> one(Cout, 1.f);
  hipMemcpy(sc, one.data(), Cout * 4, hipMemcpyHostToDevice);
  hipMemcpy(bi, one.data(), Cout * 4, hipMemcpyHostToDevice);
  hipMalloc(&wp, dv_conv3d_wino_packed_floats(Cin, Cout) * 4);
  dv_conv3d_wino_pack_weights_f32(w, wp, Cin, Cout, 0);
  { int nb = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv3d_wino_kernel<false, 1>, 256, 0); printf("occupancy blocks/CU = %d\n", nb);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); printf("sharedMemPerMultiprocessor %zu maxSharedPerBlock %zu\n", pr.maxSharedMemoryPerMultiProcessor, pr.sharedMemPerBlock); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) dv_conv3d_wino_f32(in, wp, sc, bi, nullptr, nullptr, out, B, Cin, D, H, W, Cout, 1, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int N = 5;
  for (int i = 0; i < N; ++i) dv_conv3d_wino_f32(in, wp, sc, bi, nullptr, nullptr, out, B, Cin, D, H, W, Cout, 1, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= N;
  std::vector<float> ho(1024);
  hipMemcpy(ho.data(), out + nout / 2, 4096, hipMemcpyDeviceToHost);
  double cs = 0; for (float v : ho) cs += v;
  printf("%s %d->%d %dx%dx%d: %.3f ms  %.1f TF-equivalent  checksum %.6f\n", argc > 6 ? argv[6] : "", Cin, Cout, D, H, W, ms,
         2.0 * nout * Cin * 27 / ms / 1e9, cs);
  return 0;
}
