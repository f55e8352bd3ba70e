// K3 / K8: the elementwise side of one DDIM step of the DiffuVolume volume filter.
//  - dv_noise_prepare_*: DynamicHead add + clamp + rescale to [0,1]
//      (SceneFlow/models/head.py:74-77, acv_ddim.py:256-258)
//  - dv_encode_two_hot_f32: x_T encoding of the origin disparity (acv_ddim.py:403-419)
//  - dv_ddim_step: x_start re-encoding, noise prediction, renewal mask, DDIM update,
//      ensemble accumulation (acv_ddim.py:272-294, :318-369)
// All HBM-bound and tiny (the state is [B,48,h,w]); the point is to replace ~40 small
// ATen launches per step and to keep the reference's dtype promotions: the state is
// fp32 during the first step and fp64 afterwards (float64 schedule buffers).
#include "dv_common.h"

namespace {

__global__ void noise_prepare_f32_kernel(const float* __restrict__ x, const float* __restrict__ shift,
                                         float* __restrict__ n01, int HW, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t bc = i / HW;
  float n = x[i] + shift[bc];
  n = fminf(fmaxf(n, -1.0f), 1.0f);
  n01[i] = (n + 1.0f) / 2.0f;
}

__global__ void noise_prepare_f64_kernel(const double* __restrict__ x, const float* __restrict__ shift,
                                         double* __restrict__ n01, float* __restrict__ n01f, int HW,
                                         size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t bc = i / HW;
  double n = x[i] + (double)shift[bc];
  n = fmin(fmax(n, -1.0), 1.0);
  const double v = (n + 1.0) / 2.0;
  n01[i] = v;
  n01f[i] = (float)v;
}

// two-hot weights of a quarter-resolution disparity dq in [0, nbins): bin floor(dq) gets
// coff = floor - dq + 1, the next bin 1 - coff; floor == nbins-1 is a pure one-hot.
__device__ __forceinline__ float two_hot_value(int c, float dq, int nbins) {
  const float fl = floorf(dq);
  const int real = (int)fl;
  const float coff = fl - dq + 1.0f;
  float v = 0.f;
  if (real == nbins - 1) {
    v = (c == nbins - 1) ? 1.f : 0.f;
  } else {
    if (c == real) v = coff;
    if (c == real + 1) v = 1.0f - coff;
  }
  return v;
}

__global__ void encode_two_hot_kernel(const float* __restrict__ dq, float* __restrict__ x, int nbins,
                                      int hw, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int p = (int)(i % hw);
  const int c = (int)((i / hw) % nbins);
  const size_t b = i / ((size_t)hw * nbins);
  x[i] = two_hot_value(c, dq[b * hw + p], nbins) * 2.0f - 1.0f;
}

// bilinear /4 with align_corners=False is the mean of the central 2x2 of each 4x4 cell,
// evaluated in PyTorch's order h0*(w0*a+w1*b) + h1*(w0*c+w1*d) with all weights 0.5.
__device__ __forceinline__ float down4(float a, float b, float c, float d) {
  return 0.5f * (0.5f * a + 0.5f * b) + 0.5f * (0.5f * c + 0.5f * d);
}

// one thread per quarter-resolution pixel; loops over the nbins channels
__global__ void ddim_step_kernel(const float* __restrict__ disp, const float* __restrict__ unc,
                                 const float* __restrict__ used, const float* __restrict__ coords0,
                                 const float* __restrict__ n01f,
                                 const double* __restrict__ n01d, const float* __restrict__ epsf,
                                 const double* __restrict__ epsd, const double* __restrict__ fill,
                                 float* __restrict__ mask, float* __restrict__ x_start,
                                 double* __restrict__ pred_eps, double* __restrict__ x_next, int nbins, int h, int w, size_t total,
                                 dv_ddim_coef k) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % w);
  const int y = (int)((i / w) % h);
  const size_t b = i / ((size_t)w * h);
  const int W = 4 * w;
  const size_t full = b * (size_t)(16 * h) * w;  // b * H * W
  const size_t r0 = full + (size_t)(4 * y + 1) * W + 4 * x + 1, r1 = r0 + W;
  const float maxd = k.clamp_max;
  auto cl = [&](float v) { return fminf(fmaxf(v, 0.f), maxd); };
  // quarter-resolution disparity of this step's prediction (acv_ddim.py:272-274)
  float dq = down4(cl(disp[r0]), cl(disp[r0 + 1]), cl(disp[r1]), cl(disp[r1 + 1])) / 4.0f;
  if (coords0)   // IGEV: true_coords1 = clamp(coords0 + disp_net, 0, nbins-1) (igev_stereo_ddim.py:270-272)
    dq = fminf(fmaxf(coords0[i] + dq, 0.f), (float)(nbins - 1));
  // renewal mask (acv_ddim.py:322-338); unc == nullptr: disparity test only (IGEV, :316-317)
  auto keep = [&](size_t p) {
    return (fabsf(disp[p] - used[p]) < k.dif_thr && (!unc || unc[p] < k.unc_thr)) ? 1.f : 0.f;
  };
  float mk = mask[i] + down4(keep(r0), keep(r0 + 1), keep(r1), keep(r1 + 1));
  mk = fminf(fmaxf(mk, 0.f), 1.f);
  mask[i] = mk;
  const size_t hw = (size_t)h * w;
  const size_t base = b * nbins * hw + (size_t)y * w + x;
  const float san = (float)k.sqrt_alpha_next;  // 0-dim fp64 scalar times an fp32 tensor stays fp32
  const float sgf = (float)k.sigma;
  for (int c = 0; c < nbins; ++c) {
    const size_t o = base + (size_t)c * hw;
    float xs = two_hot_value(c, dq, nbins) * 2.0f - 1.0f;
    xs = fminf(fmaxf(xs, -1.0f), 1.0f);
    x_start[o] = xs;
    if (k.last && !pred_eps) continue;
    const double n = n01d ? n01d[o] : (double)n01f[o];
    const double pn = (k.sqrt_recip_alpha * n - (double)xs) / k.sqrt_recipm1_alpha;
    if (pred_eps) pred_eps[o] = pn;
    if (k.last) continue;
    const double se = epsd ? k.sigma * epsd[o] : (double)(sgf * epsf[o]);
    const double img = ((double)(xs * san) + k.c * pn) + se;
    x_next[o] = (mk == 0.f) ? fill[o] : img;
  }
}

// ens += cof * disp; with ens_dif_thr > 0 the IGEV output rule applies first:
// disp' = |disp - used| < thr ? disp : used (igev_stereo_ddim.py:323-327)
__global__ void ensemble_accumulate_kernel(const float* __restrict__ disp, const float* __restrict__ used,
                                           float* __restrict__ ens, float cof, float thr, size_t total) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float d = disp[i];
  if (thr > 0.f && !(fabsf(d - used[i]) < thr)) d = used[i];
  ens[i] += d * cof;
}

inline unsigned nblk(size_t total, int threads = 256) { return (unsigned)((total + threads - 1) / threads); }

}  // namespace

extern "C" int dv_noise_prepare_f32(const float* x_t, const float* shift, float* n01, int B, int C,
                                    int HW, dv_stream_t stream) {
  DV_REQUIRE_PTR(x_t);
  DV_REQUIRE_PTR(shift);
  DV_REQUIRE_PTR(n01);
  DV_REQUIRE(B > 0 && C > 0 && HW > 0, DV_ERR_SHAPE);
  const size_t total = (size_t)B * C * HW;
  hipLaunchKernelGGL(noise_prepare_f32_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x_t,
                     shift, n01, HW, total);
  return dv_launch_status();
}

extern "C" int dv_noise_prepare_f64(const double* x_t, const float* shift, double* n01, float* n01_f32,
                                    int B, int C, int HW, dv_stream_t stream) {
  DV_REQUIRE_PTR(x_t);
  DV_REQUIRE_PTR(shift);
  DV_REQUIRE_PTR(n01);
  DV_REQUIRE_PTR(n01_f32);
  DV_REQUIRE(B > 0 && C > 0 && HW > 0, DV_ERR_SHAPE);
  const size_t total = (size_t)B * C * HW;
  hipLaunchKernelGGL(noise_prepare_f64_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x_t,
                     shift, n01, n01_f32, HW, total);
  return dv_launch_status();
}

extern "C" int dv_encode_two_hot_f32(const float* disp_q, float* x, int B, int nbins, int hw,
                                     dv_stream_t stream) {
  DV_REQUIRE_PTR(disp_q);
  DV_REQUIRE_PTR(x);
  DV_REQUIRE(B > 0 && nbins > 1 && hw > 0, DV_ERR_SHAPE);
  const size_t total = (size_t)B * nbins * hw;
  hipLaunchKernelGGL(encode_two_hot_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, disp_q,
                     x, nbins, hw, total);
  return dv_launch_status();
}

extern "C" int dv_ddim_step(const float* disp, const float* unc, const float* used, const float* coords0,
                            const float* n01_f32,
                            const double* n01_f64, const float* eps_f32, const double* eps_f64,
                            const double* fill, float* mask, float* x_start, double* pred_eps,
                            double* x_next, float* ens, int B, int nbins, int h, int w, const dv_ddim_coef* coef, dv_stream_t stream) {
  DV_REQUIRE_PTR(disp);
  DV_REQUIRE_PTR(used);
  DV_REQUIRE_PTR(mask);
  DV_REQUIRE_PTR(x_start);
  DV_REQUIRE_PTR(coef);
  DV_REQUIRE(B > 0 && nbins > 1 && h > 0 && w > 0, DV_ERR_SHAPE);
  if (!coef->last || pred_eps) DV_REQUIRE((n01_f32 != nullptr) != (n01_f64 != nullptr), DV_ERR_NULL);
  if (!coef->last) {
    DV_REQUIRE((eps_f32 != nullptr) != (eps_f64 != nullptr), DV_ERR_NULL);
    DV_REQUIRE_PTR(fill);
    DV_REQUIRE_PTR(x_next);
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)B * h * w;
  hipLaunchKernelGGL(ddim_step_kernel, dim3(nblk(total, 128)), dim3(128), 0, s, disp, unc, used, coords0, n01_f32,
                     n01_f64, eps_f32, eps_f64, fill, mask, x_start, pred_eps, x_next, nbins, h, w, total, *coef);
  int rc = dv_launch_status();
  if (rc != DV_OK) return rc;
  if (ens != nullptr && coef->cof != 0.f) {
    const size_t full = total * 16;
    hipLaunchKernelGGL(ensemble_accumulate_kernel, dim3(nblk(full)), dim3(256), 0, s, disp, used, ens, coef->cof,
                       coef->ens_dif_thr, full);
    rc = dv_launch_status();
  }
  return rc;
}
