// Shared helpers for the gfx950 kernels of libdiffuvolume_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/diffuvolume_hip.h"

#define DV_WAVE 64

#define DV_REQUIRE_PTR(p) \
  do {                    \
    if ((p) == nullptr) return DV_ERR_NULL; \
  } while (0)
#define DV_REQUIRE(cond, err) \
  do {                        \
    if (!(cond)) return (err); \
  } while (0)

// Launch errors are reported to the caller as a positive hipError_t.
static inline int dv_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DV_OK : (int)e;
}

static inline bool dv_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// exp(x) for x <= 0 (softmax terms after the max is subtracted): 2^(x*log2 e) on the hardware transcendental, with the
// rounding error of the product carried along -- t = fl(x*L), r = (x*L - t) + x*(log2 e - L) exactly by fma, and
// 2^(t+r) = 2^t * (1 + r ln 2) to first order (|r| <= 2^-24 |t| < 1e-5, so the next term is below 1e-10).
// Same ~1 ulp as expf() (the softmax kernels are bound by their exponentials; libm's version spends half its instructions on
// overflow / denormal handling that cannot occur here: results below 2^-126 flush to zero next to a sum >= 1).
__device__ __forceinline__ float dv_exp_le0(float x) {
  const float L = 1.44269504088896340736f, LL = 1.92596299112661746e-8f;
  x = x < -104.f ? -104.f : x;      // exp(-104) is below the smallest denormal: also keeps -inf from becoming inf - inf;
                                    // a NaN argument stays NaN (fmaxf would turn it into a ~0 term and hide it)
  const float t = x * L;
  const float r = fmaf(x, L, -t) + x * LL;
  const float p = __builtin_amdgcn_exp2f(t);
  return fmaf(p, r * 0.69314718055994530942f, p);
}

// The gate non-linearities of IGEV's ConvGRU (KITTI15/core/update.py:36-39), on the hardware exponential and reciprocal:
// libm's tanhf / expf + an IEEE division were ~40 / ~30 vector instructions per element, and a gate epilogue applies them
// to 32 outputs per lane on the pipe the fp32 MFMAs issue on (round 5: 11 % of the z / r launch, 17 % of the candidate's).
//   sigmoid(v) = 1 / (1 + e^-|v|) or its mirror e^-|v| / (1 + e^-|v|)           (the exponential never overflows)
//   tanh(v)    = sign(v) (1 - e) / (1 + e), e = e^-2|v|;  |v| < 1/8: the odd series to v^7 (no cancellation at 0)
// v_rcp_f32 is good to 1 ulp: both stay within ~3 ulp of the correctly rounded result (tests: 1e-6 against float64).
__device__ __forceinline__ float dv_sigmoid(float v) {
  const float e = dv_exp_le0(-fabsf(v));
  const float r = __builtin_amdgcn_rcpf(1.0f + e);
  return v >= 0.0f ? r : e * r;             // (NaN: e is NaN, both branches NaN)
}
__device__ __forceinline__ float dv_tanh(float v) {
  const float av = fabsf(v);
  const float e = dv_exp_le0(-2.0f * av);
  const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  const float v2 = v * v;
  const float small = av * fmaf(v2, fmaf(v2, fmaf(v2, -17.0f / 315.0f, 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
  return copysignf(av < 0.125f ? small : big, v);
}

__device__ __forceinline__ float dv_act(float v, int act) {
  switch (act) {
    case DV_ACT_RELU: return fmaxf(v, 0.0f);
    case DV_ACT_MISH: {
      // x * tanh(softplus(x)) (KITTI12/models/submodule.py:11-18).  With e = exp(x):
      // tanh(log(1+e)) = ((1+e)^2 - 1) / ((1+e)^2 + 1) = n / (n + 2),  n = e * (e + 2)
      // -- one exp and one division instead of exp + log1p + tanh (the Mish layers' epilogues were bound by
      // those three).  No cancellation anywhere; x > 20 is torch's softplus threshold, where the factor is 1.
      if (v > 20.0f) return v;
      const float e = expf(v);
      const float n = e * (e + 2.0f);
      return v * (n / (n + 2.0f));
    }
    case DV_ACT_LEAKY: return v > 0.0f ? v : 0.01f * v;
    case DV_ACT_SIGMOID: return dv_sigmoid(v);
    case DV_ACT_TANH: return dv_tanh(v);
    default: return v;
  }
}

// XCD-aware block remap: blocks b and b+8 share an XCD (and its L2), so give
// every XCD a contiguous slab of tiles.  Bijective for any grid size.
__device__ __forceinline__ unsigned dv_xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned xcd = bid & 7u, idx = bid >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
