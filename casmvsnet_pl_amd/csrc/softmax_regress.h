// softmax over depth + soft-argmin regression + confidence of ONE pixel (models/mvsnet.py:174-193,
// models/modules.py:95-104), shared by softmax_regress_kernel (depth_ops.hip) and the fused `prob` head
// (prob_regress.hip).  Built with -ffp-contract=off: every * and + is a separately rounded fp32 operation, in the
// reference's (torch's) order.
#pragma once
#include <hip/hip_runtime.h>

namespace casmvs {

// cp / dp: this pixel's cost / depth hypothesis of plane 0; consecutive planes are `stride` floats apart.
// DT > 0: D == DT, the D values stay in registers between the passes (one read of the cost); DT == 0: generic 3-pass.
template <int DT>
__device__ __forceinline__ void softmax_regress_pixel(const float *__restrict__ cp, const float *__restrict__ dp,
                                                      size_t stride, int Drt, float &depth, float &conf, int &index) {
  const int D = DT > 0 ? DT : Drt;
  constexpr int NR = DT > 0 ? DT : 1;
  float e[NR];
  float mx = -INFINITY;
  if (DT > 0) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      e[k] = cp[(size_t)k * stride];
      mx = fmaxf(mx, e[k]);
    }
  } else {
    for (int k = 0; k < D; ++k) mx = fmaxf(mx, cp[(size_t)k * stride]);
  }
  // p_k = exp(x_k - max) / sum                                   (F.softmax, mvsnet.py:175)
  float sum = 0.0f;
  if (DT > 0) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      e[k] = expf(e[k] - mx);
      sum = sum + e[k];
    }
  } else {
    for (int k = 0; k < D; ++k) sum = sum + expf(cp[(size_t)k * stride] - mx);
  }
  // depth = sum_k p_k d_k (modules.py:103); expected index = sum_k p_k k  (mvsnet.py:185-189)
  float dsum = 0.0f, isum = 0.0f;
  if (DT > 0) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      e[k] = e[k] / sum;
      dsum = dsum + e[k] * dp[(size_t)k * stride];
      isum = isum + e[k] * (float)k;
    }
  } else {
    for (int k = 0; k < D; ++k) {
      const float pk = expf(cp[(size_t)k * stride] - mx) / sum;
      dsum = dsum + pk * dp[(size_t)k * stride];
      isum = isum + pk * (float)k;
    }
  }
  // .long() truncates toward zero; isum >= 0 so this is floor; clamp to [0, D-1] (mvsnet.py:189-190)
  int idx;
  if (!(isum == isum)) {
    idx = 0;  // NaN: torch's float->int64 cast of NaN is INT64_MIN, clamped to 0
  } else {
    float cl = fminf(fmaxf(isum, 0.0f), (float)(D - 1));
    idx = (int)cl;
  }
  // confidence = p[idx-1] + p[idx] + p[idx+1] + p[idx+2], zeros outside [0, D) (mvsnet.py:181-193:
  // 4 * avg_pool3d of the (1, 2)-padded volume, window 4, then gather at idx)
  float c4 = 0.0f;
  if (DT > 0) {
#pragma unroll
    for (int k = 0; k < NR; ++k)
      if (k >= idx - 1 && k <= idx + 2) c4 = c4 + e[k];
  } else {
    for (int k = max(idx - 1, 0); k <= min(idx + 2, D - 1); ++k)
      c4 = c4 + expf(cp[(size_t)k * stride] - mx) / sum;
  }
  depth = dsum;
  conf = c4;
  index = idx;
}

}  // namespace casmvs
