// Device-side plane-sweep arithmetic shared by every cost-volume kernel (costvol.hip: gather kernels,
// costvol_lds.hip: LDS-staged kernels, costvol_bwd.hip: the scatter transpose).
//
// Reference semantics: models/modules.py:59-89 (homo_warp) + ATen's grid_sampler (bilinear, zeros padding,
// align_corners=True), operation by operation in fp32; the library is built with -ffp-contract=off so that
// every product and sum below is separately rounded unless an fmaf() is written out.
#pragma once
#include <hip/hip_runtime.h>

namespace casmvs_dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));  // 4-byte aligned pair

// The 2x2 bilinear footprint as two row pairs (x, x+1): xl = column of the LEFT element (clamped so that
// both columns exist: 0 <= xl <= W-2), yn / ys = the north / south rows (always inside the image).  The four
// weights carry ATen's zeros padding (0 for a tap outside the image); a row or column that is outside is
// replaced by a neighbouring valid one with weight 0, so every address below is a valid image address.
struct Taps {
  int xl, yn, ys;
  float w_nl, w_nr, w_sl, w_sr;  // north-left, north-right, south-left, south-right
};

// a / b.  Default: r = v_rcp_f32(b) (1 ulp) and one Newton step on the quotient - <= 1 ulp from the correctly
// rounded result (usually equal to it) at 1/3 of the instruction slots of the IEEE expansion.
// -DCASMVS_IEEE_DIV builds the correctly rounded division instead (A/B of depth-index flips, tools/gpu_ab_div.sh).
// Non-finite intermediates (depth ~ 0) end as NaN / inf coordinates either way and drop the tap.
__device__ __forceinline__ float div_by(float a, float b, float rcp_b) {
#ifdef CASMVS_IEEE_DIV
  (void)rcp_b;
  return a / b;
#else
  const float q = a * rcp_b;
  const float r = fmaf(-b, q, a);
  return fmaf(r, rcp_b, q);
#endif
}

// Where ref pixel (x, y) at depth dv lands in the source view whose (P_src @ inv(P_ref))[:3] is P (12 floats, row-major 3x4): the integer part of
// ATen's un-normalised sampling position and the fractions.  Everything may be NaN / +-inf (depth ~ 0): the consumers' bounds tests drop such taps.
struct SweepCoords {
  float x0, y0;   // floor(ix), floor(iy)
  float tw, tn;   // ix - x0, iy - y0: the weights of column x0 + 1 / row y0 + 1 (ATen: e = 1 - w, s = 1 - n for column x0 / row y0)
};

__device__ __forceinline__ SweepCoords plane_sweep_coords(const float *__restrict__ P, float xf, float yf, float dv, int W, int H) {
  // src_grid_d = R @ (x, y, 1)^T + T / depth                       (modules.py:72)
  float rx = fmaf(P[2], 1.0f, fmaf(P[1], yf, P[0] * xf));
  float ry = fmaf(P[6], 1.0f, fmaf(P[5], yf, P[4] * xf));
  float rz = fmaf(P[10], 1.0f, fmaf(P[9], yf, P[8] * xf));
  const float rdv = __builtin_amdgcn_rcpf(dv);
  float qx = rx + div_by(P[3], dv, rdv);
  float qy = ry + div_by(P[7], dv, rdv);
  float qz = rz + div_by(P[11], dv, rdv);
  // negative depth -> somewhere outside the image                  (modules.py:76-79).  Selects, not branches:
  // the compiler turned the if / else-if chains of this function into ~10 exec-masked regions per call.
  const bool behind = qz <= 1e-7f;
  qx = behind ? (float)W : qx;
  qy = behind ? (float)H : qy;
  qz = behind ? 1.0f : qz;
  const float rqz = __builtin_amdgcn_rcpf(qz);
  float u = div_by(qx, qz, rqz);  // modules.py:81
  float v = div_by(qy, qz, rqz);
  // scale to [-1, 1] (modules.py:83-84) and ATen's un-normalisation (align_corners=True)
  const float hx = (float)(W - 1) * 0.5f, hy = (float)(H - 1) * 0.5f;
  float gx = div_by(u, hx, __builtin_amdgcn_rcpf(hx)) - 1.0f;
  float gy = div_by(v, hy, __builtin_amdgcn_rcpf(hy)) - 1.0f;
  float ix = ((gx + 1.0f) * 0.5f) * (float)(W - 1);
  float iy = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
  SweepCoords c;
  c.x0 = floorf(ix);
  c.y0 = floorf(iy);
  c.tw = ix - c.x0;   // ATen CPU kernel: w = x - x_w, e = 1 - w
  c.tn = iy - c.y0;
  return c;
}

// ATen's bounds-checked taps from the coordinates: every tap inside the image with its weight, every other one with weight 0 at a valid address.
__device__ __forceinline__ Taps taps_from_coords(const SweepCoords &c, int W, int H) {
  const float x0 = c.x0, y0 = c.y0;
  const float tw = c.tw, te = 1.0f - tw;
  const float tn = c.tn, ts = 1.0f - tn;
  // Bounds tests in float: NaN / +-inf / huge coordinates fail every comparison, so the tap is
  // dropped exactly like ATen's zeros padding (its int index below is never used with a non-zero weight).
  // x: left element of the pair is column xl = clamp(x0, 0, W-2); columns x0 and x0+1 carry
  // weights te and tw when they exist.
  const float fW = (float)W, fH = (float)H;
  const bool x_both = (x0 >= 0.0f) & (x0 <= fW - 2.0f);   // both columns inside
  const bool x_right = x0 == -1.0f;                        // only column x0+1 = 0 inside
  const bool x_left = x0 == fW - 1.0f;                     // only column x0 = W-1 inside
  const int xi = (int)fminf(fmaxf(x0, 0.0f), fW - 2.0f);   // NaN -> 0
  const int xl = x_both ? xi : (x_left ? W - 2 : 0);
  const float wl = x_both ? te : (x_right ? tw : 0.0f);
  const float wr = x_both ? tw : (x_left ? te : 0.0f);
  const bool y0_in = (y0 >= 0.0f) & (y0 <= fH - 1.0f);
  const bool y1_in = (y0 >= -1.0f) & (y0 <= fH - 2.0f);
  // a row outside the image gets weight 0 and the address of the other row (of row 0 when both are outside)
  const int yi = (int)fminf(fmaxf(y0, -1.0f), fH - 1.0f);  // NaN -> -1
  const int yv0 = y0_in ? yi : 0;
  const int yv1 = y1_in ? yi + 1 : yv0;
  const float wn = y0_in ? ts : 0.0f, wsth = y1_in ? tn : 0.0f;
  Taps t;
  t.xl = xl;
  t.yn = y0_in ? yv0 : yv1;
  t.ys = yv1;
  t.w_nl = wl * wn;   // ATen: nw = e * s, ne = w * s, sw = e * n, se = w * n
  t.w_nr = wr * wn;
  t.w_sl = wl * wsth;
  t.w_sr = wr * wsth;
  return t;
}

// Coordinates + bilinear taps of ref pixel (x, y) at depth dv in the source view whose
// (P_src @ inv(P_ref))[:3] is P (12 floats, row-major 3x4).
__device__ __forceinline__ Taps plane_sweep_taps(const float *__restrict__ P, float xf, float yf,
                                                 float dv, int W, int H) {
  return taps_from_coords(plane_sweep_coords(P, xf, yf, dv, W, H), W, H);
}

// true when at least one tap of the footprint carries weight (the voxel projects into the source image)
// (the weights are products of selected values in [0, 1] - never negative, never NaN: a NaN coordinate fails every bounds
// test above and selects the constants 0 - so "any weight != 0" is "the largest weight > 0": 4 instructions instead of 7)
__device__ __forceinline__ bool taps_live(const Taps &t) {
  return fmaxf(fmaxf(t.w_nl, t.w_nr), fmaxf(t.w_sl, t.w_sr)) > 0.0f;
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int imm) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + imm, 0, 0));
}

// A pointer the compiler must treat as wave-uniform (SGPRs).  Without it LLVM carries the map bases
// through a divergent region in VGPRs and wraps EVERY buffer load in a waterfall loop (measured:
// 637 VALU instructions per wave instead of ~300, the kernel became VALU-bound).
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace casmvs_dev
