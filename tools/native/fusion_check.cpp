// Stand-alone check of the two depth-fusion kernels through the C ABI (no torch, no Python: starts in milliseconds):
// casmvs_fuse_reference_view vs casmvs_fuse_reference_view_paired on a synthetic fronto-parallel scene at the reference's eval
// size (1152 x 864, 10 source views) - bit equality of every output incl. the per-view ones, and the time of each kernel.
//   hipcc -O2 tools/native/fusion_check.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/fusion_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "casmvs.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

template <class T>
T *upload(const std::vector<T> &v) {
  T *d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
  hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}

struct Outputs {
  float *depth;
  double *image;
  int32_t *count;
  unsigned char *mask;
  float *xyz;
  unsigned char *mgeo;
  float *dreproj;
  unsigned char *is2r;
};

int main(int argc, char **argv) {
  const int H = argc > 2 ? atoi(argv[1]) : 864, W = argc > 2 ? atoi(argv[2]) : 1152, S = argc > 3 ? atoi(argv[3]) : 10;
  const size_t hw = (size_t)H * W;
  uint32_t rng = 12345u;
  auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  std::vector<float> depth_ref(hw), depth_src(hw * S), proba((hw / 16));
  std::vector<unsigned char> image_ref(hw * 3), image_src(hw * 3 * S);
  for (size_t i = 0; i < hw; ++i) depth_ref[i] = 600.0f + 0.002f * (float)(rnd() % 1000);   // a plane with sub-millimetre noise: most pixels pass
  for (auto &v : depth_src) v = 600.0f + 0.002f * (float)(rnd() % 1000);
  for (auto &v : proba) v = (float)(rnd() % 1000) / 1000.0f;
  for (auto &v : image_ref) v = (unsigned char)rnd();
  for (auto &v : image_src) v = (unsigned char)rnd();
  const float f = 2892.33f * W / 1600.0f;
  std::vector<float> r2s(S * 12, 0.0f), s2r(S * 12, 0.0f), r2w(12, 0.0f);
  for (int s = 0; s < S; ++s) {
    const float b = 15.0f * (float)(s + 1 - (S + 1) / 2.0f);   // baseline along x (mm); the last views leave the image at the border columns
    for (int k = 0; k < 3; ++k) r2s[s * 12 + 5 * k] = s2r[s * 12 + 5 * k] = 1.0f;
    r2s[s * 12 + 3] = -f * b;
    s2r[s * 12 + 3] = f * b;
    r2s[s * 12 + 7] = 0.37f * f;   // a sub-pixel-shift in y so that the vertical taps carry weight
    s2r[s * 12 + 7] = -0.37f * f;
  }
  const float cx = W / 2.0f, cy = H / 2.0f;   // inv(K [I|0]) rows
  r2w[0] = 1.0f / f; r2w[2] = -cx / f; r2w[5] = 1.0f / f; r2w[6] = -cy / f; r2w[10] = 1.0f;
  float *d_dr = upload(depth_ref), *d_ds = upload(depth_src), *d_pr = upload(proba), *d_r2s = upload(r2s), *d_s2r = upload(s2r), *d_r2w = upload(r2w);
  unsigned char *d_ir = upload(image_ref), *d_is = upload(image_src);
  if (!d_dr || !d_ds || !d_pr || !d_r2s || !d_s2r || !d_r2w || !d_ir || !d_is) { printf("allocation failed\n"); return 2; }
  Outputs o[2];
  for (auto &q : o) {
    HIP_OK(hipMalloc(&q.depth, hw * 4)); HIP_OK(hipMalloc(&q.image, hw * 24)); HIP_OK(hipMalloc(&q.count, hw * 4)); HIP_OK(hipMalloc(&q.mask, hw));
    HIP_OK(hipMalloc(&q.xyz, hw * 12)); HIP_OK(hipMalloc(&q.mgeo, hw * S)); HIP_OK(hipMalloc(&q.dreproj, hw * S * 4)); HIP_OK(hipMalloc(&q.is2r, hw * S * 3));
  }
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  auto run = [&](int which, bool per_view) {
    auto fn = which ? casmvs_fuse_reference_view_paired : casmvs_fuse_reference_view;
    Outputs &q = o[which];
    return fn(d_dr, d_ir, d_pr, d_ds, d_is, d_r2s, d_s2r, d_r2w, q.depth, q.image, q.count, q.mask, q.xyz, per_view ? q.mgeo : nullptr,
              per_view ? q.dreproj : nullptr, per_view ? q.is2r : nullptr, S, H, W, 0.5f, 3, st);
  };
  for (int which = 0; which < 2; ++which) {
    if (int rc = run(which, true)) { printf("launch %d failed: %s\n", which, casmvs_last_error()); return 3; }
  }
  HIP_OK(hipStreamSynchronize(st));
  auto same = [&](const void *a, const void *b, size_t n, const char *name) {
    std::vector<unsigned char> ha(n), hb(n);
    hipMemcpy(ha.data(), a, n, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), b, n, hipMemcpyDeviceToHost);
    size_t diff = 0;
    for (size_t i = 0; i < n; ++i) diff += ha[i] != hb[i];
    printf("  %-14s %s (%zu of %zu bytes differ)\n", name, diff ? "DIFFERENT" : "equal", diff, n);
    return diff == 0;
  };
  bool ok = true;
  ok &= same(o[0].depth, o[1].depth, hw * 4, "depth_refined");
  ok &= same(o[0].image, o[1].image, hw * 24, "image_refined");
  ok &= same(o[0].count, o[1].count, hw * 4, "mask_geo_sum");
  ok &= same(o[0].mask, o[1].mask, hw, "mask_final");
  ok &= same(o[0].xyz, o[1].xyz, hw * 12, "xyz_world");
  ok &= same(o[0].mgeo, o[1].mgeo, hw * S, "mask_geo");
  ok &= same(o[0].dreproj, o[1].dreproj, hw * S * 4, "depth_reproj");
  ok &= same(o[0].is2r, o[1].is2r, hw * S * 3, "image_s2r");
  std::vector<int32_t> cnt(hw);
  hipMemcpy(cnt.data(), o[0].count, hw * 4, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : cnt) mean += c;
  printf("  mean number of consistent source views per pixel: %.2f of %d\n", mean / hw, S);
  const double bytes = (double)hw * (4 + 3 + 4.0 / 16 + S * 7 + 4 + 24 + 4 + 1 + 12);
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which) {
    for (int i = 0; i < 3; ++i) run(which, false);
    HIP_OK(hipEventRecord(e0, st));
    const int reps = 30;
    for (int i = 0; i < reps; ++i) run(which, false);
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-34s %dx%d, %d source views: %.1f us per reference view, %.1f MB algorithmic -> %.0f GB/s = %.3f of the 8 TB/s HBM roof\n",
           which ? "casmvs_fuse_reference_view_paired" : "casmvs_fuse_reference_view", W, H, S, us, bytes / 1e6, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
  }
  printf(ok ? "ALL EQUAL\n" : "MISMATCH\n");
  return ok ? 0 : 1;
}
