// First test of casmvs_fnet_conv0_fused_f32 (csrc/fnet_conv0_fused.hip, written without a GPU run at the end of round 3), torch-free:
// against the two layer launches it replaces (casmvs_conv2d_forward_f32, CASMVS_CONV2D_K3: 3 -> 8, then 8 -> 8) on ragged small shapes
// with a float64 loop on the host beside both, twice for run-to-run bit stability, and at the headline size (24 images of 512 x 640)
// with the time of each form under dirtied caches.
//   hipcc -O2 tools/native/fnet_conv0_check.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/fnet_conv0_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "casmvs.h"

static uint32_t g_rng = 521288629u;
static float rnd() {
  g_rng ^= g_rng << 13; g_rng ^= g_rng >> 17; g_rng ^= g_rng << 5;
  return (float)(int32_t)g_rng * (1.0f / 2147483648.0f);
}

int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  void *dirty = nullptr;
  const size_t dirty_bytes = (size_t)512 << 20;
  hipMalloc(&dirty, dirty_bytes);
  std::vector<float> w0(8 * 3 * 9), w1(8 * 8 * 9), s0(8), b0(8), s1(8), b1(8);
  for (auto &v : w0) v = rnd() * 0.3f;
  for (auto &v : w1) v = rnd() * 0.2f;
  for (int c = 0; c < 8; ++c) { s0[c] = 0.6f + 0.1f * c; b0[c] = 0.05f * (c - 3); s1[c] = 1.2f - 0.07f * c; b1[c] = 0.03f * (4 - c); }
  std::vector<unsigned char> packed(casmvs_fnet_conv0_fused_packed_bytes());
  if (casmvs_fnet_conv0_fused_pack(w0.data(), s0.data(), b0.data(), w1.data(), s1.data(), b1.data(), packed.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
  std::vector<float> p0(casmvs_conv2d_packed_floats(CASMVS_CONV2D_K3, 3, 8)), p1(casmvs_conv2d_packed_floats(CASMVS_CONV2D_K3, 8, 8));
  if (casmvs_conv2d_pack_f32(CASMVS_CONV2D_K3, 3, 8, w0.data(), s0.data(), b0.data(), p0.data()) ||
      casmvs_conv2d_pack_f32(CASMVS_CONV2D_K3, 8, 8, w1.data(), s1.data(), b1.data(), p1.data())) { printf("pack f32: %s\n", casmvs_last_error()); return 3; }
  void *dpk;
  float *dp0, *dp1;
  hipMalloc(&dpk, packed.size()); hipMalloc(&dp0, p0.size() * 4); hipMalloc(&dp1, p1.size() * 4);
  hipMemcpy(dpk, packed.data(), packed.size(), hipMemcpyHostToDevice);
  hipMemcpy(dp0, p0.data(), p0.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dp1, p1.data(), p1.size() * 4, hipMemcpyHostToDevice);
  struct Shape { int N, H, W; bool host; };
  const Shape shapes[] = {{1, 20, 36, true}, {2, 33, 44, true}, {1, 16, 4, true}, {3, 50, 68, true}, {24, 512, 640, false}, {3, 512, 640, false}};
  bool all_ok = true;
  for (const Shape &s : shapes) {
    const size_t hw = (size_t)s.H * s.W, nin = (size_t)s.N * 3 * hw, nout = (size_t)s.N * 8 * hw;
    std::vector<float> x(nin);
    for (auto &v : x) v = rnd() * 2.0f;
    float *dx, *dmid, *dy[2];
    hipMalloc(&dx, nin * 4); hipMalloc(&dmid, nout * 4); hipMalloc(&dy[0], nout * 4); hipMalloc(&dy[1], nout * 4);
    hipMemcpy(dx, x.data(), nin * 4, hipMemcpyHostToDevice);
    auto run = [&](int k) {
      if (k) return casmvs_fnet_conv0_fused_f32(dpk, dx, dy[1], s.N, s.H, s.W, 0.01f, st);
      if (int rc = casmvs_conv2d_forward_f32(CASMVS_CONV2D_K3, dp0, dx, nullptr, dmid, s.N, 3, 8, s.H, s.W, 0.01f, st)) return rc;
      return casmvs_conv2d_forward_f32(CASMVS_CONV2D_K3, dp1, dmid, nullptr, dy[0], s.N, 8, 8, s.H, s.W, 0.01f, st);
    };
    std::vector<float> y[2], again(nout);
    double us[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
      hipMemset(dy[k], 0xff, nout * 4);
      if (run(k)) { printf("forward %d: %s\n", k, casmvs_last_error()); return 3; }
      if (hipStreamSynchronize(st) != hipSuccess) { printf("kernel %d failed: %s\n", k, hipGetErrorString(hipGetLastError())); return 4; }
      y[k].resize(nout);
      hipMemcpy(y[k].data(), dy[k], nout * 4, hipMemcpyDeviceToHost);
      const int reps = 6;
      float total = 0;
      for (int i = 0; i < reps; ++i) {
        hipMemsetAsync(dirty, i, dirty_bytes, st);
        hipEventRecord(e0, st);
        run(k);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        total += ms;
      }
      us[k] = total * 1e3 / reps;
    }
    hipMemcpy(again.data(), dy[1], nout * 4, hipMemcpyDeviceToHost);
    const bool stable = memcmp(again.data(), y[1].data(), nout * 4) == 0;
    double range = 0, diff = 0;
    size_t nan = 0;
    for (size_t i = 0; i < nout; ++i) {
      range = std::fmax(range, std::fabs((double)y[0][i]));
      if (!std::isfinite(y[1][i])) ++nan;
      diff = std::fmax(diff, std::fabs((double)y[0][i] - y[1][i]));
    }
    printf("N=%d %dx%d: two launches %.1f us, fused %.1f us (x%.3f); max |diff| / range = %.2e, non-finite %zu, repeat run %s", s.N, s.H, s.W, us[0], us[1],
           us[0] / us[1], diff / range, nan, stable ? "equal" : "DIFFERENT");
    bool ok = nan == 0 && stable && diff / range < 5e-6;
    if (s.host) {
      std::vector<double> mid((size_t)s.N * 8 * hw);
      auto conv = [&](auto in_at, int cin, const std::vector<float> &w, const std::vector<float> &sc, const std::vector<float> &sh, auto out_set) {
        for (int n = 0; n < s.N; ++n)
          for (int co = 0; co < 8; ++co)
            for (int yy = 0; yy < s.H; ++yy)
              for (int xx = 0; xx < s.W; ++xx) {
                double acc = 0;
                for (int ci = 0; ci < cin; ++ci)
                  for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                      const int iy = yy + ky - 1, ix = xx + kx - 1;
                      if (iy < 0 || iy >= s.H || ix < 0 || ix >= s.W) continue;
                      acc += (double)w[((size_t)co * cin + ci) * 9 + ky * 3 + kx] * in_at(n, ci, iy, ix);
                    }
                double v = acc * sc[co] + sh[co];
                out_set(n, co, yy, xx, v > 0 ? v : v * 0.01f);
              }
      };
      conv([&](int n, int c, int yy, int xx) { return (double)x[((size_t)n * 3 + c) * hw + (size_t)yy * s.W + xx]; }, 3, w0, s0, b0,
           [&](int n, int c, int yy, int xx, double v) { mid[((size_t)n * 8 + c) * hw + (size_t)yy * s.W + xx] = v; });
      double err[2] = {0, 0};
      conv([&](int n, int c, int yy, int xx) { return mid[((size_t)n * 8 + c) * hw + (size_t)yy * s.W + xx]; }, 8, w1, s1, b1,
           [&](int n, int c, int yy, int xx, double v) {
             const size_t o = ((size_t)n * 8 + c) * hw + (size_t)yy * s.W + xx;
             for (int k = 0; k < 2; ++k) err[k] = std::fmax(err[k], std::fabs(v - y[k][o]));
           });
      printf("; vs float64: two launches %.2e  fused %.2e of the range", err[0] / range, err[1] / range);
      ok = ok && err[1] / range < 5e-6;
    }
    printf("  %s\n", ok ? "ok" : "FAILED");
    all_ok &= ok;
    hipFree(dx); hipFree(dmid); hipFree(dy[0]); hipFree(dy[1]);
  }
  printf(all_ok ? "ALL OK\n" : "FAILURES\n");
  return all_ok ? 0 : 1;
}
