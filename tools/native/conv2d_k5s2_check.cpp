// First GPU test of casmvs_conv2d_k5s2_splitf16_forward_f32 (csrc/conv2d_k5s2_splitf16.hip: FeatureNet's conv1.0 / conv2.0 on the f16 matrix cores, written in
// round 4 with the CPU emulation as its only test), torch-free: against casmvs_conv2d_forward_f32(CASMVS_CONV2D_K5S2) on ragged small images with a float64
// loop on the host beside both, twice for run-to-run bit stability, and on the engine's shapes (3 x batch images) with the time of each under dirtied caches.
//   conv2d_k5s2_check [batch]
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

int main(int argc, char **argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 2;
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  void *dirty = nullptr;
  const size_t dirty_bytes = (size_t)512 << 20;
  hipMalloc(&dirty, dirty_bytes);
  bool all_ok = true;
  for (int layer = 0; layer < 2; ++layer) {
    const int cin = layer ? 16 : 8, cout = 2 * cin;
    std::vector<float> w((size_t)cout * cin * 25), scale(cout), shift(cout);
    for (auto &v : w) v = rnd() * 0.2f;
    for (int c = 0; c < cout; ++c) { scale[c] = 0.5f + 0.03f * c; shift[c] = 0.03f * (c - 8); }
    std::vector<unsigned char> packed(casmvs_conv2d_k5s2_splitf16_packed_bytes(cin, cout));
    if (packed.empty() || casmvs_conv2d_k5s2_splitf16_pack(cin, cout, w.data(), scale.data(), shift.data(), packed.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
    std::vector<float> pf(casmvs_conv2d_packed_floats(CASMVS_CONV2D_K5S2, cin, cout));
    if (pf.empty() || casmvs_conv2d_pack_f32(CASMVS_CONV2D_K5S2, cin, cout, w.data(), scale.data(), shift.data(), pf.data())) { printf("pack f32: %s\n", casmvs_last_error()); return 3; }
    void *dpk;
    float *dpf;
    hipMalloc(&dpk, packed.size()); hipMalloc(&dpf, pf.size() * 4);
    hipMemcpy(dpk, packed.data(), packed.size(), hipMemcpyHostToDevice);
    hipMemcpy(dpf, pf.data(), pf.size() * 4, hipMemcpyHostToDevice);
    struct Shape { int N, H, W; bool host; };
    const int f = layer ? 2 : 1;   // conv2.0 sees the half-resolution maps
    const Shape shapes[] = {{1, 20, 72, true}, {3, 34, 136, true}, {2, 6, 8, true}, {3 * batch, 512 / f, 640 / f, false}, {3, 512 / f, 640 / f, false}};
    for (const Shape &s : shapes) {
      const int Ho = s.H / 2, Wo = s.W / 2;
      const size_t hw = (size_t)s.H * s.W, ohw = (size_t)Ho * Wo, nin = (size_t)s.N * cin * hw, nout = (size_t)s.N * cout * ohw;
      std::vector<float> x(nin);
      for (auto &v : x) v = rnd() * 2.0f + 0.2f;
      for (size_t i = 0; i < nin; i += 1013) x[i] *= 100.0f;
      float *dx, *dy[2];
      hipMalloc(&dx, nin * 4); hipMalloc(&dy[0], nout * 4); hipMalloc(&dy[1], nout * 4);
      hipMemcpy(dx, x.data(), nin * 4, hipMemcpyHostToDevice);
      auto run = [&](int k) {
        return k ? casmvs_conv2d_k5s2_splitf16_forward_f32(dpk, dx, dy[1], s.N, cin, cout, s.H, s.W, 0.01f, st)
                 : casmvs_conv2d_forward_f32(CASMVS_CONV2D_K5S2, dpf, dx, nullptr, dy[0], s.N, cin, cout, s.H, s.W, 0.01f, st);
      };
      std::vector<float> y[2], again(nout);
      double us[2] = {0, 0};
      for (int k = 0; k < 2; ++k) {
        hipMemset(dy[k], 0xff, nout * 4);
        if (run(k)) { printf("forward %d: %s\n", k, casmvs_last_error()); return 3; }
        if (hipStreamSynchronize(st) != hipSuccess) { printf("kernel %d failed: %s\n", k, hipGetErrorString(hipGetLastError())); return 4; }
        y[k].resize(nout);
        hipMemcpy(y[k].data(), dy[k], nout * 4, hipMemcpyDeviceToHost);
        float total = 0;
        for (int i = 0; i < 6; ++i) {
          hipMemsetAsync(dirty, i, dirty_bytes, st);
          hipEventRecord(e0, st);
          run(k);
          hipEventRecord(e1, st);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          total += ms;
        }
        us[k] = total * 1e3 / 6;
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
      printf("%d -> %d N=%d %dx%d: float32 MFMA %.1f us, split-f16 %.1f us (x%.3f, %.2f TB/s algorithmic); max |diff| / range = %.2e, non-finite %zu, repeat run %s", cin, cout, s.N, s.H,
             s.W, us[0], us[1], us[0] / us[1], (nin + nout) * 4e-9 / (us[1] * 1e-6) * 1e-3, diff / range, nan, stable ? "equal" : "DIFFERENT");
      bool ok = nan == 0 && stable && diff / range < 3e-6;
      if (s.host) {
        double err[2] = {0, 0};
        for (int n = 0; n < s.N; ++n)
          for (int co = 0; co < cout; ++co)
            for (int yy = 0; yy < Ho; ++yy)
              for (int xx = 0; xx < Wo; ++xx) {
                double acc = 0;
                for (int ci = 0; ci < cin; ++ci)
                  for (int ky = 0; ky < 5; ++ky)
                    for (int kx = 0; kx < 5; ++kx) {
                      const int iy = 2 * yy + ky - 2, ix = 2 * xx + kx - 2;
                      if (iy < 0 || iy >= s.H || ix < 0 || ix >= s.W) continue;
                      acc += (double)w[((size_t)co * cin + ci) * 25 + ky * 5 + kx] * x[((size_t)n * cin + ci) * hw + (size_t)iy * s.W + ix];
                    }
                double v = acc * scale[co] + shift[co];
                v = v > 0 ? v : v * 0.01f;
                const size_t o = ((size_t)n * cout + co) * ohw + (size_t)yy * Wo + xx;
                for (int k = 0; k < 2; ++k) err[k] = std::fmax(err[k], std::fabs(v - y[k][o]));
              }
        printf("; vs float64: float32 MFMA %.2e, split-f16 %.2e of the range", err[0] / range, err[1] / range);
        ok = ok && err[1] / range < 2e-6;
      }
      printf("  %s\n", ok ? "ok" : "FAILED");
      all_ok = all_ok && ok;
      hipFree(dx); hipFree(dy[0]); hipFree(dy[1]);
    }
    hipFree(dpk); hipFree(dpf);
  }
  printf(all_ok ? "ALL OK\n" : "FAILED\n");
  return all_ok ? 0 : 1;
}
