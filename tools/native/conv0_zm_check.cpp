// Native check of casmvs_conv0_zmarch_forward_f32 (csrc/conv0_zmarch.hip), torch-free:
// against casmvs_conv0_splitf16_forward_f32 (same packed image) on ragged small shapes with a float64 convolution on the host beside both,
// twice for run-to-run bit stability, and on the cascade levels' shapes (cin 16: 32 x 256 x 320, cin 8: 8 x 512 x 640) with the time of
// each kernel under dirtied caches.   conv0_zm_check [batch]      (round 4's first run also timed both kernels on a tile grid shifted by 4 voxels
// in x - equal or slower, removed: profiles/r04_native_checks_first_run.txt)
//   hipcc -O2 tools/native/conv0_zm_check.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/conv0_zm_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "casmvs.h"

static uint32_t g_rng = 88172645u;
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
  struct Shape { int B, cin, D, H, W; bool host; };
  const Shape shapes[] = {{1, 8, 5, 20, 36, true},  {2, 16, 9, 17, 44, true}, {1, 8, 3, 33, 32, true}, {1, 16, 1, 16, 4, true}, {1, 8, 13, 50, 68, true},
                          {1, 32, 6, 20, 36, true}, {batch, 32, 48, 128, 160, false}, {batch, 16, 32, 256, 320, false}, {batch, 8, 8, 512, 640, false}, {1, 16, 32, 256, 320, false}, {1, 8, 8, 512, 640, false}};
  bool all_ok = true;
  for (const Shape &s : shapes) {
    const size_t n = (size_t)s.D * s.H * s.W, nin = (size_t)s.B * s.cin * n, nout = (size_t)s.B * 8 * n;
    std::vector<float> x(nin), w((size_t)8 * s.cin * 27), scale(8), shift(8);
    for (auto &v : x) v = rnd() * 3.0f + 0.5f;
    for (auto &v : w) v = rnd() * 0.2f;
    for (int c = 0; c < 8; ++c) { scale[c] = 0.5f + 0.1f * c; shift[c] = 0.05f * (c - 4); }
    const size_t pb = casmvs_conv0_splitf16_packed_bytes(s.cin);
    std::vector<unsigned char> packed(pb);
    if (casmvs_conv0_splitf16_pack(s.cin, w.data(), scale.data(), shift.data(), packed.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
    constexpr int K = 2;   // 0 tiled, 1 z-march
    const char *names[K] = {"tiled", "z-march"};
    float *dx, *dy[K];
    void *dp;
    hipMalloc(&dx, nin * 4); hipMalloc(&dp, pb);
    for (int k = 0; k < K; ++k) hipMalloc(&dy[k], nout * 4);
    hipMemcpy(dx, x.data(), nin * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, packed.data(), pb, hipMemcpyHostToDevice);
    auto run = [&](int k) {
      if (k == 0) return casmvs_conv0_splitf16_forward_f32(dp, dx, dy[0], s.B, s.cin, s.D, s.H, s.W, 0.01f, 0, st);
      return casmvs_conv0_zmarch_forward_f32(dp, dx, dy[1], s.B, s.cin, s.D, s.H, s.W, 0.01f, st);
    };
    std::vector<float> y[K], again(nout);
    double us[K] = {0, 0};
    for (int k = 0; k < K; ++k) {
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
    bool stable = true;   // the timed repetitions wrote the same bits as the first run
    for (int k = 1; k < K; ++k) {
      hipMemcpy(again.data(), dy[k], nout * 4, hipMemcpyDeviceToHost);
      stable = stable && memcmp(again.data(), y[k].data(), nout * 4) == 0;
    }
    double range = 0, diff = 0;
    size_t nan = 0;
    for (size_t i = 0; i < nout; ++i) range = std::fmax(range, std::fabs((double)y[0][i]));
    for (int k = 1; k < K; ++k)
      for (size_t i = 0; i < nout; ++i) {
        if (!std::isfinite(y[k][i])) ++nan;
        diff = std::fmax(diff, std::fabs((double)y[0][i] - y[k][i]));
      }
    printf("B=%d cin=%d %dx%dx%d:", s.B, s.cin, s.D, s.H, s.W);
    for (int k = 0; k < K; ++k) printf(" %s %.1f us%s", names[k], us[k], k + 1 < K ? "," : "");
    printf(" (x%.3f); max |diff| / range = %.2e, non-finite %zu, repeat runs %s", us[0] / us[1], diff / range, nan,
           stable ? "equal" : "DIFFERENT");
    bool ok = nan == 0 && stable && diff / range < 2e-6;
    if (s.host) {
      double err[K] = {0, 0};
      for (int b = 0; b < s.B; ++b)
        for (int co = 0; co < 8; ++co)
          for (int z = 0; z < s.D; ++z)
            for (int yy = 0; yy < s.H; ++yy)
              for (int xx = 0; xx < s.W; ++xx) {
                double acc = 0;
                for (int ci = 0; ci < s.cin; ++ci)
                  for (int kz = 0; kz < 3; ++kz)
                    for (int ky = 0; ky < 3; ++ky)
                      for (int kx = 0; kx < 3; ++kx) {
                        const int iz = z + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                        if (iz < 0 || iz >= s.D || iy < 0 || iy >= s.H || ix < 0 || ix >= s.W) continue;
                        acc += (double)w[(((size_t)co * s.cin + ci) * 27) + kz * 9 + ky * 3 + kx] * x[((size_t)b * s.cin + ci) * n + ((size_t)iz * s.H + iy) * s.W + ix];
                      }
                double v = acc * scale[co] + shift[co];
                v = v > 0 ? v : v * 0.01f;
                const size_t o = ((size_t)b * 8 + co) * n + ((size_t)z * s.H + yy) * s.W + xx;
                for (int k = 0; k < K; ++k) err[k] = std::fmax(err[k], std::fabs(v - y[k][o]));
              }
      printf("; vs float64: %.2e / %.2e of the range", err[0] / range, err[1] / range);
      for (int k = 1; k < K; ++k) ok = ok && err[k] / range < 2e-6;
    }
    printf("  %s\n", ok ? "ok" : "FAILED");
    all_ok &= ok;
    hipFree(dx); hipFree(dp);
    for (int k = 0; k < K; ++k) hipFree(dy[k]);
  }
  printf(all_ok ? "ALL OK\n" : "FAILURES\n");
  return all_ok ? 0 : 1;
}
