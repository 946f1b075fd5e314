// First GPU test of casmvs_conv_s2_splitf16_forward_f32 (csrc/conv_s2_splitf16.hip: conv1 / conv3 of CostRegNet on the f16 matrix cores, written in round 4
// with the CPU emulation as its only test), torch-free: against casmvs_conv3d_forward_f32(CASMVS_CONV_S2) on ragged small shapes with a float64 loop on the
// host beside both, twice for run-to-run bit stability, and on the cascade levels' shapes with the time of each kernel under dirtied caches.
//   conv_s2_check [batch]
//   hipcc -O2 tools/native/conv_s2_check.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/conv_s2_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "casmvs.h"

static uint32_t g_rng = 362436069u;
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
    std::vector<float> w((size_t)cout * cin * 27), scale(cout), shift(cout);
    for (auto &v : w) v = rnd() * 0.2f;
    for (int c = 0; c < cout; ++c) { scale[c] = 0.5f + 0.03f * c; shift[c] = 0.03f * (c - 8); }
    std::vector<unsigned char> packed(casmvs_conv_s2_splitf16_packed_bytes(cin, cout));
    if (packed.empty() || casmvs_conv_s2_splitf16_pack(cin, cout, w.data(), scale.data(), shift.data(), packed.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
    std::vector<float> pf(casmvs_conv3d_packed_floats(CASMVS_CONV_S2, cin, cout));
    if (pf.empty() || casmvs_conv3d_pack_f32(CASMVS_CONV_S2, cin, cout, w.data(), scale.data(), shift.data(), pf.data())) { printf("pack f32: %s\n", casmvs_last_error()); return 3; }
    void *dpk;
    float *dpf;
    hipMalloc(&dpk, packed.size()); hipMalloc(&dpf, pf.size() * 4);
    hipMemcpy(dpk, packed.data(), packed.size(), hipMemcpyHostToDevice);
    hipMemcpy(dpf, pf.data(), pf.size() * 4, hipMemcpyHostToDevice);
    struct Shape { int B, D, H, W; bool host; };
    const int f = layer ? 2 : 1;   // conv3 sees conv1's output volume
    const Shape shapes[] = {{1, 4, 8, 16, true}, {2, 6, 10, 72, true}, {1, 2, 26, 136, true}, {1, 10, 12, 40, true},
                            {batch, 48 / f, 128 / f, 160 / f, false}, {batch, 32 / f, 256 / f, 320 / f, false}, {batch, 8 / f, 512 / f, 640 / f, false}};
    for (const Shape &s : shapes) {
      const int Do = s.D / 2, Ho = s.H / 2, Wo = s.W / 2;   // (even sizes: the float32 kernel's requirement)
      const size_t ni = (size_t)s.D * s.H * s.W, no = (size_t)Do * Ho * Wo, nin = (size_t)s.B * cin * ni, nout = (size_t)s.B * cout * no;
      std::vector<float> x(nin);
      for (auto &v : x) v = rnd() * 2.0f + 0.2f;
      for (size_t i = 0; i < nin; i += 1013) x[i] *= 100.0f;
      float *dx, *dy[2];
      hipMalloc(&dx, nin * 4); hipMalloc(&dy[0], nout * 4); hipMalloc(&dy[1], nout * 4);
      hipMemcpy(dx, x.data(), nin * 4, hipMemcpyHostToDevice);
      auto run = [&](int k) {
        return k ? casmvs_conv_s2_splitf16_forward_f32(dpk, dx, dy[1], s.B, cin, cout, s.D, s.H, s.W, 0.01f, st)
                 : casmvs_conv3d_forward_f32(CASMVS_CONV_S2, dpf, dx, nullptr, dy[0], s.B, cin, cout, s.D, s.H, s.W, 0.01f, st);
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
      const double gb = (nin + nout) * 4e-9;
      printf("%d -> %d B=%d in %dx%dx%d: float32 MFMA %.1f us, split-f16 %.1f us (x%.3f, %.2f TB/s algorithmic); max |diff| / range = %.2e, non-finite %zu, repeat run %s",
             cin, cout, s.B, s.D, s.H, s.W, us[0], us[1], us[0] / us[1], gb / (us[1] * 1e-6) * 1e-3, diff / range, nan, stable ? "equal" : "DIFFERENT");
      bool ok = nan == 0 && stable && diff / range < 3e-6;
      if (s.host) {
        double err[2] = {0, 0};
        for (int b = 0; b < s.B; ++b)
          for (int co = 0; co < cout; ++co)
            for (int z = 0; z < Do; ++z)
              for (int yy = 0; yy < Ho; ++yy)
                for (int xx = 0; xx < Wo; ++xx) {
                  double acc = 0;
                  for (int ci = 0; ci < cin; ++ci)
                    for (int kz = 0; kz < 3; ++kz)
                      for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                          const int iz = 2 * z + kz - 1, iy = 2 * yy + ky - 1, ix = 2 * xx + kx - 1;
                          if (iz < 0 || iz >= s.D || iy < 0 || iy >= s.H || ix < 0 || ix >= s.W) continue;
                          acc += (double)w[((size_t)co * cin + ci) * 27 + kz * 9 + ky * 3 + kx] * x[((size_t)b * cin + ci) * ni + ((size_t)iz * s.H + iy) * s.W + ix];
                        }
                  double v = acc * scale[co] + shift[co];
                  v = v > 0 ? v : v * 0.01f;
                  const size_t o = ((size_t)b * cout + co) * no + ((size_t)z * Ho + yy) * Wo + xx;
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
