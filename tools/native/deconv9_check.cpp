// First test of casmvs_deconv9_splitf16_forward_f32 (csrc/deconv9_splitf16.hip, written without a GPU run at the end of round 3), torch-free:
// against casmvs_conv3d_forward_f32(CASMVS_CONV_T2, 32 -> 16, with the skip tensor) on ragged small shapes with a float64 loop on the host beside
// both, twice for run-to-run bit stability, and on the cascade levels' shapes with the time of each kernel under dirtied caches.
//   deconv9_check [batch]
//   hipcc -O2 tools/native/deconv9_check.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/deconv9_check
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
  std::vector<float> w(32 * 16 * 27), scale(16), shift(16);
  for (auto &v : w) v = rnd() * 0.2f;
  for (int c = 0; c < 16; ++c) { scale[c] = 0.5f + 0.05f * c; shift[c] = 0.03f * (c - 8); }
  std::vector<unsigned char> packed(casmvs_deconv9_splitf16_packed_bytes());
  if (casmvs_deconv9_splitf16_pack(w.data(), scale.data(), shift.data(), packed.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
  std::vector<float> pf(casmvs_conv3d_packed_floats(CASMVS_CONV_T2, 32, 16));
  if (pf.empty() || casmvs_conv3d_pack_f32(CASMVS_CONV_T2, 32, 16, w.data(), scale.data(), shift.data(), pf.data())) { printf("pack f32: %s\n", casmvs_last_error()); return 3; }
  void *dpk;
  float *dpf;
  hipMalloc(&dpk, packed.size()); hipMalloc(&dpf, pf.size() * 4);
  hipMemcpy(dpk, packed.data(), packed.size(), hipMemcpyHostToDevice);
  hipMemcpy(dpf, pf.data(), pf.size() * 4, hipMemcpyHostToDevice);
  struct Shape { int B, Di, Hi, Wi; bool host; };
  const Shape shapes[] = {{1, 2, 4, 16, true}, {2, 3, 5, 10, true}, {1, 1, 9, 22, true}, {1, 5, 6, 34, true},
                          {batch, 12, 32, 40, false}, {batch, 8, 64, 80, false}, {batch, 2, 128, 160, false}};
  bool all_ok = true;
  for (const Shape &s : shapes) {
    const size_t ni = (size_t)s.Di * s.Hi * s.Wi, no = ni * 8, nin = (size_t)s.B * 32 * ni, nout = (size_t)s.B * 16 * no;
    const int Do = 2 * s.Di, Ho = 2 * s.Hi, Wo = 2 * s.Wi;
    std::vector<float> x(nin), sk(nout);
    for (auto &v : x) v = rnd() * 2.0f + 0.2f;
    for (auto &v : sk) v = rnd();
    float *dx, *dsk, *dy[2];
    hipMalloc(&dx, nin * 4); hipMalloc(&dsk, nout * 4); hipMalloc(&dy[0], nout * 4); hipMalloc(&dy[1], nout * 4);
    hipMemcpy(dx, x.data(), nin * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsk, sk.data(), nout * 4, hipMemcpyHostToDevice);
    auto run = [&](int k) {
      return k ? casmvs_deconv9_splitf16_forward_f32(dpk, dx, dsk, dy[1], s.B, s.Di, s.Hi, s.Wi, 0.01f, st)
               : casmvs_conv3d_forward_f32(CASMVS_CONV_T2, dpf, dx, dsk, dy[0], s.B, 32, 16, s.Di, s.Hi, s.Wi, 0.01f, st);
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
    printf("B=%d in %dx%dx%d: float32 MFMA %.1f us, split-f16 %.1f us (x%.3f); max |diff| / range = %.2e, non-finite %zu, repeat run %s", s.B, s.Di, s.Hi, s.Wi,
           us[0], us[1], us[0] / us[1], diff / range, nan, stable ? "equal" : "DIFFERENT");
    bool ok = nan == 0 && stable && diff / range < 3e-6;
    if (s.host) {
      std::vector<double> ref(nout, 0.0);
      for (int b = 0; b < s.B; ++b)
        for (int ci = 0; ci < 32; ++ci)
          for (int iz = 0; iz < s.Di; ++iz)
            for (int iy = 0; iy < s.Hi; ++iy)
              for (int ix = 0; ix < s.Wi; ++ix) {
                const double v = x[((size_t)b * 32 + ci) * ni + ((size_t)iz * s.Hi + iy) * s.Wi + ix];
                for (int co = 0; co < 16; ++co)
                  for (int kz = 0; kz < 3; ++kz)
                    for (int ky = 0; ky < 3; ++ky)
                      for (int kx = 0; kx < 3; ++kx) {
                        const int oz = 2 * iz - 1 + kz, oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx;
                        if (oz < 0 || oz >= Do || oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
                        ref[((size_t)b * 16 + co) * no + ((size_t)oz * Ho + oy) * Wo + ox] += v * w[(((size_t)ci * 16 + co) * 27) + kz * 9 + ky * 3 + kx];
                      }
              }
      double err[2] = {0, 0};
      for (int b = 0; b < s.B; ++b)
        for (int co = 0; co < 16; ++co)
          for (size_t i = 0; i < no; ++i) {
            const size_t o = ((size_t)b * 16 + co) * no + i;
            double v = ref[o] * scale[co] + shift[co];
            v = (v > 0 ? v : v * 0.01f) + sk[o];
            for (int k = 0; k < 2; ++k) err[k] = std::fmax(err[k], std::fabs(v - y[k][o]));
          }
      printf("; vs float64: float32 MFMA %.2e  split-f16 %.2e of the range", err[0] / range, err[1] / range);
      ok = ok && err[1] / range < 3e-6;
    }
    printf("  %s\n", ok ? "ok" : "FAILED");
    all_ok &= ok;
    hipFree(dx); hipFree(dsk); hipFree(dy[0]); hipFree(dy[1]);
  }
  printf(all_ok ? "ALL OK\n" : "FAILURES\n");
  return all_ok ? 0 : 1;
}
