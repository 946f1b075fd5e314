// Stand-alone check of casmvs_prob_wgrad_f32 (csrc/prob_wgrad.hip) through the C ABI, without torch / Python: against the generic
// matrix-core weight gradient casmvs_conv_wgrad_f32 on the three cascade levels' `prob` shapes and ragged small ones, against a
// float64 loop on the host for the small ones, twice for bit-reproducibility, and the time of both.
//   hipcc -O2 tools/native/prob_wgrad_check.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/prob_wgrad_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "casmvs.h"

static uint32_t g_rng = 2463534242u;
static float rnd() {   // uniform in (-1, 1)
  g_rng ^= g_rng << 13; g_rng ^= g_rng >> 17; g_rng ^= g_rng << 5;
  return (float)(int32_t)g_rng * (1.0f / 2147483648.0f);
}

int main() {
  struct Shape { int B, D, H, W; bool host; };
  const Shape shapes[] = {{2, 5, 13, 20, true}, {1, 3, 7, 4, true}, {1, 9, 8, 36, true}, {1, 48, 128, 160, false}, {1, 32, 256, 320, false}, {1, 8, 512, 640, false}};
  hipStream_t st;
  if (hipStreamCreate(&st) != hipSuccess) { printf("no stream\n"); return 2; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bool all_ok = true;
  for (const Shape &s : shapes) {
    const size_t n = (size_t)s.D * s.H * s.W;
    std::vector<float> x((size_t)s.B * 8 * n), g((size_t)s.B * n);
    for (auto &v : x) v = rnd() + 0.3f;
    for (auto &v : g) v = rnd() * 0.01f;
    float *dx, *dg, *gw_new, *gw_old;
    void *ws_new, *ws_old;
    const size_t wb_new = casmvs_prob_wgrad_workspace_bytes(s.B, s.D, s.H, s.W), wb_old = casmvs_conv_wgrad_workspace_bytes(CASMVS_CONV_S1, s.B, 8, 1, s.D, s.H, s.W);
    if (!wb_new || !wb_old) { printf("shape %dx%dx%dx%d unsupported (%zu, %zu)\n", s.B, s.D, s.H, s.W, wb_new, wb_old); all_ok = false; continue; }
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dg, g.size() * 4); hipMalloc(&gw_new, 216 * 4); hipMalloc(&gw_old, 216 * 4);
    hipMalloc(&ws_new, wb_new); hipMalloc(&ws_old, wb_old);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice);
    hipMemset(gw_new, 0xff, 216 * 4);
    int rc = casmvs_prob_wgrad_f32(dx, dg, gw_new, ws_new, s.B, s.D, s.H, s.W, st);
    if (rc) { printf("prob_wgrad failed: %s\n", casmvs_last_error()); return 3; }
    rc = casmvs_conv_wgrad_f32(CASMVS_CONV_S1, dx, dg, gw_old, ws_old, s.B, 8, 1, s.D, s.H, s.W, st);
    if (rc) { printf("conv_wgrad failed: %s\n", casmvs_last_error()); return 3; }
    hipStreamSynchronize(st);
    float a[216], b[216], a2[216];
    hipMemcpy(a, gw_new, sizeof(a), hipMemcpyDeviceToHost);
    hipMemcpy(b, gw_old, sizeof(b), hipMemcpyDeviceToHost);
    casmvs_prob_wgrad_f32(dx, dg, gw_new, ws_new, s.B, s.D, s.H, s.W, st);
    hipStreamSynchronize(st);
    hipMemcpy(a2, gw_new, sizeof(a2), hipMemcpyDeviceToHost);
    double scale = 0, err_old = 0, err_host_new = 0, err_host_old = 0;
    for (int i = 0; i < 216; ++i) scale = std::fmax(scale, std::fabs((double)b[i]));
    for (int i = 0; i < 216; ++i) err_old = std::fmax(err_old, std::fabs((double)a[i] - b[i]) / scale);
    const bool repro = memcmp(a, a2, sizeof(a)) == 0;
    if (s.host) {
      for (int c = 0; c < 8; ++c)
        for (int kz = 0; kz < 3; ++kz)
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              double sum = 0;
              for (int bb = 0; bb < s.B; ++bb)
                for (int z = 0; z < s.D; ++z)
                  for (int y = 0; y < s.H; ++y)
                    for (int xx = 0; xx < s.W; ++xx) {
                      const int iz = z + kz - 1, iy = y + ky - 1, ix = xx + kx - 1;
                      if (iz < 0 || iz >= s.D || iy < 0 || iy >= s.H || ix < 0 || ix >= s.W) continue;
                      sum += (double)g[(size_t)bb * n + ((size_t)z * s.H + y) * s.W + xx] * x[((size_t)bb * 8 + c) * n + ((size_t)iz * s.H + iy) * s.W + ix];
                    }
              const int i = c * 27 + kz * 9 + ky * 3 + kx;
              err_host_new = std::fmax(err_host_new, std::fabs(sum - a[i]) / scale);
              err_host_old = std::fmax(err_host_old, std::fabs(sum - b[i]) / scale);
            }
    }
    float ms_new = 0, ms_old = 0;
    const int reps = 20;
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) casmvs_prob_wgrad_f32(dx, dg, gw_new, ws_new, s.B, s.D, s.H, s.W, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms_new, e0, e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) casmvs_conv_wgrad_f32(CASMVS_CONV_S1, dx, dg, gw_old, ws_old, s.B, 8, 1, s.D, s.H, s.W, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms_old, e0, e1);
    const bool ok = err_old < 2e-5 && repro && (!s.host || (err_host_new < 1e-5 && err_host_old < 1e-5));
    all_ok &= ok;
    printf("B=%d D=%d H=%d W=%d: |new - generic| / max|gw| = %.2e, reproducible %s", s.B, s.D, s.H, s.W, err_old, repro ? "yes" : "NO");
    if (s.host) printf(", vs float64 host loop: new %.2e generic %.2e", err_host_new, err_host_old);
    printf("; %.1f us (new) vs %.1f us (generic)  %s\n", ms_new * 1e3 / reps, ms_old * 1e3 / reps, ok ? "ok" : "FAILED");
    hipFree(dx); hipFree(dg); hipFree(gw_new); hipFree(gw_old); hipFree(ws_new); hipFree(ws_old);
  }
  printf(all_ok ? "ALL OK\n" : "FAILURES\n");
  return all_ok ? 0 : 1;
}
