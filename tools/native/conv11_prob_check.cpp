// First GPU test of casmvs_conv11_prob_zfused_f32 (csrc/conv11_prob_zfused.hip: conv11 + `prob` + softmax regression walking the depth axis; written in round 4
// with the CPU emulation as its only test), torch-free: against the two kernels it replaces - casmvs_deconv11_splitf16_forward_f32 followed by
// casmvs_prob_regress_f32 - on ragged small shapes and on the cascade levels' shapes, twice for run-to-run bit stability, with the time of both paths under
// dirtied caches.   conv11_prob_check [batch]
//   hipcc -O2 tools/native/conv11_prob_check.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/conv11_prob_check
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
  std::vector<float> w11(16 * 8 * 27), sc(8), sh(8), wp(8 * 27), pbias(1, 0.125f);
  for (auto &v : w11) v = rnd() * 0.2f;
  for (int c = 0; c < 8; ++c) { sc[c] = 0.5f + 0.05f * c; sh[c] = 0.03f * (c - 4); }
  for (auto &v : wp) v = rnd() * 0.3f;
  std::vector<unsigned char> dpk(casmvs_deconv11_splitf16_packed_bytes());
  if (casmvs_deconv11_splitf16_pack(w11.data(), sc.data(), sh.data(), dpk.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
  std::vector<float> ppk(casmvs_conv3d_packed_floats(CASMVS_CONV_S1, 8, 1));
  if (ppk.empty() || casmvs_conv3d_pack_f32(CASMVS_CONV_S1, 8, 1, wp.data(), nullptr, pbias.data(), ppk.data())) { printf("pack prob: %s\n", casmvs_last_error()); return 3; }
  void *ddpk;
  float *dppk;
  hipMalloc(&ddpk, dpk.size()); hipMalloc(&dppk, ppk.size() * 4);
  hipMemcpy(ddpk, dpk.data(), dpk.size(), hipMemcpyHostToDevice);
  hipMemcpy(dppk, ppk.data(), ppk.size() * 4, hipMemcpyHostToDevice);
  struct Shape { int B, Di, Hi, Wi; };
  const Shape shapes[] = {{1, 2, 5, 34}, {2, 4, 9, 32}, {1, 3, 8, 62}, {1, 1, 1, 2}, {batch, 24, 64, 80}, {batch, 16, 128, 160}, {batch, 4, 256, 320}};
  bool all_ok = true;
  for (const Shape &s : shapes) {
    const int D = 2 * s.Di, H = 2 * s.Hi, W = 2 * s.Wi;
    const size_t ni = (size_t)s.Di * s.Hi * s.Wi, no = (size_t)D * H * W, hw = (size_t)H * W;
    const size_t nin = (size_t)s.B * 16 * ni, nsk = (size_t)s.B * 8 * no, nvol = (size_t)s.B * no, npix = (size_t)s.B * hw;
    std::vector<float> x(nin), sk(nsk), dv(nvol);
    for (auto &v : x) v = rnd() * 2.0f + 0.2f;
    for (size_t i = 0; i < nin; i += 509) x[i] *= 40.0f;
    for (auto &v : sk) v = rnd();
    for (int b = 0; b < s.B; ++b)
      for (int z = 0; z < D; ++z)
        for (size_t p = 0; p < hw; ++p) dv[((size_t)b * D + z) * hw + p] = 425.0f + 2.5f * z + 0.01f * (float)(p % 7);
    float *dx, *dsk, *ddv, *du11, *dcost[2], *ddepth[2], *dconf[2];
    int32_t *didx[2];
    hipMalloc(&dx, nin * 4); hipMalloc(&dsk, nsk * 4); hipMalloc(&ddv, nvol * 4); hipMalloc(&du11, nsk * 4);
    for (int k = 0; k < 2; ++k) { hipMalloc(&dcost[k], nvol * 4); hipMalloc(&ddepth[k], npix * 4); hipMalloc(&dconf[k], npix * 4); hipMalloc(&didx[k], npix * 4); }
    hipMemcpy(dx, x.data(), nin * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsk, sk.data(), nsk * 4, hipMemcpyHostToDevice);
    hipMemcpy(ddv, dv.data(), nvol * 4, hipMemcpyHostToDevice);
    auto run = [&](int k) {
      if (k == 0) {
        if (int rc = casmvs_deconv11_splitf16_forward_f32(ddpk, dx, dsk, du11, s.B, s.Di, s.Hi, s.Wi, 0.01f, st)) return rc;
        return casmvs_prob_regress_f32(dppk, du11, ddv, dcost[0], ddepth[0], dconf[0], didx[0], s.B, 8, D, H, W, 1.0f, 0, st);
      }
      return casmvs_conv11_prob_zfused_f32(ddpk, dppk, dx, dsk, ddv, dcost[1], ddepth[1], dconf[1], didx[1], s.B, s.Di, s.Hi, s.Wi, 0.01f, 1.0f, st);
    };
    std::vector<float> cost[2], depth[2], conf[2], again(npix);
    std::vector<int32_t> idx[2];
    double us[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
      hipMemset(dcost[k], 0xff, nvol * 4); hipMemset(ddepth[k], 0xff, npix * 4);
      if (run(k)) { printf("forward %d: %s\n", k, casmvs_last_error()); return 3; }
      if (hipStreamSynchronize(st) != hipSuccess) { printf("kernel %d failed: %s\n", k, hipGetErrorString(hipGetLastError())); return 4; }
      cost[k].resize(nvol); depth[k].resize(npix); conf[k].resize(npix); idx[k].resize(npix);
      hipMemcpy(cost[k].data(), dcost[k], nvol * 4, hipMemcpyDeviceToHost);
      hipMemcpy(depth[k].data(), ddepth[k], npix * 4, hipMemcpyDeviceToHost);
      hipMemcpy(conf[k].data(), dconf[k], npix * 4, hipMemcpyDeviceToHost);
      hipMemcpy(idx[k].data(), didx[k], npix * 4, hipMemcpyDeviceToHost);
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
    hipMemcpy(again.data(), ddepth[1], npix * 4, hipMemcpyDeviceToHost);
    const bool stable = memcmp(again.data(), depth[1].data(), npix * 4) == 0;
    double crange = 0, cdiff = 0, ddiff = 0, fdiff = 0;
    size_t nan = 0, idiff = 0;
    for (size_t i = 0; i < nvol; ++i) {
      crange = std::fmax(crange, std::fabs((double)cost[0][i]));
      if (!std::isfinite(cost[1][i])) ++nan;
      cdiff = std::fmax(cdiff, std::fabs((double)cost[0][i] - cost[1][i]));
    }
    for (size_t i = 0; i < npix; ++i) {
      if (!std::isfinite(depth[1][i]) || !std::isfinite(conf[1][i])) ++nan;
      ddiff = std::fmax(ddiff, std::fabs((double)depth[0][i] - depth[1][i]) / std::fabs((double)depth[0][i]));
      fdiff = std::fmax(fdiff, std::fabs((double)conf[0][i] - conf[1][i]));
      idiff += idx[0][i] != idx[1][i];
    }
    printf("B=%d in %dx%dx%d: deconv11 + prob_regress %.1f us, fused %.1f us (x%.3f); cost max |diff| / range = %.2e, depth rel %.2e, confidence %.2e, indices differing %zu of %zu, "
           "non-finite %zu, repeat run %s", s.B, s.Di, s.Hi, s.Wi, us[0], us[1], us[0] / us[1], cdiff / crange, ddiff, fdiff, idiff, npix, nan, stable ? "equal" : "DIFFERENT");
    const bool ok = nan == 0 && stable && cdiff / crange < 3e-6 && ddiff < 1e-4 && idiff <= npix / 1000 + 1;   // (indices: trunc() boundaries under other roundings)
    printf("  %s\n", ok ? "ok" : "FAILED");
    all_ok = all_ok && ok;
    hipFree(dx); hipFree(dsk); hipFree(ddv); hipFree(du11);
    for (int k = 0; k < 2; ++k) { hipFree(dcost[k]); hipFree(ddepth[k]); hipFree(dconf[k]); hipFree(didx[k]); }
  }
  printf(all_ok ? "ALL OK\n" : "FAILED\n");
  return all_ok ? 0 : 1;
}
