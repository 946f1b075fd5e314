// First GPU test of casmvs_conv11_prob_regress_f32 (csrc/conv11_prob_fused.hip: conv11 + skip + `prob` + softmax regression as one kernel; written without a
// GPU run, correct on the CPU under tests/hipemu), torch-free: against the two launches it replaces - casmvs_conv3d_forward_f32(CASMVS_CONV_T2, 16 -> 8, skip) then
// casmvs_prob_regress_f32 - on ragged small shapes and on the cascade levels' shapes: cost / depth / confidence differences, run-to-run bit stability, and
// the time of each form under dirtied caches.   conv11_prob_check [batch]
//   hipcc -O2 tools/native/conv11_prob_check.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/conv11_prob_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "casmvs.h"

static uint32_t g_rng = 123456789u;
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
  std::vector<float> w11(16 * 8 * 27), sc(8), sh(8), wp(8 * 27), bias(1, 0.125f);
  for (auto &v : w11) v = rnd() * 0.2f;
  for (auto &v : wp) v = rnd() * 0.3f;
  for (int c = 0; c < 8; ++c) { sc[c] = 0.5f + 0.1f * c; sh[c] = 0.05f * (c - 4); }
  std::vector<unsigned char> dimg(casmvs_deconv11_splitf16_packed_bytes());
  std::vector<float> p11(casmvs_conv3d_packed_floats(CASMVS_CONV_T2, 16, 8)), pp(casmvs_conv3d_packed_floats(CASMVS_CONV_S1, 8, 1));
  if (casmvs_deconv11_splitf16_pack(w11.data(), sc.data(), sh.data(), dimg.data()) || casmvs_conv3d_pack_f32(CASMVS_CONV_T2, 16, 8, w11.data(), sc.data(), sh.data(), p11.data()) ||
      casmvs_conv3d_pack_f32(CASMVS_CONV_S1, 8, 1, wp.data(), nullptr, bias.data(), pp.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
  void *ddimg;
  float *dp11, *dpp;
  hipMalloc(&ddimg, dimg.size()); hipMalloc(&dp11, p11.size() * 4); hipMalloc(&dpp, pp.size() * 4);
  hipMemcpy(ddimg, dimg.data(), dimg.size(), hipMemcpyHostToDevice);
  hipMemcpy(dp11, p11.data(), p11.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dpp, pp.data(), pp.size() * 4, hipMemcpyHostToDevice);
  struct Shape { int B, D, H, W; };
  const Shape shapes[] = {{1, 8, 10, 68}, {2, 6, 18, 124}, {1, 16, 24, 36}, {batch, 48, 128, 160}, {batch, 32, 256, 320}, {batch, 8, 512, 640}};
  bool all_ok = true;
  for (const Shape &s : shapes) {
    const size_t ni = (size_t)(s.D / 2) * (s.H / 2) * (s.W / 2), no = (size_t)s.D * s.H * s.W, hw = (size_t)s.H * s.W;
    std::vector<float> u9((size_t)s.B * 16 * ni), skip((size_t)s.B * 8 * no), dv((size_t)s.B * no);
    for (auto &v : u9) v = rnd() * 2.0f + 0.2f;
    for (auto &v : skip) v = rnd();
    for (int b = 0; b < s.B; ++b)
      for (int z = 0; z < s.D; ++z)
        for (size_t p = 0; p < hw; ++p) dv[((size_t)b * s.D + z) * hw + p] = 425.0f + 2.5f * z + 0.01f * (float)(p % 7);
    float *du9, *dsk, *ddv, *dx, *cost[2], *depth[2], *conf[2];
    hipMalloc(&du9, u9.size() * 4); hipMalloc(&dsk, skip.size() * 4); hipMalloc(&ddv, dv.size() * 4); hipMalloc(&dx, skip.size() * 4);
    for (int k = 0; k < 2; ++k) { hipMalloc(&cost[k], (size_t)s.B * no * 4); hipMalloc(&depth[k], (size_t)s.B * hw * 4); hipMalloc(&conf[k], (size_t)s.B * hw * 4); }
    hipMemcpy(du9, u9.data(), u9.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsk, skip.data(), skip.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ddv, dv.data(), dv.size() * 4, hipMemcpyHostToDevice);
    auto run = [&](int k) {
      if (k) return casmvs_conv11_prob_regress_f32(ddimg, dpp, du9, dsk, ddv, cost[1], depth[1], conf[1], nullptr, s.B, s.D, s.H, s.W, 0.01f, 0, st);
      if (int rc = casmvs_conv3d_forward_f32(CASMVS_CONV_T2, dp11, du9, dsk, dx, s.B, 16, 8, s.D / 2, s.H / 2, s.W / 2, 0.01f, st)) return rc;
      return casmvs_prob_regress_f32(dpp, dx, ddv, cost[0], depth[0], conf[0], nullptr, s.B, 8, s.D, s.H, s.W, 1.0f, 0, st);
    };
    std::vector<float> hc[2], hd[2], hf[2], again((size_t)s.B * no);
    double us[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
      hipMemset(cost[k], 0xff, (size_t)s.B * no * 4); hipMemset(depth[k], 0xff, (size_t)s.B * hw * 4); hipMemset(conf[k], 0xff, (size_t)s.B * hw * 4);
      if (run(k)) { printf("forward %d: %s\n", k, casmvs_last_error()); return 3; }
      if (hipStreamSynchronize(st) != hipSuccess) { printf("kernel %d failed: %s\n", k, hipGetErrorString(hipGetLastError())); return 4; }
      hc[k].resize((size_t)s.B * no); hd[k].resize((size_t)s.B * hw); hf[k].resize((size_t)s.B * hw);
      hipMemcpy(hc[k].data(), cost[k], hc[k].size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hd[k].data(), depth[k], hd[k].size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hf[k].data(), conf[k], hf[k].size() * 4, hipMemcpyDeviceToHost);
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
    hipMemcpy(again.data(), cost[1], again.size() * 4, hipMemcpyDeviceToHost);
    const bool stable = memcmp(again.data(), hc[1].data(), again.size() * 4) == 0;
    double range = 0, diff = 0, ddiff = 0, cdiff = 0;
    size_t nan = 0;
    for (size_t i = 0; i < hc[0].size(); ++i) {
      range = std::fmax(range, std::fabs((double)hc[0][i]));
      if (!std::isfinite(hc[1][i])) ++nan;
      diff = std::fmax(diff, std::fabs((double)hc[0][i] - hc[1][i]));
    }
    for (size_t i = 0; i < hd[0].size(); ++i) {
      if (!std::isfinite(hd[1][i]) || !std::isfinite(hf[1][i])) ++nan;
      ddiff = std::fmax(ddiff, std::fabs((double)hd[0][i] - hd[1][i]) / std::fabs((double)hd[0][i]));
      cdiff = std::fmax(cdiff, std::fabs((double)hf[0][i] - hf[1][i]));
    }
    const bool ok = nan == 0 && stable && diff / range < 3e-6 && ddiff < 1e-4;   // (confidence may differ where the expected index sits on an integer)
    printf("B=%d %dx%dx%d: conv11 + prob %.1f us, fused %.1f us (x%.3f); cost max |diff| / range = %.2e, depth rel %.2e, confidence abs %.2e, non-finite %zu, repeat run %s  %s\n",
           s.B, s.D, s.H, s.W, us[0], us[1], us[0] / us[1], diff / range, ddiff, cdiff, nan, stable ? "equal" : "DIFFERENT", ok ? "ok" : "FAILED");
    all_ok &= ok;
    hipFree(du9); hipFree(dsk); hipFree(ddv); hipFree(dx);
    for (int k = 0; k < 2; ++k) { hipFree(cost[k]); hipFree(depth[k]); hipFree(conf[k]); }
  }
  printf(all_ok ? "ALL OK\n" : "FAILURES\n");
  return all_ok ? 0 : 1;
}
