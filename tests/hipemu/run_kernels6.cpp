// As run_kernels.cpp, for the production FPN tail on the f16 matrix cores (csrc/fpn_fused_sf.hip: feat0 = smooth0(lat0(conv0) + upsample2x(feat1')) as one
// 3x3 convolution over 40 channels with the bilinear interpolation inside its staging; GPU-validated): a regression test of its device code that needs no
// GPU, its barriers under ThreadSanitizer and its LDS traffic under tools/lds_bank_profile.py.  Against the definition in float64 (ATen's align_corners rule).
#include "support.h"

#include "fpn_fused_sf.hip"

static double fpn_check(int N, int H, int W) {
  const int h = H / 2, w = W / 2;
  const size_t hw = (size_t)H * W, hw1 = (size_t)h * w;
  std::vector<float> c0((size_t)N * 8 * hw), f1((size_t)N * 32 * hw1), w40(8 * 40 * 9), b9(9 * 8);
  for (auto &v : c0) v = rnd() * 2.0f + 0.3f;
  for (auto &v : f1) v = rnd() * 1.5f;
  for (auto &v : w40) v = rnd() * 0.2f;
  for (auto &v : b9) v = rnd() * 0.1f;
  auto dup = [](const void *src, size_t bytes) {
    void *p = std::aligned_alloc(256, (bytes + 255) & ~(size_t)255);
    std::memcpy(p, src, bytes);
    return p;
  };
  unsigned char *pk = (unsigned char *)std::aligned_alloc(256, (casmvs_fpn_tail0_splitf16_packed_bytes() + 255) & ~(size_t)255);
  if (casmvs_fpn_tail0_splitf16_pack(w40.data(), pk)) { printf("fpn pack: %s\n", casmvs_last_error()); return 1e9; }
  float *c0a = (float *)dup(c0.data(), c0.size() * 4), *f1a = (float *)dup(f1.data(), f1.size() * 4), *b9a = (float *)dup(b9.data(), b9.size() * 4);
  std::vector<float> nanv((size_t)N * 8 * hw, NAN);
  float *out = (float *)dup(nanv.data(), nanv.size() * 4), *out_nhwc = (float *)dup(nanv.data(), nanv.size() * 4);
  if (casmvs_fpn_tail0_splitf16_f32(pk, b9a, c0a, f1a, out, out_nhwc, N, H, W, nullptr)) { printf("fpn_tail0: %s\n", casmvs_last_error()); return 1e9; }
  // upsample2x, align_corners = True: source coordinate = destination * (in - 1) / (out - 1)
  std::vector<double> up((size_t)N * 32 * hw);
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < 32; ++c)
      for (int y = 0; y < H; ++y) {
        const double sy = H > 1 ? (double)y * (h - 1) / (H - 1) : 0.0;
        const int y0 = (int)sy, y1 = std::min(y0 + 1, h - 1);
        const double fy = sy - y0;
        for (int x = 0; x < W; ++x) {
          const double sx = W > 1 ? (double)x * (w - 1) / (W - 1) : 0.0;
          const int x0 = (int)sx, x1 = std::min(x0 + 1, w - 1);
          const double fx = sx - x0;
          const float *p = f1.data() + ((size_t)n * 32 + c) * hw1;
          up[((size_t)n * 32 + c) * hw + (size_t)y * W + x] = (1 - fy) * ((1 - fx) * p[(size_t)y0 * w + x0] + fx * p[(size_t)y0 * w + x1]) +
                                                               fy * ((1 - fx) * p[(size_t)y1 * w + x0] + fx * p[(size_t)y1 * w + x1]);
        }
      }
  double err = 0, range = 0, err2 = 0;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < 8; ++co)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          const int rc = y == 0 ? 0 : (y == H - 1 ? 2 : 1), cc = x == 0 ? 0 : (x == W - 1 ? 2 : 1);
          double acc = b9[(size_t)(rc * 3 + cc) * 8 + co];
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int iy = y + ky - 1, ix = x + kx - 1;
              if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
              for (int c = 0; c < 8; ++c) acc += (double)w40[(((size_t)co * 40 + c) * 3 + ky) * 3 + kx] * c0[((size_t)n * 8 + c) * hw + (size_t)iy * W + ix];
              for (int c = 0; c < 32; ++c) acc += (double)w40[(((size_t)co * 40 + 8 + c) * 3 + ky) * 3 + kx] * up[((size_t)n * 32 + c) * hw + (size_t)iy * W + ix];
            }
          const float got = out[((size_t)n * 8 + co) * hw + (size_t)y * W + x], got2 = out_nhwc[((size_t)n * hw + (size_t)y * W + x) * 8 + co];
          range = std::fmax(range, std::fabs(acc));
          err = std::fmax(err, std::isfinite(got) ? std::fabs(acc - got) : 1e30);
          err2 = std::fmax(err2, got2 == got ? 0.0 : 1e30);
        }
  std::free(pk); std::free(c0a); std::free(f1a); std::free(b9a); std::free(out); std::free(out_nhwc);
  printf("fpn_tail0  N=%d %dx%d: max error / range = %.2e (pixel-major copy %s)\n", N, H, W, err / range, err2 == 0 ? "equal" : "DIFFERENT");
  return std::fmax(err / range, err2);
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem_raw;
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) take(fpn_check(1, 22, 36));     // two tiles in y (20 + 2 rows) and in x (32 + 4)
  if (all) take(fpn_check(2, 40, 64));
  if (which == "streams") take(fpn_check(1, 40, 128));
  // the bound of tests/test_gpu_parity.py for this kernel: the interpolation weights are float32 (ATen's rule), ~1e-7 of a source coordinate of up to W / 2
  printf(worst < 1.2e-5 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 1.2e-5 ? 0 : 1;
}
