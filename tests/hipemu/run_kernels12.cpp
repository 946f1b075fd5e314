// As run_kernels11.cpp, for FeatureNet.conv0 as ONE kernel with both layers on the f16 matrix cores (fnet_conv0_mm.hip, round 6): the kernel's own source
// against ConvBnReLU(3, 8, 3) -> ConvBnReLU(8, 8, 3) in float64 - image borders inside a tile, ragged images (several 20 x 30 tiles in y and x, widths that
// are not multiples of 30, heights below one tile), persistent workgroups that walk several tiles, several images.
#include "support.h"

#include "fnet_conv0_mm.hip"

static double fnet0_check(int N, int H, int W, float amp) {
  const size_t hw = (size_t)H * W;
  std::vector<float> x((size_t)N * 3 * hw), w0(8 * 3 * 9), w1(8 * 8 * 9), sc0(8), sh0(8), sc1(8), sh1(8);
  for (auto &v : x) v = (rnd() * 3.0f + 0.4f) * amp;
  for (size_t i = 0; i < x.size(); i += 101) x[i] *= 64.0f;
  for (auto &v : w0) v = rnd() * 0.3f;
  for (auto &v : w1) v = rnd() * 0.2f;
  for (int i = 0; i < 8; ++i) { sc0[i] = 0.5f + 0.05f * i; sh0[i] = 0.02f * (i - 4) * amp; sc1[i] = 0.8f - 0.03f * i; sh1[i] = 0.01f * (3 - i) * amp; }
  const size_t pb = casmvs_fnet_conv0_mm_packed_bytes();
  unsigned char *pk = (unsigned char *)std::aligned_alloc(256, (pb + 255) & ~(size_t)255);
  if (casmvs_fnet_conv0_mm_pack(w0.data(), sc0.data(), sh0.data(), w1.data(), sc1.data(), sh1.data(), pk)) { printf("fnet_conv0_mm pack: %s\n", casmvs_last_error()); return 1e9; }
  float *xa = (float *)std::aligned_alloc(256, (x.size() * 4 + 255) & ~(size_t)255), *ya = (float *)std::aligned_alloc(256, ((size_t)N * 8 * hw * 4 + 255) & ~(size_t)255);
  std::memcpy(xa, x.data(), x.size() * 4);
  for (size_t i = 0; i < (size_t)N * 8 * hw; ++i) ya[i] = NAN;
  if (casmvs_fnet_conv0_mm_f32(pk, xa, ya, N, H, W, 0.01f, nullptr)) { printf("fnet_conv0_mm: %s\n", casmvs_last_error()); return 1e9; }
  std::vector<double> mid((size_t)N * 8 * hw);
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < 8; ++co)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {
          double acc = 0;
          for (int ci = 0; ci < 3; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                acc += (double)w0[((co * 3 + ci) * 3 + ky) * 3 + kx] * x[((size_t)n * 3 + ci) * hw + (size_t)iy * W + ix];
              }
          mid[((size_t)n * 8 + co) * hw + (size_t)yy * W + xx] = lrelu(acc * sc0[co] + sh0[co]);
        }
  double err = 0, range = 0;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < 8; ++co)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {
          double acc = 0;
          for (int ci = 0; ci < 8; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                acc += (double)w1[((co * 8 + ci) * 3 + ky) * 3 + kx] * mid[((size_t)n * 8 + ci) * hw + (size_t)iy * W + ix];
              }
          const double v = lrelu(acc * sc1[co] + sh1[co]);
          const float got = ya[((size_t)n * 8 + co) * hw + (size_t)yy * W + xx];
          range = std::fmax(range, std::fabs(v));
          err = std::fmax(err, std::isfinite(got) ? std::fabs(v - got) : 1e30);
        }
  std::free(pk); std::free(xa); std::free(ya);
  printf("fnet_conv0_mm N=%d %dx%d amplitude %.0e: max error / range = %.2e\n", N, H, W, amp, err / range);
  return err / range;
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem_raw;
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) { take(fnet0_check(1, 22, 64, 1.0f)); take(fnet0_check(2, 6, 8, 1.0f)); }   // two tiles in y, three in x (the last 4 pixels wide); images below one tile
  if (all) { take(fnet0_check(3, 44, 92, 1.0f)); take(fnet0_check(1, 2, 2, 1.0f)); take(fnet0_check(1, 20, 30, 1e-20f)); take(fnet0_check(1, 24, 36, 3e4f)); }
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
