// As run_kernels2.cpp, for FeatureNet's stride-2 layers on the f16 matrix cores (conv1.0 / conv2.0: conv2d_k5s2_splitf16.hip): the kernel's own source against
// Conv2d k5 s2 p2 + ABN + leaky-relu in float64 - ragged images (borders inside a tile, several tiles), persistent workgroups that walk several units.
// Written in round 4 with this emulation as its first test.
#include "support.h"

#include "conv2d_k5s2_splitf16.hip"

static double k5s2_check(int cin, int cout, int N, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t hw = (size_t)H * W, ohw = (size_t)Ho * Wo;
  std::vector<float> x((size_t)N * cin * hw), w((size_t)cout * cin * 25), sc(cout), sh(cout);
  for (auto &v : x) v = rnd() * 3.0f + 0.4f;
  for (size_t i = 0; i < x.size(); i += 101) x[i] *= 64.0f;
  for (auto &v : w) v = rnd() * 0.15f;
  for (int i = 0; i < cout; ++i) { sc[i] = 0.5f + 0.02f * i; sh[i] = 0.01f * (i - 4); }
  const size_t pb = casmvs_conv2d_k5s2_splitf16_packed_bytes(cin, cout);
  unsigned char *pk = (unsigned char *)std::aligned_alloc(256, (pb + 255) & ~(size_t)255);
  if (casmvs_conv2d_k5s2_splitf16_pack(cin, cout, w.data(), sc.data(), sh.data(), pk)) { printf("k5s2 pack: %s\n", casmvs_last_error()); return 1e9; }
  float *xa = (float *)std::aligned_alloc(256, (x.size() * 4 + 255) & ~(size_t)255), *ya = (float *)std::aligned_alloc(256, ((size_t)N * cout * ohw * 4 + 255) & ~(size_t)255);
  std::memcpy(xa, x.data(), x.size() * 4);
  for (size_t i = 0; i < (size_t)N * cout * ohw; ++i) ya[i] = NAN;
  if (casmvs_conv2d_k5s2_splitf16_forward_f32(pk, xa, ya, N, cin, cout, H, W, 0.01f, nullptr)) { printf("k5s2: %s\n", casmvs_last_error()); return 1e9; }
  double err = 0, range = 0;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < cout; ++co)
      for (int yy = 0; yy < Ho; ++yy)
        for (int xx = 0; xx < Wo; ++xx) {
          double acc = 0;
          for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < 5; ++ky)
              for (int kx = 0; kx < 5; ++kx) {
                const int iy = 2 * yy + ky - 2, ix = 2 * xx + kx - 2;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                acc += (double)w[((size_t)co * cin + ci) * 25 + ky * 5 + kx] * x[((size_t)n * cin + ci) * hw + (size_t)iy * W + ix];
              }
          const double v = lrelu(acc * sc[co] + sh[co]);
          const float got = ya[((size_t)n * cout + co) * ohw + (size_t)yy * Wo + xx];
          range = std::fmax(range, std::fabs(v));
          err = std::fmax(err, std::isfinite(got) ? std::fabs(v - got) : 1e30);
        }
  std::free(pk); std::free(xa); std::free(ya);
  printf("conv2d_k5s2 %d -> %d N=%d %dx%d: max error / range = %.2e\n", cin, cout, N, H, W, err / range);
  return err / range;
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem_raw;
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) { take(k5s2_check(8, 16, 1, 20, 72)); take(k5s2_check(16, 32, 1, 18, 40)); }   // two tiles in y and x; one chunk / two chunks
  if (all) { take(k5s2_check(8, 16, 3, 34, 136)); take(k5s2_check(16, 32, 2, 6, 8)); take(k5s2_check(8, 16, 1, 2, 4)); }
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
