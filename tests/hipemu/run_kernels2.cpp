// As run_kernels.cpp, for kernels whose helper names collide with conv0_splitf16.hip's in one translation unit: the channel-inner split-f16 kernels of
// CostRegNet (conv2 / conv4 / conv6: conv_ci_splitf16.hip) and FeatureNet (conv2d_ci_splitf16.hip).  Both are PRODUCTION kernels validated on the MI355X: run
// here they are regression tests of the device code that need no GPU.
#include "support.h"

#include "conv_ci_splitf16.hip"
#include "conv2d_ci_splitf16.hip"

static double conv_ci_check(int c, int B, int D, int H, int W) {
  const size_t n = (size_t)D * H * W;
  std::vector<float> x((size_t)B * c * n), w((size_t)c * c * 27), sc(c), sh(c);
  for (auto &v : x) v = rnd() * 3.0f + 0.4f;
  for (auto &v : w) v = rnd() * 0.15f;
  for (int i = 0; i < c; ++i) { sc[i] = 0.5f + 0.02f * i; sh[i] = 0.01f * (i - 4); }
  const size_t pb = casmvs_conv_ci_splitf16_packed_bytes(c, c);
  unsigned char *pk = (unsigned char *)std::aligned_alloc(256, (pb + 255) & ~(size_t)255);
  if (casmvs_conv_ci_splitf16_pack(c, c, w.data(), sc.data(), sh.data(), pk)) { printf("conv_ci pack: %s\n", casmvs_last_error()); return 1e9; }
  float *xa = (float *)std::aligned_alloc(256, (x.size() * 4 + 63) & ~(size_t)63), *ya = (float *)std::aligned_alloc(256, (x.size() * 4 + 255) & ~(size_t)255);
  std::memcpy(xa, x.data(), x.size() * 4);
  for (size_t i = 0; i < x.size(); ++i) ya[i] = NAN;
  if (casmvs_conv_ci_splitf16_forward_f32(pk, xa, ya, B, c, c, D, H, W, 0.01f, nullptr)) { printf("conv_ci: %s\n", casmvs_last_error()); return 1e9; }
  double err = 0, range = 0;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < c; ++co)
      for (int z = 0; z < D; ++z)
        for (int yy = 0; yy < H; ++yy)
          for (int xx = 0; xx < W; ++xx) {
            double acc = 0;
            for (int ci = 0; ci < c; ++ci)
              for (int kz = 0; kz < 3; ++kz)
                for (int ky = 0; ky < 3; ++ky)
                  for (int kx = 0; kx < 3; ++kx) {
                    const int iz = z + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                    if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                    acc += (double)w[((size_t)co * c + ci) * 27 + kz * 9 + ky * 3 + kx] * x[((size_t)b * c + ci) * n + ((size_t)iz * H + iy) * W + ix];
                  }
            const double v = lrelu(acc * sc[co] + sh[co]);
            const float got = ya[((size_t)b * c + co) * n + ((size_t)z * H + yy) * W + xx];
            range = std::fmax(range, std::fabs(v));
            err = std::fmax(err, std::isfinite(got) ? std::fabs(v - got) : 1e30);
          }
  std::free(pk); std::free(xa); std::free(ya);
  printf("conv_ci    %d -> %d B=%d %dx%dx%d: max error / range = %.2e\n", c, c, B, D, H, W, err / range);
  return err / range;
}

static double conv2d_ci_check(int cin, int cout, int N, int H, int W) {
  const size_t hw = (size_t)H * W;
  std::vector<float> x((size_t)N * cin * hw), w((size_t)cout * cin * 9), sc(cout), sh(cout);
  for (auto &v : x) v = rnd() * 3.0f + 0.4f;
  for (auto &v : w) v = rnd() * 0.15f;
  for (int i = 0; i < cout; ++i) { sc[i] = 0.5f + 0.02f * i; sh[i] = 0.01f * (i - 4); }
  const size_t pb = casmvs_conv2d_ci_splitf16_packed_bytes(cin, cout);
  unsigned char *pk = (unsigned char *)std::aligned_alloc(256, (pb + 255) & ~(size_t)255);
  if (casmvs_conv2d_ci_splitf16_pack(cin, cout, w.data(), sc.data(), sh.data(), pk)) { printf("conv2d_ci pack: %s\n", casmvs_last_error()); return 1e9; }
  const size_t no = (size_t)N * cout * hw;
  float *xa = (float *)std::aligned_alloc(256, (x.size() * 4 + 63) & ~(size_t)63), *ya = (float *)std::aligned_alloc(256, (no * 4 + 255) & ~(size_t)255),
        *yb = (float *)std::aligned_alloc(256, (no * 4 + 255) & ~(size_t)255);
  std::memcpy(xa, x.data(), x.size() * 4);
  for (size_t i = 0; i < no; ++i) ya[i] = yb[i] = NAN;
  if (casmvs_conv2d_ci_splitf16_forward_f32(pk, xa, ya, yb, N, cin, cout, H, W, 0.01f, nullptr)) { printf("conv2d_ci: %s\n", casmvs_last_error()); return 1e9; }
  double err = 0, range = 0;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < cout; ++co)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {
          double acc = 0;
          for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                acc += (double)w[((size_t)co * cin + ci) * 9 + ky * 3 + kx] * x[((size_t)n * cin + ci) * hw + (size_t)iy * W + ix];
              }
          const double v = lrelu(acc * sc[co] + sh[co]);
          const float got = ya[((size_t)n * cout + co) * hw + (size_t)yy * W + xx], got2 = yb[((size_t)n * hw + (size_t)yy * W + xx) * cout + co];   // NCHW and pixel-major copy
          range = std::fmax(range, std::fabs(v));
          err = std::fmax(err, std::isfinite(got) && got == got2 ? std::fabs(v - got) : 1e30);
        }
  std::free(pk); std::free(xa); std::free(ya); std::free(yb);
  printf("conv2d_ci  %d -> %d N=%d %dx%d: max error / range = %.2e (pixel-major copy equal)\n", cin, cout, N, H, W, err / range);
  return err / range;
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem_raw;
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick || which == "conv_ci") take(conv_ci_check(16, 1, 5, 6, 18));
  if (all || which == "conv_ci") { take(conv_ci_check(32, 1, 2, 9, 16)); take(conv_ci_check(16, 2, 4, 4, 34)); }
  if (which == "streams") {   // interior-dominated problems with cache-line-aligned rows (tools/lds_bank_profile.py: request streams)
    take(conv_ci_check(16, 1, 8, 16, 64));
    take(conv2d_ci_check(16, 16, 1, 32, 128));
  }
  if (all || quick || which == "conv2d_ci") take(conv2d_ci_check(32, 16, 1, 18, 20));
  if (all || which == "conv2d_ci") { take(conv2d_ci_check(16, 16, 2, 17, 34)); take(conv2d_ci_check(32, 32, 1, 16, 18)); }
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
