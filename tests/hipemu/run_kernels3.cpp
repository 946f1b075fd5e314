// As run_kernels2.cpp, for CostRegNet's stride-2 layers on the f16 matrix cores (conv1 / conv3: conv_s2_splitf16.hip): the kernel's own source against the
// layer in float64 - ragged volumes (odd sizes along z and y, rows shorter and longer than a patch), several z segments, persistent workgroups that walk
// several items.  Written in round 4 with this emulation as its first test.
#include "support.h"

#include "conv_s2_splitf16.hip"

static double conv_s2_check(int cin, int cout, int B, int D, int H, int W) {
  const size_t n = (size_t)D * H * W;
  const int Do = (D - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t no = (size_t)Do * Ho * Wo;
  std::vector<float> x((size_t)B * cin * n), w((size_t)cout * cin * 27), sc(cout), sh(cout);
  for (auto &v : x) v = rnd() * 3.0f + 0.4f;
  for (size_t i = 0; i < x.size(); i += 97) x[i] *= 64.0f;   // units of very different magnitude
  for (auto &v : w) v = rnd() * 0.15f;
  for (int i = 0; i < cout; ++i) { sc[i] = 0.5f + 0.02f * i; sh[i] = 0.01f * (i - 4); }
  const size_t pb = casmvs_conv_s2_splitf16_packed_bytes(cin, cout);
  unsigned char *pk = (unsigned char *)std::aligned_alloc(256, (pb + 255) & ~(size_t)255);
  if (casmvs_conv_s2_splitf16_pack(cin, cout, w.data(), sc.data(), sh.data(), pk)) { printf("conv_s2 pack: %s\n", casmvs_last_error()); return 1e9; }
  float *xa = (float *)std::aligned_alloc(256, (x.size() * 4 + 255) & ~(size_t)255), *ya = (float *)std::aligned_alloc(256, ((size_t)B * cout * no * 4 + 255) & ~(size_t)255);
  std::memcpy(xa, x.data(), x.size() * 4);
  for (size_t i = 0; i < (size_t)B * cout * no; ++i) ya[i] = NAN;
  if (casmvs_conv_s2_splitf16_forward_f32(pk, xa, ya, B, cin, cout, D, H, W, 0.01f, nullptr)) { printf("conv_s2: %s\n", casmvs_last_error()); return 1e9; }
  double err = 0, range = 0;
  for (int b = 0; b < B; ++b)
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
                    if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                    acc += (double)w[((size_t)co * cin + ci) * 27 + kz * 9 + ky * 3 + kx] * x[((size_t)b * cin + ci) * n + ((size_t)iz * H + iy) * W + ix];
                  }
            const double v = lrelu(acc * sc[co] + sh[co]);
            const float got = ya[((size_t)b * cout + co) * no + ((size_t)z * Ho + yy) * Wo + xx];
            range = std::fmax(range, std::fabs(v));
            err = std::fmax(err, std::isfinite(got) ? std::fabs(v - got) : 1e30);
          }
  std::free(pk); std::free(xa); std::free(ya);
  printf("conv_s2    %d -> %d B=%d %dx%dx%d: max error / range = %.2e\n", cin, cout, B, D, H, W, err / range);
  return err / range;
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem_raw;
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  // the three prefetch depths (register sets in flight); production: depth 2 for both layers
  g_s2_emu_depth = 2;
  if (all || quick || which == "conv_s2") { take(conv_s2_check(8, 16, 1, 6, 14, 72)); take(conv_s2_check(16, 32, 1, 4, 10, 40)); }
  if (all || which == "conv_s2") { take(conv_s2_check(8, 16, 1, 16, 4, 12)); take(conv_s2_check(16, 32, 2, 10, 12, 72)); take(conv_s2_check(8, 16, 2, 5, 27, 132)); }
  g_s2_emu_depth = 1;
  if (all || which == "conv_s2") { take(conv_s2_check(8, 16, 1, 7, 9, 68)); take(conv_s2_check(16, 32, 1, 9, 13, 8)); }
  g_s2_emu_depth = 3;
  if (all || which == "conv_s2") { take(conv_s2_check(16, 32, 1, 6, 14, 40)); take(conv_s2_check(8, 16, 1, 1, 2, 4)); }
  g_s2_emu_depth = 2;
  if (which == "streams") take(conv_s2_check(8, 16, 1, 8, 24, 128));
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
