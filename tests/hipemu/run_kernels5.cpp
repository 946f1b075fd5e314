// As run_kernels.cpp, for two GPU-validated kernels that keep their LDS in function-scope `__shared__` arrays (one `static` array on the host: workgroups run
// one after the other): the `prob` layer's weight gradient (csrc/prob_wgrad.hip: DPP row sums, LDS reduction over the 16 rows, a second kernel over the
// workgroups) against a float64 loop, and the two depth-fusion kernels (csrc/fusion.hip) against each other - every output bit-equal, as
// tools/native/fusion_check.cpp asserts on the GPU.
#define __shared__ static
#include "support.h"

#include "prob_wgrad.hip"
#include "fusion.hip"

static double wgrad_check(int B, int D, int H, int W) {
  const size_t n = (size_t)D * H * W;
  std::vector<float> x((size_t)B * 8 * n), g((size_t)B * n);
  for (auto &v : x) v = rnd();
  for (auto &v : g) v = rnd();
  auto dup = [](const std::vector<float> &v) {
    float *p = (float *)std::aligned_alloc(64, (v.size() * 4 + 63) & ~(size_t)63);
    std::memcpy(p, v.data(), v.size() * 4);
    return p;
  };
  float *xa = dup(x), *ga = dup(g);
  void *ws = std::aligned_alloc(64, casmvs_prob_wgrad_workspace_bytes(B, D, H, W));
  std::vector<float> got(216, NAN);
  if (casmvs_prob_wgrad_f32(xa, ga, got.data(), ws, B, D, H, W, nullptr)) {
    printf("prob_wgrad: %s\n", casmvs_last_error());
    return 1e9;
  }
  double err = 0, range = 0;
  for (int c = 0; c < 8; ++c)
    for (int kz = 0; kz < 3; ++kz)
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          double acc = 0;
          for (int b = 0; b < B; ++b)
            for (int z = 0; z < D; ++z)
              for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx) {
                  const int iz = z + kz - 1, iy = y + ky - 1, ix = xx + kx - 1;
                  if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                  acc += (double)g[(size_t)b * n + ((size_t)z * H + y) * W + xx] * x[((size_t)b * 8 + c) * n + ((size_t)iz * H + iy) * W + ix];
                }
          const float v = got[(size_t)c * 27 + kz * 9 + ky * 3 + kx];
          range = std::fmax(range, std::fabs(acc));
          err = std::fmax(err, std::isfinite(v) ? std::fabs(acc - v) : 1e30);
        }
  std::free(xa); std::free(ga); std::free(ws);
  printf("prob_wgrad B=%d %dx%dx%d: max error / largest gradient = %.2e\n", B, D, H, W, err / range);
  return err / range;
}

static double fusion_check(int H, int W, int S) {
  const size_t hw = (size_t)H * W;
  uint32_t rng = 12345u;
  auto rn = [&] { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  std::vector<float> depth_ref(hw), depth_src(hw * S), proba(hw / 16);
  std::vector<unsigned char> image_ref(hw * 3), image_src(hw * 3 * S);
  for (auto &v : depth_ref) v = 600.0f + 0.002f * (float)(rn() % 1000);
  for (auto &v : depth_src) v = 600.0f + 0.002f * (float)(rn() % 1000);
  for (auto &v : proba) v = (float)(rn() % 1000) / 1000.0f;
  for (auto &v : image_ref) v = (unsigned char)rn();
  for (auto &v : image_src) v = (unsigned char)rn();
  const float f = 2892.33f * W / 1600.0f;
  std::vector<float> r2s(S * 12, 0.0f), s2r(S * 12, 0.0f), r2w(12, 0.0f);
  for (int s = 0; s < S; ++s) {
    const float b = 150.0f * (float)(s + 1 - (S + 1) / 2.0f);   // baseline along x: some views leave the (small) image at the border columns
    for (int k = 0; k < 3; ++k) r2s[s * 12 + 5 * k] = s2r[s * 12 + 5 * k] = 1.0f;
    r2s[s * 12 + 3] = -f * b;
    s2r[s * 12 + 3] = f * b;
    r2s[s * 12 + 7] = 0.37f * f;
    s2r[s * 12 + 7] = -0.37f * f;
  }
  r2w[0] = 1.0f / f; r2w[2] = -(W / 2.0f) / f; r2w[5] = 1.0f / f; r2w[6] = -(H / 2.0f) / f; r2w[10] = 1.0f;
  struct Out {
    std::vector<float> depth, xyz, dreproj;
    std::vector<double> image;
    std::vector<int32_t> count;
    std::vector<unsigned char> mask, mgeo, is2r;
  } o[2];
  for (int which = 0; which < 2; ++which) {
    Out &q = o[which];
    q.depth.assign(hw, NAN); q.xyz.assign(hw * 3, NAN); q.dreproj.assign(hw * S, NAN); q.image.assign(hw * 3, NAN); q.count.assign(hw, -1);
    q.mask.assign(hw, 0xCD); q.mgeo.assign(hw * S, 0xCD); q.is2r.assign(hw * S * 3, 0xCD);
    auto fn = which ? casmvs_fuse_reference_view_paired : casmvs_fuse_reference_view;
    if (fn(depth_ref.data(), image_ref.data(), proba.data(), depth_src.data(), image_src.data(), r2s.data(), s2r.data(), r2w.data(), q.depth.data(), q.image.data(),
           q.count.data(), q.mask.data(), q.xyz.data(), q.mgeo.data(), q.dreproj.data(), q.is2r.data(), S, H, W, 0.5f, 2, nullptr)) {
      printf("fusion %d: %s\n", which, casmvs_last_error());
      return 1e9;
    }
  }
  auto diff = [](const auto &a, const auto &b) { return (double)(std::memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) != 0); };
  const double bad = diff(o[0].depth, o[1].depth) + diff(o[0].xyz, o[1].xyz) + diff(o[0].dreproj, o[1].dreproj) + diff(o[0].image, o[1].image) +
                     diff(o[0].count, o[1].count) + diff(o[0].mask, o[1].mask) + diff(o[0].mgeo, o[1].mgeo) + diff(o[0].is2r, o[1].is2r);
  double mean = 0, kept = 0;
  for (auto c : o[0].count) mean += c;
  for (auto m : o[0].mask) kept += m != 0;
  printf("fusion %dx%d, %d source views: paired vs one tap per load: %s; mean consistent views %.2f, %.0f %% of the pixels kept\n", W, H, S,
         bad ? "DIFFERENT" : "all 8 outputs bit-equal", mean / hw, 100.0 * kept / hw);
  return (bad || mean <= 0) ? 1.0 : 0.0;
}

int main(int argc, char **argv) {
  hipemu::g_lds = nullptr;   // no dynamic LDS in these kernels
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) take(wgrad_check(1, 5, 9, 36));    // ragged in z, y and x: 2 x 2 x 2 tiles walked by one workgroup per channel pair
  if (all || quick) take(fusion_check(12, 72, 3));
  if (all) {
    take(wgrad_check(2, 4, 8, 64));
    take(fusion_check(8, 300, 5));                     // two workgroups per row, the second one ragged
  }
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
