// As run_kernels4.cpp, for the depth-walking fusion of CostRegNet's tail (csrc/conv11_prob_zfused.hip: conv11 = ConvTranspose3d 16 -> 8 + ABN + leaky-relu +
// skip, `prob` = Conv3d 8 -> 1 + bias, softmax regression): written in round 4 with this emulation as its first test.  Against the three layers in float64 on
// ragged shapes (image borders inside a tile, several tiles in x and y, 512-thread workgroups), and - under ThreadSanitizer - of its one-barrier-per-plane
// rotation of plane slots and input-plane boxes.
#include "support.h"

#include "conv11_prob_zfused.hip"
#include "deconv11_splitf16.hip"

static double fused_check(int B, int Di, int Hi, int Wi) {
  const int D = 2 * Di, H = 2 * Hi, W = 2 * Wi;
  const size_t ni = (size_t)Di * Hi * Wi, no = (size_t)D * H * W;
  std::vector<float> x((size_t)B * 16 * ni), sk((size_t)B * 8 * no), w11(16 * 8 * 27), sc(8), sh(8), wp(8 * 27), dv((size_t)B * no);
  for (auto &v : x) v = rnd() * 2.0f + 0.2f;
  for (size_t i = 0; i < x.size(); i += 53) x[i] *= 30.0f;   // planes of different magnitude -> different scales of the two chains
  for (auto &v : sk) v = rnd();
  for (auto &v : w11) v = rnd() * 0.2f;
  for (int c = 0; c < 8; ++c) { sc[c] = 0.5f + 0.05f * c; sh[c] = 0.03f * (c - 4); }
  for (auto &v : wp) v = rnd() * 0.3f;
  const float bias = 0.125f;
  for (int b = 0; b < B; ++b)
    for (int z = 0; z < D; ++z)
      for (size_t p = 0; p < (size_t)H * W; ++p) dv[((size_t)b * D + z) * H * W + p] = 425.0f + 2.5f * z + 0.01f * (float)(p % 7);
  std::vector<float> ppk(4 * 64 + 8 + 64, 0.0f);   // `prob`'s P1 image (run_kernels4.cpp)
  for (int un = 0; un < 4; ++un)
    for (int l = 0; l < 54; ++l) ppk[un * 64 + l] = wp[(size_t)(2 * un + (l & 1)) * 27 + (l >> 1)];
  ppk[256] = 1.0f;
  ppk[260] = bias;
  auto dup = [](const std::vector<float> &v) {
    float *p = (float *)std::aligned_alloc(256, (v.size() * 4 + 255) & ~(size_t)255);
    std::memcpy(p, v.data(), v.size() * 4);
    return p;
  };
  unsigned char *dpk = (unsigned char *)std::aligned_alloc(256, (casmvs_deconv11_splitf16_packed_bytes() + 255) & ~(size_t)255);
  if (casmvs_deconv11_splitf16_pack(w11.data(), sc.data(), sh.data(), dpk)) { printf("pack: %s\n", casmvs_last_error()); return 1e9; }
  float *xa = dup(x), *ska = dup(sk), *dva = dup(dv), *ppa = dup(ppk);
  std::vector<float> nanv((size_t)B * no, NAN), nan2((size_t)B * H * W, NAN);
  float *cost = dup(nanv), *depth = dup(nan2), *conf = dup(nan2);
  std::vector<int32_t> index((size_t)B * H * W, -1);
  if (casmvs_conv11_prob_zfused_f32(dpk, ppa, xa, ska, dva, cost, depth, conf, index.data(), B, Di, Hi, Wi, 0.01f, 1.0f, nullptr)) {
    printf("conv11_prob_zfused: %s\n", casmvs_last_error());
    return 1e9;
  }
  double err = 0, range = 0, derr = 0, cerr = 0;
  long index_off = 0;
  std::vector<double> u11(8 * no), cref(no);
  for (int b = 0; b < B; ++b) {
    // conv11 in float64: out[o] += in[i] w[k], o = 2 i - 1 + k
    std::fill(u11.begin(), u11.end(), 0.0);
    for (int ci = 0; ci < 16; ++ci)
      for (int iz = 0; iz < Di; ++iz)
        for (int iy = 0; iy < Hi; ++iy)
          for (int ix = 0; ix < Wi; ++ix) {
            const double v = x[((size_t)b * 16 + ci) * ni + ((size_t)iz * Hi + iy) * Wi + ix];
            for (int co = 0; co < 8; ++co)
              for (int kz = 0; kz < 3; ++kz)
                for (int ky = 0; ky < 3; ++ky)
                  for (int kx = 0; kx < 3; ++kx) {
                    const int oz = 2 * iz - 1 + kz, oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx;
                    if (oz < 0 || oz >= D || oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
                    u11[(size_t)co * no + ((size_t)oz * H + oy) * W + ox] += v * w11[(((size_t)ci * 8 + co) * 27) + kz * 9 + ky * 3 + kx];
                  }
          }
    for (int co = 0; co < 8; ++co)
      for (size_t i = 0; i < no; ++i) {
        double v = u11[(size_t)co * no + i] * sc[co] + sh[co];
        u11[(size_t)co * no + i] = (v > 0 ? v : v * 0.01f) + sk[((size_t)b * 8 + co) * no + i];
      }
    for (int z = 0; z < D; ++z)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {
          double acc = bias;
          for (int ci = 0; ci < 8; ++ci)
            for (int kz = 0; kz < 3; ++kz)
              for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                  const int iz = z + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                  if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                  acc += (double)wp[(size_t)ci * 27 + kz * 9 + ky * 3 + kx] * u11[(size_t)ci * no + ((size_t)iz * H + iy) * W + ix];
                }
          cref[((size_t)z * H + yy) * W + xx] = acc;
          const float got = cost[(size_t)b * no + ((size_t)z * H + yy) * W + xx];
          range = std::fmax(range, std::fabs(acc));
          err = std::fmax(err, std::isfinite(got) ? std::fabs(acc - got) : 1e30);
        }
    for (size_t p = 0; p < (size_t)H * W; ++p) {
      double mx = -1e300, sum = 0, ds = 0, is = 0;
      for (int z = 0; z < D; ++z) mx = std::fmax(mx, cref[(size_t)z * H * W + p]);
      for (int z = 0; z < D; ++z) sum += std::exp(cref[(size_t)z * H * W + p] - mx);
      for (int z = 0; z < D; ++z) {
        const double pk = std::exp(cref[(size_t)z * H * W + p] - mx) / sum;
        ds += pk * dv[((size_t)b * D + z) * H * W + p];
        is += pk * z;
      }
      const int idx = (int)std::fmin(std::fmax(is, 0.0), D - 1.0);
      double c4 = 0;
      for (int z = std::max(idx - 1, 0); z <= std::min(idx + 2, D - 1); ++z) c4 += std::exp(cref[(size_t)z * H * W + p] - mx) / sum;
      const float gd = depth[(size_t)b * H * W + p], gc = conf[(size_t)b * H * W + p];
      derr = std::fmax(derr, std::isfinite(gd) ? std::fabs(ds - gd) / ds : 1e30);
      if (std::fabs(is - std::round(is)) > 1e-3) {   // away from an index boundary
        cerr = std::fmax(cerr, std::isfinite(gc) ? std::fabs(c4 - gc) : 1e30);
        index_off += index[(size_t)b * H * W + p] != idx;
      }
    }
  }
  std::free(xa); std::free(ska); std::free(dva); std::free(ppa); std::free(cost); std::free(depth); std::free(conf); std::free(dpk);
  printf("conv11_prob_zfused B=%d in %dx%dx%d: cost max error / range = %.2e, depth rel %.2e, confidence abs %.2e, %ld indices off\n", B, Di, Hi, Wi, err / range, derr, cerr,
         index_off);
  return std::fmax(err / range, std::fmax(derr * 1e-2, std::fmax(cerr * 1e-2, (double)index_off)));
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem_raw;
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) take(fused_check(1, 2, 5, 34));      // one tile in y (10 of 16 rows), two in x (60 + 8): generic depth (4 planes)
  if (all) {
    take(fused_check(2, 4, 9, 32));                      // depth 8 (compile-time softmax), two tiles in y, 64 = 60 + 4 columns
    take(fused_check(1, 3, 8, 62));                      // odd number of input planes, 124 columns = 2 tiles + 4
    take(fused_check(1, 1, 1, 2));                       // the smallest volume
  }
  printf(worst < 3e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 3e-6 ? 0 : 1;
}
