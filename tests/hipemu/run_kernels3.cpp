// As run_kernels.cpp, for the fused tail of CostRegNet (csrc/conv11_prob_fused.hip: conv11 + skip + `prob` + softmax regression in one depth-walking kernel,
// written without a GPU run): against the layers in float64 - the cost volume to 2e-6 of its range, depth / confidence through the same softmax - with the depth
// range as one chunk (regression fused) and cut into chunks (halo planes re-produced at the chunk ends).
#include "support.h"

#include "deconv11_splitf16.hip"
#include "conv11_prob_fused.hip"

// the non-fused path hands the cost volume to casmvs_softmax_regress_f32 (depth_ops.hip): here, the shared per-pixel routine on the host
extern "C" int casmvs_softmax_regress_f32(const float *cost, const float *depth_values, float *depth, float *confidence, int32_t *index, int B, int D, int h,
                                          int w, void *) {
  const size_t hw = (size_t)h * w;
  for (int b = 0; b < B; ++b)
    for (size_t p = 0; p < hw; ++p) {
      float d, c;
      int ix;
      casmvs::softmax_regress_pixel<0>(cost + (size_t)b * D * hw + p, depth_values + (size_t)b * D * hw + p, hw, D, d, c, ix);
      depth[(size_t)b * hw + p] = d;
      confidence[(size_t)b * hw + p] = c;
      if (index) index[(size_t)b * hw + p] = ix;
    }
  return 0;
}

static double fused_check(int B, int D, int H, int W, int zchunk) {
  const int Dh = D / 2, Hh = H / 2, Wh = W / 2;
  const size_t ni = (size_t)Dh * Hh * Wh, no = (size_t)D * H * W;
  std::vector<float> u9((size_t)B * 16 * ni), w11(16 * 8 * 27), sc(8), sh(8), skip((size_t)B * 8 * no), wp(8 * 27), dv((size_t)B * no);
  for (auto &v : u9) v = rnd() * 2.0f + 0.2f;
  for (auto &v : w11) v = rnd() * 0.2f;
  for (auto &v : skip) v = rnd();
  for (auto &v : wp) v = rnd() * 0.3f;
  for (int c = 0; c < 8; ++c) { sc[c] = 0.5f + 0.1f * c; sh[c] = 0.05f * (c - 4); }
  const float bias = 0.125f;
  for (int b = 0; b < B; ++b)
    for (int z = 0; z < D; ++z)
      for (size_t p = 0; p < (size_t)H * W; ++p) dv[((size_t)b * D + z) * H * W + p] = 425.0f + 2.5f * z + 0.01f * (float)(p % 7);
  unsigned char *dpk = (unsigned char *)std::aligned_alloc(256, (casmvs_deconv11_splitf16_packed_bytes() + 255) & ~(size_t)255);
  casmvs_deconv11_splitf16_pack(w11.data(), sc.data(), sh.data(), dpk);
  // `prob` image as casmvs_conv3d_pack_f32(CASMVS_CONV_S1, 8, 1) writes it (conv3d_mfma.hip: P1 format): [pair][2 tap + channel & 1] (64 floats), scale[4] | shift[4], 64 zeros
  std::vector<float> ppk(4 * 64 + 8 + 64, 0.0f);
  for (int un = 0; un < 4; ++un)
    for (int l = 0; l < 54; ++l) ppk[un * 64 + l] = wp[(size_t)(2 * un + (l & 1)) * 27 + (l >> 1)];
  ppk[256] = 1.0f;
  ppk[260] = bias;
  auto dup = [](const std::vector<float> &v) {
    float *p = (float *)std::aligned_alloc(256, (v.size() * 4 + 255) & ~(size_t)255);
    std::memcpy(p, v.data(), v.size() * 4);
    return p;
  };
  float *u9a = dup(u9), *ska = dup(skip), *dva = dup(dv), *ppa = dup(ppk);
  std::vector<float> nanv((size_t)B * no, NAN), nan2((size_t)B * H * W, NAN);
  float *cost = dup(nanv), *depth = dup(nan2), *conf = dup(nan2);
  if (casmvs_conv11_prob_regress_f32(dpk, ppa, u9a, ska, dva, cost, depth, conf, nullptr, B, D, H, W, 0.01f, zchunk, nullptr)) {
    printf("conv11_prob: %s\n", casmvs_last_error());
    return 1e9;
  }
  // float64: x = lrelu(abn(deconv(u9))) + skip; cost = conv3d(x, wp) + bias; softmax regression
  std::vector<double> x((size_t)B * 8 * no, 0.0);
  for (int b = 0; b < B; ++b)
    for (int ci = 0; ci < 16; ++ci)
      for (int iz = 0; iz < Dh; ++iz)
        for (int iy = 0; iy < Hh; ++iy)
          for (int ix = 0; ix < Wh; ++ix) {
            const double v = u9[((size_t)b * 16 + ci) * ni + ((size_t)iz * Hh + iy) * Wh + ix];
            for (int co = 0; co < 8; ++co)
              for (int kz = 0; kz < 3; ++kz)
                for (int ky = 0; ky < 3; ++ky)
                  for (int kx = 0; kx < 3; ++kx) {
                    const int oz = 2 * iz - 1 + kz, oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx;
                    if (oz < 0 || oz >= D || oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
                    x[((size_t)b * 8 + co) * no + ((size_t)oz * H + oy) * W + ox] += v * w11[((size_t)ci * 8 + co) * 27 + kz * 9 + ky * 3 + kx];
                  }
          }
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < 8; ++co)
      for (size_t i = 0; i < no; ++i) {
        const size_t o = ((size_t)b * 8 + co) * no + i;
        x[o] = lrelu(x[o] * sc[co] + sh[co]) + skip[o];
      }
  double err = 0, range = 0, derr = 0, cerr = 0;
  std::vector<double> cref(no);
  for (int b = 0; b < B; ++b) {
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
                  acc += (double)wp[(size_t)ci * 27 + kz * 9 + ky * 3 + kx] * x[((size_t)b * 8 + ci) * no + ((size_t)iz * H + iy) * W + ix];
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
      if (std::fabs(is - std::round(is)) > 1e-3) cerr = std::fmax(cerr, std::isfinite(gc) ? std::fabs(c4 - gc) : 1e30);   // away from an index boundary
    }
  }
  std::free(dpk); std::free(u9a); std::free(ska); std::free(dva); std::free(ppa); std::free(cost); std::free(depth); std::free(conf);
  printf("conv11_prob B=%d %dx%dx%d zchunk %d: cost max error / range = %.2e, depth rel %.2e, confidence abs %.2e\n", B, D, H, W, zchunk, err / range, derr, cerr);
  return std::fmax(err / range, std::fmax(derr * 1e-2, cerr * 1e-2));   // depth / confidence to 2e-4
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem_raw;
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) take(fused_check(1, 8, 10, 68, 8));      // one chunk: regression fused; two tiles in x (62 + 6), two in y
  if (which == "streams") take(fused_check(1, 8, 16, 124, 8));   // two full x tiles (62 + 62): the request stream of an interior-dominated problem
  if (all) {
    take(fused_check(1, 8, 10, 68, 4));                      // chunks of 4 planes: halo planes at the chunk ends, separate regression
    take(fused_check(2, 6, 18, 124, 6));                     // generic-depth fused path, exact multiple of the x tile stride
    take(fused_check(1, 16, 8, 64, 16));                     // the fused instantiations of the cascade's plane counts: DT = 16, 32, 48
    take(fused_check(1, 32, 8, 64, 32));
    take(fused_check(1, 48, 8, 64, 48));
    take(fused_check(1, 48, 8, 64, 16));                     // 48 planes in three chunks
  }
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
