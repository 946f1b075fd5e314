// As run_kernels.cpp, for the production depth-walking `prob` head (csrc/prob_regress.hip: Conv3d 8 -> 1 + bias with the softmax regression fused or as a
// second step; GPU-validated): a regression test of its device code that needs no GPU, and - under ThreadSanitizer - of its one-barrier-per-plane slot
// rotation.  Against the layer and mvsnet.py:174-193 in float64.
#include <hip/hip_runtime.h>
namespace {
alignas(64) float smem[HIPEMU_LDS_BYTES / 4];   // prob_zwalk_kernel's `extern __shared__ float smem[]`
}
#include "support.h"

#include "prob_regress.hip"

// chunked depth ranges hand the cost volume to casmvs_softmax_regress_f32 (depth_ops.hip): here, the shared per-pixel routine on the host
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

static double prob_check(int B, int D, int H, int W, int zchunk) {
  const size_t no = (size_t)D * H * W;
  std::vector<float> x((size_t)B * 8 * no), wp(8 * 27), dv((size_t)B * no);
  for (auto &v : x) v = rnd();
  for (auto &v : wp) v = rnd() * 0.3f;
  const float bias = 0.125f;
  for (int b = 0; b < B; ++b)
    for (int z = 0; z < D; ++z)
      for (size_t p = 0; p < (size_t)H * W; ++p) dv[((size_t)b * D + z) * H * W + p] = 425.0f + 2.5f * z + 0.01f * (float)(p % 7);
  // `prob` image as casmvs_conv3d_pack_f32(CASMVS_CONV_S1, 8, 1) writes it (conv3d_mfma.hip, P1 format): [pair][2 tap + channel & 1] (64 floats), scale[4] | shift[4]
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
  float *xa = dup(x), *dva = dup(dv), *ppa = dup(ppk);
  std::vector<float> nanv((size_t)B * no, NAN), nan2((size_t)B * H * W, NAN);
  float *cost = dup(nanv), *depth = dup(nan2), *conf = dup(nan2);
  std::vector<int32_t> index((size_t)B * H * W, -1);
  if (casmvs_prob_regress_f32(ppa, xa, dva, cost, depth, conf, index.data(), B, 8, D, H, W, 1.0f, zchunk, nullptr)) {
    printf("prob_regress: %s\n", casmvs_last_error());
    return 1e9;
  }
  double err = 0, range = 0, derr = 0, cerr = 0;
  long index_off = 0;
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
      if (std::fabs(is - std::round(is)) > 1e-3) {   // away from an index boundary
        cerr = std::fmax(cerr, std::isfinite(gc) ? std::fabs(c4 - gc) : 1e30);
        index_off += index[(size_t)b * H * W + p] != idx;
      }
    }
  }
  std::free(xa); std::free(dva); std::free(ppa); std::free(cost); std::free(depth); std::free(conf);
  printf("prob_zwalk B=%d %dx%dx%d zchunk %d: cost max error / range = %.2e, depth rel %.2e, confidence abs %.2e, %ld indices off\n", B, D, H, W, zchunk,
         err / range, derr, cerr, index_off);
  return std::fmax(err / range, std::fmax(derr * 1e-2, std::fmax(cerr * 1e-2, (double)index_off)));   // depth / confidence to 2e-4
}

int main(int argc, char **argv) {
  hipemu::g_lds = reinterpret_cast<unsigned char *>(smem);
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) take(prob_check(1, 8, 10, 68, 8));       // one chunk: regression fused (DT = 8); two tiles in x (64 + 4), two in y
  if (all || quick) take(prob_check(1, 8, 10, 68, 4));       // chunks of 4 planes: halo planes at the chunk ends, separate regression
  if (which == "streams") take(prob_check(1, 8, 16, 128, 8));
  if (all) {
    take(prob_check(2, 6, 9, 132, 0));                       // generic-depth fused path, three x tiles, automatic chunking
    take(prob_check(1, 16, 8, 64, 0));
  }
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
