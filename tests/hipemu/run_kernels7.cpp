// As run_kernels.cpp, for the production LDS-staged plane sweep (csrc/costvol_lds.hip: the fused homo_warp + variance cost volume, the second largest
// kernel of the step; GPU-validated, bit-identical to the gather kernels there): a regression test of its device code that needs no GPU - box extents by
// wave reductions, both source boxes staged in one pass, the plane loop's taps from LDS, the transposed 16-byte volume stores - and, under
// ThreadSanitizer, of its barriers.  Reference: models/mvsnet.py:147-167 per voxel on the host, the tap positions and weights from the SAME float32
// routine the kernels use (plane_sweep.h: plane_sweep_taps - an ordinary inline function here), the sums in float64.
#include <hip/hip_runtime.h>
namespace {
alignas(64) unsigned char smem[HIPEMU_LDS_BYTES];   // costvol_lds_kernel's `extern __shared__ unsigned char smem[]`
}
#include "support.h"

#include "costvol_lds.hip"

static double costvol_check(int B, int V, int C, int D, int h, int w, float slide) {
  const size_t hw = (size_t)h * w;
  std::vector<float> feats((size_t)B * V * hw * C), proj((size_t)B * (V - 1) * 12, 0.0f), depth((size_t)B * D * hw);
  for (auto &v : feats) v = rnd();
  for (int b = 0; b < B; ++b) {
    for (int v = 0; v < V - 1; ++v) {   // near-identity rotation, a baseline that slides the view by `slide` pixels per plane, a small vertical offset
      float *P = proj.data() + ((size_t)b * (V - 1) + v) * 12;
      P[0] = 1.0f; P[5] = 1.0f; P[10] = 1.0f;
      P[1] = 0.002f * (v + 1); P[4] = -0.002f * (v + 1);
      // x shift = P[3] / depth + P[2]: `slide` pixels per plane (depth step 2.5 at 425: 1 / 425 - 1 / 427.5 = 1.376e-5), about -2 at the first plane
      P[3] = (v % 2 ? -1.0f : 1.0f) * slide / 1.376e-5f;
      P[2] = -P[3] / 425.0f - 2.0f;
      P[7] = 0.3f * 425.0f * (v + 1);
    }
    for (int d = 0; d < D; ++d)
      for (size_t p = 0; p < hw; ++p) depth[((size_t)b * D + d) * hw + p] = 425.0f + 2.5f * d + 0.02f * (float)(p % 5);
    // hypotheses of 0 / denormal / negative depth (modules.py:72 divides by them): NaN / inf coordinates, whose taps ATen drops - and so must the
    // zero-padded boxes' plain weights (0 * NaN is NaN)
    if (hw > 40 && D > 2) {
      depth[((size_t)b * D + 1) * hw + 17] = 0.0f;
      depth[((size_t)b * D + 2) * hw + 18] = 1e-42f;
      depth[((size_t)b * D + 0) * hw + 19] = -3.0f;
      depth[((size_t)b * D + 1) * hw + 40] = INFINITY;
    }
  }
  auto dup = [](const std::vector<float> &v) {
    float *p = (float *)std::aligned_alloc(256, (v.size() * 4 + 255) & ~(size_t)255);
    std::memcpy(p, v.data(), v.size() * 4);
    return p;
  };
  float *fa = dup(feats), *pa = dup(proj), *da = dup(depth);
  std::vector<float> nanv((size_t)B * C * D * hw, NAN);
  float *out = dup(nanv);
  if (!casmvs_costvol_lds_supported(C, w, D, V - 1, 1)) { printf("costvol_lds: shape not supported\n"); return 1e9; }
  if (casmvs_costvol_var_lds_f32(fa, pa, da, out, B, V, C, h, w, D, nullptr)) { printf("costvol_lds: %s\n", casmvs_last_error()); return 1e9; }
  double err = 0, range = 0;
  long live = 0, total = 0;
  for (int b = 0; b < B; ++b)
    for (int d = 0; d < D; ++d)
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
          const float dv = depth[((size_t)b * D + d) * hw + (size_t)y * w + x];
          std::vector<double> s(C), q(C);
          const float *ref = feats.data() + (((size_t)b * V) * hw + (size_t)y * w + x) * C;
          for (int c = 0; c < C; ++c) { s[c] = ref[c]; q[c] = (double)ref[c] * ref[c]; }
          for (int v = 0; v < V - 1; ++v) {
            const casmvs_dev::Taps t = casmvs_dev::plane_sweep_taps(proj.data() + ((size_t)b * (V - 1) + v) * 12, (float)x, (float)y, dv, w, h);
            const float *src = feats.data() + ((size_t)b * V + 1 + v) * hw * C;
            live += casmvs_dev::taps_live(t);
            ++total;
            for (int c = 0; c < C; ++c) {
              const double val = (double)t.w_nl * src[((size_t)t.yn * w + t.xl) * C + c] + (double)t.w_nr * src[((size_t)t.yn * w + t.xl + 1) * C + c] +
                                 (double)t.w_sl * src[((size_t)t.ys * w + t.xl) * C + c] + (double)t.w_sr * src[((size_t)t.ys * w + t.xl + 1) * C + c];
              s[c] += val;
              q[c] += val * val;
            }
          }
          for (int c = 0; c < C; ++c) {
            const double m = s[c] / V, want = q[c] / V - m * m;
            const float got = out[(((size_t)b * C + c) * D + d) * hw + (size_t)y * w + x];
            range = std::fmax(range, std::fabs(want));
            err = std::fmax(err, std::isfinite(got) ? std::fabs(want - got) : 1e30);
          }
        }
  std::free(fa); std::free(pa); std::free(da); std::free(out);
  printf("costvol_lds B=%d V=%d C=%d %dx%dx%d: max error / range = %.2e (%.0f %% of the taps inside the source views)\n", B, V, C, D, h, w, err / range, 100.0 * live / total);
  return err / range;
}

// homo_warp with the source box staged straight from the channel planes (casmvs_homo_warp_lds_f32, the reference's (B, C, H, W) layout) against the same
// sweep on a pixel-major copy (casmvs_homo_warp_nhwc_f32): the same taps, the same LDS contents, the same arithmetic - equal bits
static double warp_nchw_check(int B, int C, int D, int h, int w, float slide) {
  const size_t hw = (size_t)h * w;
  std::vector<float> nchw((size_t)B * C * hw), nhwc((size_t)B * hw * C), proj((size_t)B * 12, 0.0f), depth((size_t)B * D * hw);
  for (auto &v : nchw) v = rnd();
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (size_t p = 0; p < hw; ++p) nhwc[((size_t)b * hw + p) * C + c] = nchw[((size_t)b * C + c) * hw + p];
  for (int b = 0; b < B; ++b) {
    float *P = proj.data() + (size_t)b * 12;
    P[0] = 1.0f; P[5] = 1.0f; P[10] = 1.0f; P[1] = 0.002f; P[4] = -0.002f;
    P[3] = slide / 1.376e-5f;
    P[2] = -P[3] / 425.0f - 2.0f;
    P[7] = 0.3f * 425.0f;
    for (int d = 0; d < D; ++d)
      for (size_t p = 0; p < hw; ++p) depth[((size_t)b * D + d) * hw + p] = 425.0f + 2.5f * d + 0.02f * (float)(p % 5);
  }
  auto dup = [](const std::vector<float> &v) {
    float *p = (float *)std::aligned_alloc(256, (v.size() * 4 + 255) & ~(size_t)255);
    std::memcpy(p, v.data(), v.size() * 4);
    return p;
  };
  float *a = dup(nchw), *bp = dup(nhwc), *pa = dup(proj), *da = dup(depth);
  std::vector<float> nanv((size_t)B * C * D * hw, NAN);
  float *o1 = dup(nanv), *o2 = dup(nanv);
  if (!casmvs_homo_warp_lds_supported(C, w, D)) { printf("warp_nchw: shape not supported\n"); return 1e9; }
  if (casmvs_homo_warp_lds_f32(a, pa, da, o1, B, C, h, w, D, nullptr)) { printf("warp_nchw: %s\n", casmvs_last_error()); return 1e9; }
  if (casmvs_homo_warp_nhwc_f32(bp, pa, da, o2, B, C, h, w, D, nullptr)) { printf("warp_nhwc: %s\n", casmvs_last_error()); return 1e9; }
  size_t diff = 0, live = 0;
  for (size_t i = 0; i < nanv.size(); ++i) { diff += std::memcmp(o1 + i, o2 + i, 4) != 0; live += o1[i] != 0.0f; }
  std::free(a); std::free(bp); std::free(pa); std::free(da); std::free(o1); std::free(o2);
  printf("warp_nchw B=%d C=%d %dx%dx%d: %zu of %zu values differ from the pixel-major sweep (%.0f %% non-zero)\n", B, C, D, h, w, diff, nanv.size(), 100.0 * live / nanv.size());
  return diff ? 1.0 : 0.0;
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem;
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) take(costvol_check(1, 3, 8, 8, 10, 64, 0.6f));     // C = 8: 64 x 4 tiles, two full tile rows + a ragged one
  if (all || quick) take(costvol_check(1, 3, 16, 8, 12, 36, 0.6f));    // C = 16: 32 x 8 tiles, ragged in x and y
  // persistent workgroups: 4 tiles x 2 chunks x 2 batch elements on the emulation's one-CU-per-XCD chip = every workgroup walks two items (and the XCDs
  // without a tile walk empty slots); under ThreadSanitizer: the barrier between an item's plane loop and the next item's box table / staging
  if (all || quick) take(costvol_check(2, 3, 16, 16, 9, 40, 0.6f));
  if (all || quick || which == "warp") take(warp_nchw_check(1, 8, 8, 10, 64, 0.6f));    // C = 8 (unit(px) = 2 px + px / 8), ragged rows
  if (all || which == "warp") { take(warp_nchw_check(2, 16, 8, 12, 36, 0.6f)); take(warp_nchw_check(1, 32, 8, 9, 32, 1.5f)); }
  if (all || quick) {   // 16 planes per workgroup (production: only where the launch keeps >= 4 rounds of workgroups)
    g_dc16_min_rounds = 0;
    take(costvol_check(1, 3, 16, 16, 12, 36, 0.6f));
    if (all) take(warp_nchw_check(1, 16, 32, 9, 36, 0.5f));
    g_dc16_min_rounds = 4;
  }
  if (all) {
    take(costvol_check(2, 3, 32, 16, 9, 32, 0.4f));                    // C = 32 as two channel splits, two plane chunks
    take(costvol_check(1, 2, 16, 8, 16, 64, 1.5f));                    // one source view, a fast epipolar slide (wide boxes)
  }
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
