// As run_kernels.cpp, for the float32 matrix-core layers of CostRegNet (csrc/conv3d_mfma.hip: conv16db_kernel / conv16_kernel / deconv16_kernel on
// v_mfma_f32_16x16x4_f32 - conv1 / conv3 / conv5, conv7 / conv9 / conv11 and every layer of the all-float32 replicas: ~35 % of the forward's GPU time;
// validated on the MI355X since round 1): Conv3d k3 s1 / s2 and ConvTranspose3d k3 s2 (+ skip) through casmvs_conv3d_forward_f32 against the layers in float64.
// The other kernel families the file's engine functions call are link stubs (tests/test_hip_emulation.py generates them from the undefined symbols).
#include <hip/hip_runtime.h>
inline hipError_t hipEventRecord(void *, void *) { return 0; }
inline hipError_t hipEventCreate(void **e) { *e = nullptr; return 0; }
inline hipError_t hipEventDestroy(void *) { return 0; }
inline hipError_t hipEventSynchronize(void *) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, void *, void *) { *ms = 0.0f; return 0; }
template <class V>
inline V hipemu_mfma_unsupported(float, float, V c, int, int, int) { std::fprintf(stderr, "hipemu: this MFMA shape (probe kernels only) is not emulated\n"); std::abort(); return c; }
#define __builtin_amdgcn_mfma_f32_4x4x1f32 hipemu_mfma_unsupported
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_unsupported
#define __builtin_amdgcn_mfma_f32_16x16x1f32 hipemu_mfma_unsupported
namespace {
alignas(64) float smem[HIPEMU_LDS_BYTES / 4];   // `extern __shared__ float smem[]` of the 3D kernels
alignas(64) float lds2[HIPEMU_LDS_BYTES / 4];   // ... of the 2D kernels
}
#include "support.h"

#include "conv3d_mfma.hip"

static double conv3d_f32_check(const char *name, int kind, int B, int cin, int cout, int D, int H, int W, bool with_skip) {
  const bool tr = kind == CASMVS_CONV_T2;
  const int S = kind == CASMVS_CONV_S1 ? 1 : 2;
  const int Do = tr ? 2 * D : D / (kind == CASMVS_CONV_S2 ? 2 : 1), Ho = tr ? 2 * H : H / (kind == CASMVS_CONV_S2 ? 2 : 1), Wo = tr ? 2 * W : W / (kind == CASMVS_CONV_S2 ? 2 : 1);
  const size_t ni = (size_t)D * H * W, no = (size_t)Do * Ho * Wo;
  std::vector<float> x((size_t)B * cin * ni), w((size_t)cin * cout * 27), sc(cout), sh(cout), skip((size_t)B * cout * no);
  for (auto &v : x) v = rnd();
  for (auto &v : w) v = rnd() * 0.2f;
  for (auto &v : skip) v = rnd();
  for (int c = 0; c < cout; ++c) { sc[c] = 0.5f + 0.05f * c; sh[c] = 0.03f * (c - 4); }
  const size_t pf = casmvs_conv3d_packed_floats(kind, cin, cout);
  if (!pf) { printf("%s: layer not supported\n", name); return 1e9; }
  float *pk = (float *)std::aligned_alloc(256, (pf * 4 + 255) & ~(size_t)255);
  if (casmvs_conv3d_pack_f32(kind, cin, cout, w.data(), sc.data(), sh.data(), pk)) { printf("%s: pack: %s\n", name, casmvs_last_error()); return 1e9; }
  auto dup = [](const std::vector<float> &v) {
    float *p = (float *)std::aligned_alloc(256, (v.size() * 4 + 255) & ~(size_t)255);
    std::memcpy(p, v.data(), v.size() * 4);
    return p;
  };
  float *xa = dup(x), *ska = dup(skip);
  std::vector<float> nanv((size_t)B * cout * no, NAN);
  float *out = dup(nanv);
  if (casmvs_conv3d_forward_f32(kind, pk, xa, with_skip ? ska : nullptr, out, B, cin, cout, D, H, W, 0.01f, nullptr)) { printf("%s: %s\n", name, casmvs_last_error()); return 1e9; }
  std::vector<double> ref((size_t)B * cout * no, 0.0);
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < cout; ++co)
      for (int ci = 0; ci < cin; ++ci)
        for (int kz = 0; kz < 3; ++kz)
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const double wt = tr ? w[(((size_t)ci * cout + co) * 27) + kz * 9 + ky * 3 + kx] : w[(((size_t)co * cin + ci) * 27) + kz * 9 + ky * 3 + kx];
              if (!tr) {
                for (int oz = 0; oz < Do; ++oz)
                  for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox) {
                      const int iz = S * oz - 1 + kz, iy = S * oy - 1 + ky, ix = S * ox - 1 + kx;
                      if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                      ref[((size_t)b * cout + co) * no + ((size_t)oz * Ho + oy) * Wo + ox] += wt * x[((size_t)b * cin + ci) * ni + ((size_t)iz * H + iy) * W + ix];
                    }
              } else {
                for (int iz = 0; iz < D; ++iz)
                  for (int iy = 0; iy < H; ++iy)
                    for (int ix = 0; ix < W; ++ix) {
                      const int oz = 2 * iz - 1 + kz, oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx;
                      if (oz < 0 || oz >= Do || oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
                      ref[((size_t)b * cout + co) * no + ((size_t)oz * Ho + oy) * Wo + ox] += wt * x[((size_t)b * cin + ci) * ni + ((size_t)iz * H + iy) * W + ix];
                    }
              }
            }
  double err = 0, range = 0;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < cout; ++co)
      for (size_t i = 0; i < no; ++i) {
        const size_t o = ((size_t)b * cout + co) * no + i;
        const double v = lrelu(ref[o] * sc[co] + sh[co]) + (with_skip ? (double)skip[o] : 0.0);
        range = std::fmax(range, std::fabs(v));
        err = std::fmax(err, std::isfinite(out[o]) ? std::fabs(v - out[o]) : 1e30);
      }
  std::free(pk); std::free(xa); std::free(ska); std::free(out);
  printf("conv3d_f32 %-4s B=%d %d -> %d input %dx%dx%d%s: max error / range = %.2e\n", name, B, cin, cout, D, H, W, with_skip ? " + skip" : "", err / range);
  return err / range;
}

// FeatureNet's float32 layers through casmvs_conv2d_forward_f32: Conv2d k3 s1 p1 / k5 s2 p2 (+ folded ABN + leaky-relu)
static double conv2d_f32_check(const char *name, int kind, int N, int cin, int cout, int H, int W) {
  const int KS = kind == CASMVS_CONV2D_K5S2 ? 5 : 3, S = kind == CASMVS_CONV2D_K5S2 ? 2 : 1, P = KS / 2, Ho = H / S, Wo = W / S;
  std::vector<float> x((size_t)N * cin * H * W), w((size_t)cout * cin * KS * KS), sc(cout), sh(cout);
  for (auto &v : x) v = rnd();
  for (auto &v : w) v = rnd() * 0.2f;
  for (int c = 0; c < cout; ++c) { sc[c] = 0.5f + 0.05f * c; sh[c] = 0.03f * (c - 4); }
  const size_t pf = casmvs_conv2d_packed_floats(kind, cin, cout);
  if (!pf) { printf("%s: layer not supported\n", name); return 1e9; }
  float *pk = (float *)std::aligned_alloc(256, (pf * 4 + 255) & ~(size_t)255);
  if (casmvs_conv2d_pack_f32(kind, cin, cout, w.data(), sc.data(), sh.data(), pk)) { printf("%s: pack: %s\n", name, casmvs_last_error()); return 1e9; }
  float *xa = (float *)std::aligned_alloc(256, (x.size() * 4 + 255) & ~(size_t)255);
  std::memcpy(xa, x.data(), x.size() * 4);
  std::vector<float> out((size_t)N * cout * Ho * Wo, NAN);
  float *oa = (float *)std::aligned_alloc(256, (out.size() * 4 + 255) & ~(size_t)255);
  std::memcpy(oa, out.data(), out.size() * 4);
  if (casmvs_conv2d_forward_f32(kind, pk, xa, nullptr, oa, N, cin, cout, H, W, 0.01f, nullptr)) { printf("%s: %s\n", name, casmvs_last_error()); return 1e9; }
  double err = 0, range = 0;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < cout; ++co)
      for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox) {
          double acc = 0;
          for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < KS; ++ky)
              for (int kx = 0; kx < KS; ++kx) {
                const int iy = S * oy - P + ky, ix = S * ox - P + kx;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                acc += (double)w[(((size_t)co * cin + ci) * KS + ky) * KS + kx] * x[(((size_t)n * cin + ci) * H + iy) * W + ix];
              }
          const double v = lrelu(acc * sc[co] + sh[co]);
          const float got = oa[(((size_t)n * cout + co) * Ho + oy) * Wo + ox];
          range = std::fmax(range, std::fabs(v));
          err = std::fmax(err, std::isfinite(got) ? std::fabs(v - got) : 1e30);
        }
  std::free(pk); std::free(xa); std::free(oa);
  printf("conv2d_f32 %-4s N=%d %d -> %d input %dx%d: max error / range = %.2e\n", name, N, cin, cout, H, W, err / range);
  return err / range;
}

int main(int argc, char **argv) {
  hipemu::g_lds = reinterpret_cast<unsigned char *>(smem);
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) {
    take(conv3d_f32_check("S2", CASMVS_CONV_S2, 1, 8, 16, 4, 6, 20, false));      // conv1
    take(conv3d_f32_check("T2", CASMVS_CONV_T2, 1, 16, 8, 2, 3, 10, true));       // conv11 + skip
  }
  if (which == "streams") {   // interior-dominated, cache-line-aligned rows (tools/lds_bank_profile.py: request streams)
    take(conv3d_f32_check("S2", CASMVS_CONV_S2, 1, 8, 16, 8, 16, 128, false));    // conv1
    take(conv3d_f32_check("T2", CASMVS_CONV_T2, 1, 16, 8, 4, 8, 64, true));       // conv11 + skip
  }
  if (all) {
    take(conv3d_f32_check("S1", CASMVS_CONV_S1, 1, 8, 8, 3, 5, 20, false));       // conv0 on the float32 kernel (PX form)
    take(conv3d_f32_check("S1", CASMVS_CONV_S1, 1, 16, 16, 3, 5, 18, false));     // conv2 (CI form)
    take(conv3d_f32_check("S2", CASMVS_CONV_S2, 2, 16, 32, 2, 4, 12, false));     // conv3
    take(conv3d_f32_check("T2", CASMVS_CONV_T2, 1, 32, 16, 1, 3, 6, true));       // conv9 + skip
    take(conv3d_f32_check("S1", CASMVS_CONV_S1, 1, 8, 1, 3, 5, 20, false));       // prob (tile kernels)
    take(conv2d_f32_check("K3", CASMVS_CONV2D_K3, 2, 3, 8, 10, 36));               // FeatureNet conv0.0
    take(conv2d_f32_check("K3", CASMVS_CONV2D_K3, 1, 16, 16, 9, 20));              // conv1.1 on the float32 kernel
    take(conv2d_f32_check("K5S2", CASMVS_CONV2D_K5S2, 1, 8, 16, 12, 40));          // conv1.0
    take(conv2d_f32_check("K5S2", CASMVS_CONV2D_K5S2, 2, 16, 32, 8, 20));          // conv2.0
  }
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
