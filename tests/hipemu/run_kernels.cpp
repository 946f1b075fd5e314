// Runs kernels of casmvsnet_pl_amd/csrc on the CPU through tests/hipemu/hip/hip_runtime.h - their own source - against float64 references:
//   conv0_sf    the established tiled conv0 kernel (validated on the MI355X): checks the EMULATOR
//   conv0_zm / deconv11 / deconv9    written without a GPU run at the end of round 3 with this emulation as their only test; all three passed
//                                    their first run on the MI355X unchanged (profiles/r04_native_checks_first_run.txt) and are defaults now
// Build (tests/test_hip_emulation.py does it):
//   /opt/rocm/lib/llvm/bin/clang++ -std=c++20 -O1 -pthread -DCASMVS_SPLIT_NOASM -Itests/hipemu -Iinclude -Icasmvsnet_pl_amd/csrc tests/hipemu/run_kernels.cpp -o <exe>
#include "support.h"

#include "conv0_splitf16.hip"
#include "conv0_zmarch.hip"
#include "deconv11_splitf16.hip"
#include "deconv9_splitf16.hip"

static double conv3d_check(const char *name, int cin, int B, int D, int H, int W, bool zmarch) {
  const size_t n = (size_t)D * H * W;
  std::vector<float> x((size_t)B * cin * n), w((size_t)8 * cin * 27), sc(8), sh(8), y((size_t)B * 8 * n, NAN);
  for (auto &v : x) v = rnd() * 3.0f + 0.4f;
  for (auto &v : w) v = rnd() * 0.2f;
  for (int c = 0; c < 8; ++c) { sc[c] = 0.5f + 0.1f * c; sh[c] = 0.05f * (c - 4); }
  std::vector<unsigned char> packed(casmvs_conv0_splitf16_packed_bytes(cin) + 16);
  unsigned char *pk = packed.data() + ((16 - (reinterpret_cast<size_t>(packed.data()) & 15)) & 15);
  if (casmvs_conv0_splitf16_pack(cin, w.data(), sc.data(), sh.data(), pk)) { printf("%s: pack: %s\n", name, casmvs_last_error()); return 1e9; }
  float *xa = (float *)std::aligned_alloc(256, (x.size() * 4 + 255) & ~(size_t)255), *ya = (float *)std::aligned_alloc(256, (y.size() * 4 + 255) & ~(size_t)255);   // as device allocations: whole cache lines
  std::memcpy(xa, x.data(), x.size() * 4);
  std::memcpy(ya, y.data(), y.size() * 4);
  const int rc = zmarch ? casmvs_conv0_zmarch_forward_f32(pk, xa, ya, B, cin, D, H, W, 0.01f, nullptr)
                        : casmvs_conv0_splitf16_forward_f32(pk, xa, ya, B, cin, D, H, W, 0.01f, 0, nullptr);
  if (rc) { printf("%s: %s\n", name, casmvs_last_error()); return 1e9; }
  double err = 0, range = 0;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < 8; ++co)
      for (int z = 0; z < D; ++z)
        for (int yy = 0; yy < H; ++yy)
          for (int xx = 0; xx < W; ++xx) {
            double acc = 0;
            for (int ci = 0; ci < cin; ++ci)
              for (int kz = 0; kz < 3; ++kz)
                for (int ky = 0; ky < 3; ++ky)
                  for (int kx = 0; kx < 3; ++kx) {
                    const int iz = z + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                    if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                    acc += (double)w[((size_t)co * cin + ci) * 27 + kz * 9 + ky * 3 + kx] * x[((size_t)b * cin + ci) * n + ((size_t)iz * H + iy) * W + ix];
                  }
            const double v = lrelu(acc * sc[co] + sh[co]);
            const float got = ya[((size_t)b * 8 + co) * n + ((size_t)z * H + yy) * W + xx];
            range = std::fmax(range, std::fabs(v));
            err = std::fmax(err, std::isfinite(got) ? std::fabs(v - got) : 1e30);
          }
  std::free(xa); std::free(ya);
  printf("%-10s cin=%d B=%d %dx%dx%d: max error / range = %.2e\n", name, cin, B, D, H, W, err / range);
  return err / range;
}

static double deconv_check(int cin, int cout, int B, int Di, int Hi, int Wi) {
  const size_t ni = (size_t)Di * Hi * Wi, no = ni * 8;
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  std::vector<float> x((size_t)B * cin * ni), w((size_t)cin * cout * 27), sc(cout), sh(cout), sk((size_t)B * cout * no);
  for (auto &v : x) v = rnd() * 2.0f + 0.2f;
  for (auto &v : w) v = rnd() * 0.2f;
  for (auto &v : sk) v = rnd();
  for (int c = 0; c < cout; ++c) { sc[c] = 0.5f + 0.05f * c; sh[c] = 0.03f * (c - 4); }
  const size_t pb = cout == 8 ? casmvs_deconv11_splitf16_packed_bytes() : casmvs_deconv9_splitf16_packed_bytes();
  unsigned char *pk = (unsigned char *)std::aligned_alloc(256, (pb + 255) & ~(size_t)255);
  if (cout == 8 ? casmvs_deconv11_splitf16_pack(w.data(), sc.data(), sh.data(), pk) : casmvs_deconv9_splitf16_pack(w.data(), sc.data(), sh.data(), pk)) {
    printf("deconv pack: %s\n", casmvs_last_error());
    return 1e9;
  }
  float *xa = (float *)std::aligned_alloc(256, (x.size() * 4 + 255) & ~(size_t)255), *ska = (float *)std::aligned_alloc(256, (sk.size() * 4 + 255) & ~(size_t)255),
        *ya = (float *)std::aligned_alloc(256, (sk.size() * 4 + 255) & ~(size_t)255);
  std::memcpy(xa, x.data(), x.size() * 4);
  std::memcpy(ska, sk.data(), sk.size() * 4);
  for (size_t i = 0; i < sk.size(); ++i) ya[i] = NAN;
  const int rc = cout == 8 ? casmvs_deconv11_splitf16_forward_f32(pk, xa, ska, ya, B, Di, Hi, Wi, 0.01f, nullptr)
                           : casmvs_deconv9_splitf16_forward_f32(pk, xa, ska, ya, B, Di, Hi, Wi, 0.01f, nullptr);
  if (rc) { printf("deconv: %s\n", casmvs_last_error()); return 1e9; }
  std::vector<double> ref(sk.size(), 0.0);
  for (int b = 0; b < B; ++b)
    for (int ci = 0; ci < cin; ++ci)
      for (int iz = 0; iz < Di; ++iz)
        for (int iy = 0; iy < Hi; ++iy)
          for (int ix = 0; ix < Wi; ++ix) {
            const double v = x[((size_t)b * cin + ci) * ni + ((size_t)iz * Hi + iy) * Wi + ix];
            for (int co = 0; co < cout; ++co)
              for (int kz = 0; kz < 3; ++kz)
                for (int ky = 0; ky < 3; ++ky)
                  for (int kx = 0; kx < 3; ++kx) {
                    const int oz = 2 * iz - 1 + kz, oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx;
                    if (oz < 0 || oz >= Do || oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
                    ref[((size_t)b * cout + co) * no + ((size_t)oz * Ho + oy) * Wo + ox] += v * w[((size_t)ci * cout + co) * 27 + kz * 9 + ky * 3 + kx];
                  }
          }
  double err = 0, range = 0;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < cout; ++co)
      for (size_t i = 0; i < no; ++i) {
        const size_t o = ((size_t)b * cout + co) * no + i;
        const double v = lrelu(ref[o] * sc[co] + sh[co]) + sk[o];
        range = std::fmax(range, std::fabs(v));
        err = std::fmax(err, std::isfinite(ya[o]) ? std::fabs(v - ya[o]) : 1e30);
      }
  std::free(pk); std::free(xa); std::free(ska); std::free(ya);
  printf("deconv%-2d    B=%d in %dx%dx%d: max error / range = %.2e\n", cout == 8 ? 11 : 9, B, Di, Hi, Wi, err / range);
  return err / range;
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem_raw;
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";   // quick: one small ragged case per kernel (the CPU test suite)
  if (all || quick || which == "conv0_sf") take(conv3d_check("conv0_sf", 8, 1, 5, 9, 36, false));
  if (all || which == "conv0_sf") take(conv3d_check("conv0_sf", 16, 1, 3, 6, 32, false));
  if (all || quick || which == "conv0_zm") take(conv3d_check("conv0_zm", 16, 1, 5, 17, 36, true));
  if (all || quick || which == "conv0_zw") take(conv3d_check("conv0_zw", 32, 1, 3, 20, 36, true));   // cin = 32: the warp-specialised form (8 waves, two patch buffers)
  if (all || which == "conv0_zm") {
    take(conv3d_check("conv0_zm", 8, 1, 5, 20, 36, true));
    take(conv3d_check("conv0_zm", 16, 2, 9, 17, 44, true));
    take(conv3d_check("conv0_zm", 32, 1, 3, 16, 32, true));
  }
  if (which == "conv0_compare") {   // both conv0 kernels on ONE interior-dominated problem (tools/lds_bank_profile.py: their global request streams)
    take(conv3d_check("conv0_sf", 16, 1, 8, 32, 128, false));
    take(conv3d_check("conv0_zm", 16, 1, 8, 32, 128, true));
  }
  if (which == "streams") {   // interior-dominated problems with cache-line-aligned rows: the request streams of the other unmeasured kernels (tools/lds_bank_profile.py)
    take(deconv_check(16, 8, 1, 4, 8, 64));
    take(deconv_check(32, 16, 1, 2, 8, 64));
  }
  if (all || quick || which == "deconv11") take(deconv_check(16, 8, 1, 2, 5, 18));
  if (all || which == "deconv11") { take(deconv_check(16, 8, 2, 3, 5, 10)); take(deconv_check(16, 8, 1, 1, 9, 22)); }
  if (all || quick || which == "deconv9") take(deconv_check(32, 16, 1, 1, 5, 18));
  if (all || which == "deconv9") take(deconv_check(32, 16, 1, 2, 5, 18));
  printf(worst < 2e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
