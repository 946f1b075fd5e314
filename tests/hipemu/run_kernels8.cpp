// As run_kernels.cpp, for the training path's matrix-core kernel (csrc/train.hip: conv_wgrad_kernel - the weight gradient of every convolution kind of the
// model as a GEMM over the positions on v_mfma_f32_16x16x4_f32, 2 ms of the 14.5 ms training step - and its fixed-order reduction) and the per-channel
// statistics kernel (channel_sums_kernel): GPU-validated device code as a regression test that needs no GPU, for the round that will work on the training
// step.  Weight gradients against the definition in float64 for all six kinds; (sum x, sum x^2) against float64 sums.
#define __shared__ static   // train.hip keeps its LDS in function-scope arrays (workgroups run one after the other here)
#include "support.h"

#include "train.hip"   // (includes plane_sweep.h)

// kind -> stride, kernel depth, kernel size, pad, transposed (train.hip: wgrad_geom)
struct Geo { int S, KZ, KS, transposed; };
static Geo geo_of(int kind) {
  switch (kind) {
    case CASMVS_CONV_S1: return {1, 3, 3, 0};
    case CASMVS_CONV_S2: return {2, 3, 3, 0};
    case CASMVS_CONV_T2: return {2, 3, 3, 1};
    case CASMVS_CONV2D_K3: return {1, 1, 3, 0};
    case CASMVS_CONV2D_K5S2: return {2, 1, 5, 0};
    default: return {1, 1, 1, 0};
  }
}

static double wgrad_check(const char *name, int kind, int B, int cin, int cout, int D, int H, int W) {
  const Geo g = geo_of(kind);
  const int P = g.KS / 2, PZ = g.KZ / 2, T = g.KZ * g.KS * g.KS;
  // output dims of the layer
  const int Do = g.transposed ? 2 * D : (g.KZ == 3 ? (g.S == 2 ? D / 2 : D) : 1), Ho = g.transposed ? 2 * H : H / g.S, Wo = g.transposed ? 2 * W : W / g.S;
  const size_t ni = (size_t)D * H * W, no = (size_t)Do * Ho * Wo;
  std::vector<float> x((size_t)B * cin * ni), gy((size_t)B * cout * no);
  for (auto &v : x) v = rnd();
  for (auto &v : gy) v = rnd();
  auto dup = [](const std::vector<float> &v) {
    float *p = (float *)std::aligned_alloc(256, (v.size() * 4 + 255) & ~(size_t)255);
    std::memcpy(p, v.data(), v.size() * 4);
    return p;
  };
  float *xa = dup(x), *ga = dup(gy);
  const size_t wsb = casmvs_conv_wgrad_workspace_bytes(kind, B, cin, cout, D, H, W);
  if (!wsb) { printf("%s: shape not supported\n", name); return 1e9; }
  void *ws = std::aligned_alloc(256, (wsb + 255) & ~(size_t)255);
  std::vector<float> got((size_t)cin * cout * T, NAN), got2((size_t)cin * cout * T, NAN);
  if (casmvs_conv_wgrad_f32(kind, xa, ga, got.data(), ws, B, cin, cout, D, H, W, nullptr)) { printf("%s: %s\n", name, casmvs_last_error()); return 1e9; }
  // a second run: the fixed-order reduction makes the gradient reproducible bit for bit
  if (casmvs_conv_wgrad_f32(kind, xa, ga, got2.data(), ws, B, cin, cout, D, H, W, nullptr)) { printf("%s (second run): %s\n", name, casmvs_last_error()); return 1e9; }
  const bool same = std::memcmp(got.data(), got2.data(), got.size() * 4) == 0;
  double err = 0, range = 0;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int kz = 0; kz < g.KZ; ++kz)
        for (int ky = 0; ky < g.KS; ++ky)
          for (int kx = 0; kx < g.KS; ++kx) {
            double acc = 0;
            for (int b = 0; b < B; ++b) {
              if (!g.transposed) {   // out[o] += w[co][ci][k] in[S o - P + k]
                for (int oz = 0; oz < Do; ++oz)
                  for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox) {
                      const int iz = g.S * oz - PZ + kz, iy = g.S * oy - P + ky, ix = g.S * ox - P + kx;
                      if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                      acc += (double)gy[((size_t)b * cout + co) * no + ((size_t)oz * Ho + oy) * Wo + ox] * x[((size_t)b * cin + ci) * ni + ((size_t)iz * H + iy) * W + ix];
                    }
              } else {               // out[2 i - 1 + k] += in[i] w[ci][co][k]
                for (int iz = 0; iz < D; ++iz)
                  for (int iy = 0; iy < H; ++iy)
                    for (int ix = 0; ix < W; ++ix) {
                      const int oz = 2 * iz - 1 + kz, oy = 2 * iy - 1 + ky, ox = 2 * ix - 1 + kx;
                      if (oz < 0 || oz >= Do || oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
                      acc += (double)x[((size_t)b * cin + ci) * ni + ((size_t)iz * H + iy) * W + ix] * gy[((size_t)b * cout + co) * no + ((size_t)oz * Ho + oy) * Wo + ox];
                    }
              }
            }
            const int t = (kz * g.KS + ky) * g.KS + kx;
            const float v = g.transposed ? got[((size_t)ci * cout + co) * T + t] : got[((size_t)co * cin + ci) * T + t];
            range = std::fmax(range, std::fabs(acc));
            err = std::fmax(err, std::isfinite(v) ? std::fabs(acc - v) : 1e30);
          }
  std::free(xa); std::free(ga); std::free(ws);
  printf("wgrad %-8s B=%d %d -> %d input %dx%dx%d: max error / largest gradient = %.2e; second run %s\n", name, B, cin, cout, D, H, W, err / range,
         same ? "bit-identical" : "DIFFERENT");
  return same ? err / range : 1.0;
}

static double sums_check(int N, int C, int n) {
  std::vector<float> x((size_t)N * C * n);
  for (auto &v : x) v = rnd() * 3.0f + 0.5f;
  const int blocks = casmvs_channel_sums_blocks(N, (size_t)n);
  std::vector<double> out((size_t)C * blocks * 2, NAN);
  float *xa = (float *)std::aligned_alloc(256, (x.size() * 4 + 255) & ~(size_t)255);
  std::memcpy(xa, x.data(), x.size() * 4);
  if (casmvs_channel_sums_f64(xa, out.data(), N, C, (size_t)n, nullptr)) { printf("channel_sums: %s\n", casmvs_last_error()); return 1e9; }
  double err = 0;
  for (int c = 0; c < C; ++c) {
    double s0 = 0, s1 = 0, g0 = 0, g1 = 0;
    for (int i = 0; i < N; ++i)
      for (int e = 0; e < n; ++e) {
        const double v = x[((size_t)i * C + c) * n + e];
        s0 += v;
        s1 += v * v;
      }
    for (int b = 0; b < blocks; ++b) { g0 += out[((size_t)c * blocks + b) * 2]; g1 += out[((size_t)c * blocks + b) * 2 + 1]; }
    err = std::fmax(err, std::fmax(std::fabs(g0 - s0) / std::fabs(s0), std::fabs(g1 - s1) / s1));
  }
  std::free(xa);
  printf("channel_sums N=%d C=%d n=%d (%d blocks): max relative error = %.2e\n", N, C, n, blocks, err);
  return err;
}

// casmvs_pack_gather_batch_f32 (every layer image of a training step in one launch) against casmvs_pack_gather_f32 per segment: ragged segment sizes, a
// segment without a bias, the constants 0 / 1 of the virtual source vector
static double pack_batch_check() {
  const int nseg = 5, sizes[nseg] = {300, 256, 1, 777, 2049}, nw[nseg] = {40, 7, 3, 500, 64}, nb[nseg] = {8, 0, 2, 16, 0};
  std::vector<std::vector<float>> w(nseg), b(nseg), want(nseg), got(nseg);
  std::vector<std::vector<int>> idx(nseg);
  std::vector<casmvs_pack_segment> segs(nseg);
  int block = 0;
  for (int s = 0; s < nseg; ++s) {
    w[s].resize(nw[s]); b[s].resize(nb[s] ? nb[s] : 1); idx[s].resize(sizes[s]); want[s].assign(sizes[s], NAN); got[s].assign(sizes[s], NAN);
    for (auto &v : w[s]) v = rnd();
    for (auto &v : b[s]) v = rnd() + 10.0f;
    for (int i = 0; i < sizes[s]; ++i) idx[s][i] = (int)((unsigned)(i * 2654435761u + s * 97u) % (unsigned)(nw[s] + nb[s] + 2));
    if (casmvs_pack_gather_f32(w[s].data(), nb[s] ? b[s].data() : nullptr, idx[s].data(), want[s].data(), nw[s], nb[s], sizes[s], nullptr)) return 1e9;
    segs[s] = casmvs_pack_segment{w[s].data(), nb[s] ? b[s].data() : nullptr, idx[s].data(), got[s].data(), nw[s], nb[s], sizes[s], block};
    block += (sizes[s] + 255) / 256;
  }
  if (casmvs_pack_gather_batch_f32(segs.data(), nseg, block, nullptr)) { printf("pack_gather_batch: %s\n", casmvs_last_error()); return 1e9; }
  int bad = 0;
  for (int s = 0; s < nseg; ++s)
    for (int i = 0; i < sizes[s]; ++i) bad += std::memcmp(&want[s][i], &got[s][i], 4) != 0;
  printf("pack_gather_batch: %d segments in %d workgroups, %d differing elements\n", nseg, block, bad);
  return bad ? 1.0 : 0.0;
}

// Train-mode ABN with the per-channel epilogue inside the elementwise kernels (casmvs_abn_train_apply_f32, casmvs_abn_backward_apply_fused_f32): forward values,
// folded constants, running statistics, input gradient and parameter gradients against float64 - 16-byte and scalar paths (n % 4), several chunks per plane
static double abn_check(int N, int C, int n) {
  const size_t total = (size_t)N * C * n;
  std::vector<float> x(total), gy(total), w(C), b(C), rm(C, 0.25f), rv(C, 2.0f);
  for (auto &v : x) v = rnd() * 2.0f + 0.3f;
  for (auto &v : gy) v = rnd();
  for (int c = 0; c < C; ++c) { w[c] = (c % 2 ? -1.0f : 1.0f) * (0.5f + 0.1f * c); b[c] = 0.05f * (c - 2); }
  const float eps = 1e-5f, momentum = 0.1f, slope = 0.01f, abs_eps = 1e-5f;
  const int blocks = casmvs_channel_sums_blocks(N, (size_t)n);
  const double M = (double)N * n;
  auto al = [](size_t floats) { return (float *)std::aligned_alloc(256, (floats * 4 + 255) & ~(size_t)255); };
  float *xa = al(total), *ga = al(total), *ya = al(total), *gxa = al(total);
  std::memcpy(xa, x.data(), total * 4); std::memcpy(ga, gy.data(), total * 4);
  std::vector<double> part((size_t)C * blocks * 2, NAN);
  std::vector<float> vec(4 * (size_t)C, NAN), pg(2 * (size_t)C, NAN);
  if (casmvs_channel_sums_f64(xa, part.data(), N, C, (size_t)n, nullptr) ||
      casmvs_abn_train_apply_f32(xa, part.data(), blocks, M, w.data(), b.data(), abs_eps, eps, momentum, rm.data(), rv.data(), vec.data(), vec.data() + C,
                                 vec.data() + 2 * C, vec.data() + 3 * C, ya, N, C, (size_t)n, slope, nullptr) ||
      casmvs_abn_backward_sums_f64(ga, ya, xa, vec.data() + 2 * C, vec.data() + 3 * C, part.data(), N, C, (size_t)n, slope, nullptr) ||
      casmvs_abn_backward_apply_fused_f32(ga, ya, xa, part.data(), blocks, M, w.data(), abs_eps, vec.data(), vec.data() + 2 * C, vec.data() + 3 * C, pg.data(),
                                          pg.data() + C, gxa, N, C, (size_t)n, slope, nullptr)) {
    printf("abn: %s\n", casmvs_last_error());
    return 1e9;
  }
  double err = 0;
  for (int c = 0; c < C; ++c) {
    double s0 = 0, s1 = 0;
    for (int i = 0; i < N; ++i)
      for (int e = 0; e < n; ++e) { const double v = x[((size_t)i * C + c) * n + e]; s0 += v; s1 += v * v; }
    const double mu = s0 / M, var = s1 / M - mu * mu, rs = 1.0 / std::sqrt(var + eps), g = std::fabs((double)w[c]) + abs_eps;
    err = std::fmax(err, std::fabs(vec[2 * C + c] - mu) / (std::fabs(mu) + 1e-3));
    err = std::fmax(err, std::fabs(vec[3 * C + c] - rs) / rs);
    err = std::fmax(err, std::fabs(rm[c] - (0.25 * 0.9 + 0.1 * mu)) / 0.25);
    err = std::fmax(err, std::fabs(rv[c] - (2.0 * 0.9 + 0.1 * var * M / (M - 1))) / 2.0);
    double gs0 = 0, gs1 = 0, ymax = 0, gmax = 0;
    std::vector<double> gq((size_t)N * n), xh((size_t)N * n);
    for (int i = 0; i < N; ++i)
      for (int e = 0; e < n; ++e) {
        const size_t o = ((size_t)i * C + c) * n + e;
        const double xhat = (x[o] - mu) * rs, pre = xhat * g + b[c], yv = pre > 0 ? pre : pre * slope;
        ymax = std::fmax(ymax, std::fabs(yv));
        err = std::fmax(err, std::fabs(ya[o] - yv) / (std::fabs(yv) + 1.0));
        const double gg = gy[o] * (ya[o] > 0 ? 1.0 : slope);
        gq[(size_t)i * n + e] = gg; xh[(size_t)i * n + e] = xhat;
        gs0 += gg; gs1 += gg * xhat;
      }
    err = std::fmax(err, std::fabs(pg[C + c] - gs0) / (std::fabs(gs0) + 1.0));
    err = std::fmax(err, std::fabs(pg[c] - gs1 * (w[c] > 0 ? 1.0 : -1.0)) / (std::fabs(gs1) + 1.0));
    for (int i = 0; i < N; ++i)
      for (int e = 0; e < n; ++e) {
        const size_t o = ((size_t)i * C + c) * n + e;
        const double want = g * rs * (gq[(size_t)i * n + e] - gs0 / M - xh[(size_t)i * n + e] * gs1 / M);
        gmax = std::fmax(gmax, std::fabs(want));
        err = std::fmax(err, std::fabs(gxa[o] - want) / (std::fabs(want) + 1.0));
      }
  }
  std::free(xa); std::free(ga); std::free(ya); std::free(gxa);
  printf("abn_fused N=%d C=%d n=%d (%d blocks): max error = %.2e\n", N, C, n, blocks, err);
  return err;
}

// casmvs_costvol_var_backward_f32 (volume_absmax_kernel -> costvol_var_bwd_kernel: the scatter transpose of the plane sweep through a 64-bit fixed-point LDS
// box image into a 64-bit fixed-point gradient map -> costvol_fixed_finish_kernel; the largest kernel of the training step) against
// d var / d x_v = 2 x_v / V - 2 sum x / V^2 through the bilinear weights, taps from the shared float32 routine, sums in float64 - and a SECOND run with the
// workgroups in the opposite order, which must give the same bits (integer sums: train.py:99-127 reproducible run to run)
// mode 1: the source views' features are 1e4 x the reference view's; mode 2: reference features all zero; mode 3: upstream gradients of 1e-30;
// mode 4: upstream gradients of 1e30 with features of 1e-3; mode 5: one feature of 3e5 (its channel's scale is 1e5 x the others'); mode 6: a NaN upstream
// gradient and an infinite feature - exactly the elements the float64 sums make non-finite must be non-finite, every other one at the bound
// G > 0: the group-wise correlation volume's backward (casmvs_costvol_gwc_backward_f32), gvol (B, G, D, h, w)
static double varbwd_check(int B, int V, int C, int D, int h, int w, int mode = 0, int G = 0) {
  const size_t hw = (size_t)h * w;
  const int GC = G > 0 ? G : C, cpg = G > 0 ? C / G : 1;
  std::vector<float> feats((size_t)B * V * C * hw), proj((size_t)B * (V - 1) * 12, 0.0f), depth((size_t)B * D * hw), gvol((size_t)B * GC * D * hw);
  for (auto &v : feats) v = rnd();
  for (auto &v : gvol) v = rnd() * (mode == 3 ? 1e-30f : mode == 4 ? 1e30f : 1.0f);
  if (mode == 4) for (auto &v : feats) v *= 1e-3f;
  if (mode == 5) feats[((size_t)1 * C + 3) * hw + 2 * w + 7] = 3e5f;
  if (mode == 6) {
    gvol[((size_t)(GC > 2 ? 2 : 0) * D + 1) * hw + 3 * w + 9] = NAN;
    feats[((size_t)1 * C + 1) * hw + 4 * w + 20] = INFINITY;
  }
  if (mode == 1 || mode == 2)
    for (int b = 0; b < B; ++b)
      for (size_t i = 0; i < (size_t)C * hw; ++i) {
        float &r = feats[(size_t)b * V * C * hw + i];
        r = mode == 1 ? r * 1e-4f : 0.0f;
      }
  for (int b = 0; b < B; ++b) {
    for (int v = 0; v < V - 1; ++v) {
      float *P = proj.data() + ((size_t)b * (V - 1) + v) * 12;
      P[0] = 1.0f; P[5] = 1.0f; P[10] = 1.0f;
      P[1] = 0.002f * (v + 1); P[4] = -0.002f * (v + 1);
      P[3] = (v % 2 ? -1.0f : 1.0f) * 0.6f / 1.376e-5f;   // 0.6 pixels of epipolar slide per plane (run_kernels7.cpp)
      P[2] = -P[3] / 425.0f - 2.0f;
      P[7] = 0.3f * 425.0f * (v + 1);
    }
    for (int d = 0; d < D; ++d)
      for (size_t p = 0; p < hw; ++p) depth[((size_t)b * D + d) * hw + p] = 425.0f + 2.5f * d + 0.02f * (float)(p % 5);
  }
  auto dup = [](const std::vector<float> &v) {
    float *p = (float *)std::aligned_alloc(256, (v.size() * 4 + 255) & ~(size_t)255);
    std::memcpy(p, v.data(), v.size() * 4);
    return p;
  };
  float *fa = dup(feats), *pa = dup(proj), *da = dup(depth), *ga = dup(gvol);
  std::vector<float> nanv(feats.size(), NAN);
  float *out = dup(nanv), *out2 = dup(nanv);
  const size_t wsb = casmvs_costvol_backward_workspace_bytes(B, V, C, G, D, h, w);
  void *ws = std::aligned_alloc(256, (wsb + 255) & ~(size_t)255);
  std::memset(ws, 0xCD, wsb);   // the call owns the zeroing
  auto run = [&](float *o) {
    return G > 0 ? casmvs_costvol_gwc_backward_f32(fa, pa, da, ga, o, ws, B, V, C, G, h, w, D, nullptr)
                 : casmvs_costvol_var_backward_f32(fa, pa, da, ga, o, ws, B, V, C, h, w, D, nullptr);
  };
  if (run(out)) { printf("var_backward: %s\n", casmvs_last_error()); return 1e9; }
  hipemu::g_reverse_blocks = true;
  const int rc2 = run(out2);
  hipemu::g_reverse_blocks = false;
  if (rc2) { printf("var_backward (second run): %s\n", casmvs_last_error()); return 1e9; }
  const bool same = std::memcmp(out, out2, feats.size() * 4) == 0;
  std::vector<double> want(feats.size(), 0.0);
  for (int b = 0; b < B; ++b)
    for (int d = 0; d < D; ++d)
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
          const float dv = depth[((size_t)b * D + d) * hw + (size_t)y * w + x];
          std::vector<casmvs_dev::Taps> taps(V - 1);
          std::vector<std::vector<double>> xv(V - 1, std::vector<double>(C));
          std::vector<double> S(C);
          for (int c = 0; c < C; ++c) S[c] = feats[(((size_t)b * V) * C + c) * hw + (size_t)y * w + x];
          for (int v = 0; v < V - 1; ++v) {
            taps[v] = casmvs_dev::plane_sweep_taps(proj.data() + ((size_t)b * (V - 1) + v) * 12, (float)x, (float)y, dv, w, h);
            const casmvs_dev::Taps &t = taps[v];
            for (int c = 0; c < C; ++c) {
              const float *f = feats.data() + (((size_t)b * V + 1 + v) * C + c) * hw;
              xv[v][c] = (double)t.w_nl * f[(size_t)t.yn * w + t.xl] + (double)t.w_nr * f[(size_t)t.yn * w + t.xl + 1] + (double)t.w_sl * f[(size_t)t.ys * w + t.xl] +
                         (double)t.w_sr * f[(size_t)t.ys * w + t.xl + 1];
              S[c] += xv[v][c];
            }
          }
          for (int c = 0; c < C; ++c) {
            const double g = gvol[(((size_t)b * GC + c / cpg) * D + d) * hw + (size_t)y * w + x];
            const double xr = feats[(((size_t)b * V) * C + c) * hw + (size_t)y * w + x];
            const double kg = 1.0 / ((double)cpg * (V - 1));
            want[(((size_t)b * V) * C + c) * hw + (size_t)y * w + x] += G > 0 ? g * kg * (S[c] - xr) : g * (2.0 * xr / V - 2.0 * S[c] / ((double)V * V));
            for (int v = 0; v < V - 1; ++v) {
              const casmvs_dev::Taps &t = taps[v];
              const double k = G > 0 ? g * kg * xr : g * (2.0 * xv[v][c] / V - 2.0 * S[c] / ((double)V * V));
              double *q = want.data() + (((size_t)b * V + 1 + v) * C + c) * hw;
              q[(size_t)t.yn * w + t.xl] += k * t.w_nl;
              q[(size_t)t.yn * w + t.xl + 1] += k * t.w_nr;
              q[(size_t)t.ys * w + t.xl] += k * t.w_sl;
              q[(size_t)t.ys * w + t.xl + 1] += k * t.w_sr;
            }
          }
        }
  double err = 0, range = 0;
  size_t poisoned = 0;
  for (size_t i = 0; i < want.size(); ++i) {
    if (!std::isfinite(want[i])) {   // (mode 6) a non-finite sum: the kernel's element must be non-finite too
      ++poisoned;
      if (std::isfinite(out[i])) err = 1e30;
      continue;
    }
    range = std::fmax(range, std::fabs(want[i]));
    err = std::fmax(err, std::isfinite(out[i]) ? std::fabs(want[i] - out[i]) : 1e30);
  }
  std::free(fa); std::free(pa); std::free(da); std::free(ga); std::free(out); std::free(out2); std::free(ws);
  printf("%s B=%d V=%d C=%d %dx%dx%d mode %d: max error / largest gradient = %.2e (%zu non-finite elements), workgroups in reverse order: %s\n",
         G > 0 ? "gwc_backward" : "var_backward", B, V, C, D, h, w, mode, err / range, poisoned, same ? "bit-identical" : "DIFFERENT");
  return same ? err / range : 1e9;
}

int main(int argc, char **argv) {
  hipemu::g_lds = smem_raw;   // CASMVS_DYNAMIC_LDS (common.h)
  const std::string which = argc > 1 ? argv[1] : "all";
  double worst = 0;
  auto take = [&](double e) { worst = std::fmax(worst, e); };
  const bool all = which == "all", quick = which == "quick";
  if (all || quick) {
    take(wgrad_check("S1", CASMVS_CONV_S1, 1, 8, 8, 2, 3, 20));          // two tiles, ragged in z (2 of 4 planes), y (3 of 4 rows) and x (20 of 32)
    take(wgrad_check("K5S2", CASMVS_CONV2D_K5S2, 1, 8, 16, 1, 8, 16));
    take(sums_check(2, 8, 1000));
    take(pack_batch_check());
    take(abn_check(2, 5, 1003));                                          // scalar path, one chunk
    take(abn_check(1, 3, 4608));                                          // 16-byte path, three chunks (the last one short)
    take(varbwd_check(1, 3, 8, 8, 6, 36));                                // two 32 x 32 tiles (ragged), one chunk of 8 planes, two source views
    take(varbwd_check(1, 3, 4, 8, 6, 36, 1));                             // source views 1e4 x the reference view
    take(varbwd_check(1, 3, 8, 8, 6, 36, 6));                             // non-finite contributions: the float map takes them, the finish adds the integer sums
    take(varbwd_check(1, 3, 8, 8, 6, 36, 0, 4));                          // group-wise correlation: two channels per group
  }
  if (all) {
    take(wgrad_check("S1", CASMVS_CONV_S1, 1, 8, 8, 5, 6, 20));          // ragged in z (tile 4), y and x
    take(wgrad_check("S2", CASMVS_CONV_S2, 1, 8, 16, 4, 6, 12));
    take(wgrad_check("T2", CASMVS_CONV_T2, 1, 16, 8, 2, 3, 10));
    take(wgrad_check("S1", CASMVS_CONV_S1, 2, 16, 16, 3, 5, 9));
    take(wgrad_check("S1", CASMVS_CONV_S1, 1, 8, 1, 4, 6, 12));          // `prob` on the generic kernel: one output channel in a 16-row tile
    take(wgrad_check("K3", CASMVS_CONV2D_K3, 2, 8, 8, 1, 9, 20));
    take(wgrad_check("K1", CASMVS_CONV2D_K1, 1, 32, 16, 1, 6, 10));
    take(wgrad_check("T2", CASMVS_CONV_T2, 1, 32, 16, 1, 4, 6));
    take(sums_check(1, 16, 70000));
    take(abn_check(1, 2, 40000));                                         // three partial-sum blocks per channel
    take(varbwd_check(1, 3, 8, 8, 12, 36));
    take(varbwd_check(2, 2, 16, 16, 34, 40));                             // four channel groups, two plane chunks, four tiles
    take(varbwd_check(1, 3, 8, 8, 6, 36, 2));
    take(varbwd_check(1, 2, 4, 5, 6, 36, 3));
    take(varbwd_check(1, 3, 8, 8, 6, 36, 4));
    take(varbwd_check(1, 3, 8, 8, 12, 36, 5));
    take(varbwd_check(1, 3, 8, 8, 6, 36, 6, 4));
    take(varbwd_check(2, 4, 16, 9, 20, 40, 0, 16));                       // one channel per group, three source views (run-time view loop)
    take(varbwd_check(1, 3, 8, 8, 12, 36, 0, 1));                         // one group
  }
  printf(worst < 3e-6 ? "ALL OK (worst %.2e)\n" : "FAILED (worst %.2e)\n", worst);
  return worst < 3e-6 ? 0 : 1;
}
