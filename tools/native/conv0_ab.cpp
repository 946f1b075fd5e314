// A/B of two builds of libcasmvs_hip.so on CostRegNet.conv0 (split-f16 form) through the C ABI, without torch / Python:
//   conv0_ab <libA.so> <libB.so> [batch]
// For the three cascade levels' shapes (cin 32 / 16 / 8): B's output against A's (equal bits, or the largest difference relative to
// the output range), each build's time per launch with 512 MB of other traffic between timed launches (the state inside a forward),
// and on a small ragged shape both against a float64 convolution on the host.  A variant build: tools/build_variant.py.
//   hipcc -O2 tools/native/conv0_ab.cpp -Iinclude -ldl -o tools/probes/bin/conv0_ab
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef size_t (*packed_bytes_fn)(int);
typedef int (*pack_fn)(int, const float *, const float *, const float *, void *);
typedef int (*forward_fn)(const void *, const float *, float *, int, int, int, int, int, float, int, void *);
typedef const char *(*error_fn)(void);

struct Lib {
  void *h = nullptr;
  packed_bytes_fn packed_bytes;
  pack_fn pack;
  forward_fn forward;
  error_fn last_error;
  bool open(const char *path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen %s: %s\n", path, dlerror()); return false; }
    packed_bytes = (packed_bytes_fn)dlsym(h, "casmvs_conv0_splitf16_packed_bytes");
    pack = (pack_fn)dlsym(h, "casmvs_conv0_splitf16_pack");
    forward = (forward_fn)dlsym(h, "casmvs_conv0_splitf16_forward_f32");
    last_error = (error_fn)dlsym(h, "casmvs_last_error");
    return packed_bytes && pack && forward && last_error;
  }
};

static uint32_t g_rng = 88172645u;
static float rnd() {
  g_rng ^= g_rng << 13; g_rng ^= g_rng >> 17; g_rng ^= g_rng << 5;
  return (float)(int32_t)g_rng * (1.0f / 2147483648.0f);
}

int main(int argc, char **argv) {
  if (argc < 3) { printf("usage: conv0_ab libA.so libB.so [batch]\n"); return 2; }
  const int batch = argc > 3 ? atoi(argv[3]) : 2;
  Lib L[2];
  if (!L[0].open(argv[1]) || !L[1].open(argv[2])) return 2;
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  void *dirty = nullptr;
  const size_t dirty_bytes = (size_t)512 << 20;
  hipMalloc(&dirty, dirty_bytes);
  struct Shape { int B, cin, D, H, W; bool host; };
  const Shape shapes[] = {{1, 16, 5, 9, 44, true}, {batch, 32, 48, 128, 160, false}, {batch, 16, 32, 256, 320, false}, {batch, 8, 8, 512, 640, false}};
  bool all_ok = true;
  for (const Shape &s : shapes) {
    const size_t n = (size_t)s.D * s.H * s.W, nin = (size_t)s.B * s.cin * n, nout = (size_t)s.B * 8 * n;
    std::vector<float> x(nin), w((size_t)8 * s.cin * 27), scale(8), shift(8);
    for (auto &v : x) v = rnd() * 3.0f + 0.5f;
    for (auto &v : w) v = rnd() * 0.2f;
    for (int c = 0; c < 8; ++c) { scale[c] = 0.5f + 0.1f * c; shift[c] = 0.05f * (c - 4); }
    float *dx, *dy[2];
    hipMalloc(&dx, nin * 4);
    hipMemcpy(dx, x.data(), nin * 4, hipMemcpyHostToDevice);
    std::vector<float> y[2];
    double us[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
      const size_t pb = L[k].packed_bytes(s.cin);
      std::vector<unsigned char> packed(pb);
      if (L[k].pack(s.cin, w.data(), scale.data(), shift.data(), packed.data())) { printf("pack: %s\n", L[k].last_error()); return 3; }
      void *dp;
      hipMalloc(&dp, pb);
      hipMemcpy(dp, packed.data(), pb, hipMemcpyHostToDevice);
      hipMalloc(&dy[k], nout * 4);
      hipMemset(dy[k], 0xff, nout * 4);
      if (L[k].forward(dp, dx, dy[k], s.B, s.cin, s.D, s.H, s.W, 0.01f, 0, st)) { printf("forward: %s\n", L[k].last_error()); return 3; }
      hipStreamSynchronize(st);
      y[k].resize(nout);
      hipMemcpy(y[k].data(), dy[k], nout * 4, hipMemcpyDeviceToHost);
      const int reps = 6;
      float total = 0;
      for (int i = 0; i < reps; ++i) {
        hipMemsetAsync(dirty, i, dirty_bytes, st);
        hipEventRecord(e0, st);
        L[k].forward(dp, dx, dy[k], s.B, s.cin, s.D, s.H, s.W, 0.01f, 0, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        total += ms;
      }
      us[k] = total * 1e3 / reps;
      hipFree(dp);
    }
    double range = 0, diff = 0;
    size_t ndiff = 0;
    for (size_t i = 0; i < nout; ++i) {
      range = std::fmax(range, std::fabs((double)y[0][i]));
      const double d = std::fabs((double)y[0][i] - y[1][i]);
      diff = std::fmax(diff, d);
      ndiff += memcmp(&y[0][i], &y[1][i], 4) != 0;
    }
    printf("B=%d cin=%d %dx%dx%d: A %.1f us, B %.1f us (x%.3f); B vs A: %zu of %zu values differ, max |diff| / range = %.2e", s.B, s.cin, s.D, s.H, s.W,
           us[0], us[1], us[0] / us[1], ndiff, nout, diff / range);
    bool ok = std::isfinite(diff) && diff / range < 2e-6;
    if (s.host) {
      double err[2] = {0, 0};
      for (int b = 0; b < s.B; ++b)
        for (int co = 0; co < 8; ++co)
          for (int z = 0; z < s.D; ++z)
            for (int yy = 0; yy < s.H; ++yy)
              for (int xx = 0; xx < s.W; ++xx) {
                double acc = 0;
                for (int ci = 0; ci < s.cin; ++ci)
                  for (int kz = 0; kz < 3; ++kz)
                    for (int ky = 0; ky < 3; ++ky)
                      for (int kx = 0; kx < 3; ++kx) {
                        const int iz = z + kz - 1, iy = yy + ky - 1, ix = xx + kx - 1;
                        if (iz < 0 || iz >= s.D || iy < 0 || iy >= s.H || ix < 0 || ix >= s.W) continue;
                        acc += (double)w[(((size_t)co * s.cin + ci) * 27) + kz * 9 + ky * 3 + kx] * x[((size_t)b * s.cin + ci) * n + ((size_t)iz * s.H + iy) * s.W + ix];
                      }
                double v = acc * scale[co] + shift[co];
                v = v > 0 ? v : v * 0.01f;
                const size_t o = ((size_t)b * 8 + co) * n + ((size_t)z * s.H + yy) * s.W + xx;
                for (int k = 0; k < 2; ++k) err[k] = std::fmax(err[k], std::fabs(v - y[k][o]));
              }
      printf("; vs float64: A %.2e  B %.2e of the range", err[0] / range, err[1] / range);
      ok = ok && err[0] / range < 2e-6 && err[1] / range < 2e-6;
    }
    printf("  %s\n", ok ? "ok" : "FAILED");
    all_ok &= ok;
    hipFree(dx); hipFree(dy[0]); hipFree(dy[1]);
  }
  printf(all_ok ? "ALL OK\n" : "FAILURES\n");
  return all_ok ? 0 : 1;
}
