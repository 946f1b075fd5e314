// Companion of tools/probes/mfma_coresidency_repro.hip (whose synthetic victims did NOT reproduce the fault): the same synthetic NEIGHBOUR kernels on
// stream A, but the victims on stream B are the library's own float32 layer kernels through the C ABI, torch-free:
//   px   casmvs_conv3d_forward_f32 S1 16 -> 8   (conv16db_kernel<PX>: the kernel whose output was the first wrong tensor in round 3)
//   ci   casmvs_conv3d_forward_f32 S1 16 -> 16  (conv16db_kernel<CI>)
//   s2   casmvs_conv3d_forward_f32 S2 8 -> 16   (stride 2)
//   t2   casmvs_conv3d_forward_f32 T2 16 -> 8   (transposed + skip)
// Each victim launch is compared bit for bit with the victim's output when it ran alone.   coresidency_lib_victim [rounds = 200 [victims, e.g. px,ciw]]
// Another build of the library as the victim: LD_PRELOAD=casmvsnet_pl_amd/libcasmvs_trace.so CASMVS_NO_DB=1 ... (the profiling build's A/B switches)
//   hipcc -O2 --offload-arch=gfx950 tools/native/coresidency_lib_victim.cpp -Iinclude -Lcasmvsnet_pl_amd -lcasmvs_hip -Wl,-rpath,'$ORIGIN/../../../casmvsnet_pl_amd' -o tools/probes/bin/coresidency_lib_victim
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "casmvs.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int KIND>   // 1 f32 MFMA, 2 f16 MFMA, 3 bf16 MFMA, 4 VALU
__global__ __launch_bounds__(256) void neighbour(float *sink, int iters) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  const u32x4 ua = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  for (int it = 0; it < iters; ++it) {
    if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    else if (KIND == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, ua), acc, 0, 0, 0);
    else if (KIND == 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ua), acc, 0, 0, 0);
    else { acc[0] = fmaf(acc[0], a, b); acc[1] = fmaf(acc[1], a, b); }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

__global__ void compare(const unsigned *got, const unsigned *ref, size_t n, unsigned *counts) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (got[i] != ref[i]) atomicAdd(&counts[0], 1u);
}

static uint32_t g_rng = 2463534242u;
static float rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 17; g_rng ^= g_rng << 5; return (float)(int32_t)g_rng * (1.0f / 2147483648.0f); }

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipStream_t sa, sb;
  CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
  struct Victim { const char *name; int kind, cin, cout, B, D, H, W; };
  // px / ciw / s2w: the WIDE double-buffered forms (conv16db_kernel<.., CK 4, NT 4 | 2, ..>); ci / s2: the deep forms (NT 1) the small volumes select
  const Victim victims[] = {{"px", CASMVS_CONV_S1, 16, 8, 1, 32, 32, 48}, {"ci", CASMVS_CONV_S1, 16, 16, 1, 16, 32, 48},
                            {"ciw", CASMVS_CONV_S1, 16, 16, 1, 32, 64, 128}, {"s2", CASMVS_CONV_S2, 8, 16, 1, 32, 32, 48},
                            {"s2w", CASMVS_CONV_S2, 8, 16, 1, 64, 128, 128}, {"t2", CASMVS_CONV_T2, 16, 8, 1, 16, 16, 24}};
  const char *only = argc > 2 ? argv[2] : nullptr;   // comma list of victim names
  const char *nnames[5] = {"none", "f32mfma", "f16mfma", "bf16mfma", "valu"};
  float *dsink;
  unsigned *dcounts;
  CHECK(hipMalloc(&dsink, 64)); CHECK(hipMalloc(&dcounts, 4));
  auto launch_neighbour = [&](int k) {   // 512 workgroups x 3000 iterations x 6 launches: round 3's tools/debug/disturber.py (k1:1024)
    for (int rep = 0; rep < 6; ++rep) {
      const dim3 g(2 * cus), b(256);
      if (k == 1) hipLaunchKernelGGL(neighbour<1>, g, b, 0, sa, dsink, 1500);
      else if (k == 2) hipLaunchKernelGGL(neighbour<2>, g, b, 0, sa, dsink, 3000);
      else if (k == 3) hipLaunchKernelGGL(neighbour<3>, g, b, 0, sa, dsink, 3000);
      else if (k == 4) hipLaunchKernelGGL(neighbour<4>, g, b, 0, sa, dsink, 6000);
    }
  };
  printf("%s, %d CUs; %d rounds per pair; a round = 6 neighbour launches on stream A + 4 victim launches on stream B, the last compared with the victim alone\n",
         prop.gcnArchName, cus, rounds);
  int any = 0;
  for (const Victim &v : victims) {
    if (only) {
      const std::string list = std::string(",") + only + ",", key = std::string(",") + v.name + ",";
      if (list.find(key) == std::string::npos) continue;
    }
    const bool t2 = v.kind == CASMVS_CONV_T2, s2 = v.kind == CASMVS_CONV_S2;
    const size_t nvox = (size_t)v.D * v.H * v.W, nin = (size_t)v.B * v.cin * nvox;
    const size_t nout = (size_t)v.B * v.cout * (t2 ? nvox * 8 : (s2 ? nvox / 8 : nvox));
    std::vector<float> x(nin), w((size_t)v.cin * v.cout * 27), sc(v.cout), sh(v.cout), sk(nout);
    for (auto &e : x) e = rnd() * 0.3f + 0.15f;
    for (auto &e : w) e = rnd() * 0.2f;
    for (auto &e : sk) e = rnd();
    for (int c = 0; c < v.cout; ++c) { sc[c] = 0.6f + 0.05f * c; sh[c] = 0.02f * (c - 4); }
    std::vector<float> pk(casmvs_conv3d_packed_floats(v.kind, v.cin, v.cout));
    if (pk.empty() || casmvs_conv3d_pack_f32(v.kind, v.cin, v.cout, w.data(), sc.data(), sh.data(), pk.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
    float *dx, *dpk, *dsk, *dout, *dref;
    CHECK(hipMalloc(&dx, nin * 4)); CHECK(hipMalloc(&dpk, pk.size() * 4)); CHECK(hipMalloc(&dsk, nout * 4)); CHECK(hipMalloc(&dout, nout * 4)); CHECK(hipMalloc(&dref, nout * 4));
    CHECK(hipMemcpy(dx, x.data(), nin * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dsk, sk.data(), nout * 4, hipMemcpyHostToDevice));
    auto run = [&]() {
      if (casmvs_conv3d_forward_f32(v.kind, dpk, dx, t2 ? dsk : nullptr, dout, v.B, v.cin, v.cout, v.D, v.H, v.W, 0.01f, sb)) { printf("forward: %s\n", casmvs_last_error()); exit(3); }
    };
    run();
    CHECK(hipStreamSynchronize(sb));
    CHECK(hipMemcpy(dref, dout, nout * 4, hipMemcpyDeviceToDevice));
    for (int k = 0; k < 5; ++k) {
      int bad = 0, dumps = 0;
      unsigned long long wrong = 0;
      for (int r = 0; r < rounds; ++r) {
        CHECK(hipMemsetAsync(dcounts, 0, 4, sb));
        launch_neighbour(k);
        for (int rep = 0; rep < 4; ++rep) run();   // several victim launches inside the neighbour's ~1.5 ms
        hipLaunchKernelGGL(compare, dim3(512), dim3(256), 0, sb, (const unsigned *)dout, (const unsigned *)dref, nout, dcounts);
        unsigned c = 0;
        CHECK(hipMemcpyAsync(&c, dcounts, 4, hipMemcpyDeviceToHost, sb));
        CHECK(hipStreamSynchronize(sb));
        CHECK(hipStreamSynchronize(sa));
        bad += c > 0;
        wrong += c;
        if (c > 0 && getenv("CORES_DUMP") && dumps < atoi(getenv("CORES_DUMP"))) {   // where and how wrong: the first mismatches of a failing round + histograms
          ++dumps;
          std::vector<float> got(nout), ref(nout);
          CHECK(hipMemcpy(got.data(), dout, nout * 4, hipMemcpyDeviceToHost));
          CHECK(hipMemcpy(ref.data(), dref, nout * 4, hipMemcpyDeviceToHost));
          const int Wd = v.W, Hd = v.H, Dd = v.D;
          std::vector<int> hc(v.cout, 0), hx(32, 0), hy(8, 0), hz(8, 0);
          int shown = 0;
          for (size_t i = 0; i < nout; ++i) {
            if (memcmp(&got[i], &ref[i], 4) == 0) continue;
            const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd), z = (int)((i / ((size_t)Wd * Hd)) % Dd), ch = (int)(i / ((size_t)Wd * Hd * Dd));
            ++hc[ch]; ++hx[x % 32]; ++hy[y % 8]; ++hz[z % 8];
            if (shown++ < 12) printf("    c %d z %d y %d x %d: got %.6g want %.6g diff %.3g\n", ch, z, y, x, got[i], ref[i], got[i] - ref[i]);
          }
          printf("    by channel:"); for (int e : hc) printf(" %d", e);
          printf("\n    by x %% 32:"); for (int e : hx) printf(" %d", e);
          printf("\n    by y %% 8:"); for (int e : hy) printf(" %d", e);
          printf("\n    by z %% 8:"); for (int e : hz) printf(" %d", e);
          printf("\n");
        }
      }
      printf("victim %-3s (kind %d, %d -> %d, %dx%dx%d) beside %-8s: %4d of %4d rounds differ from the solo run", v.name, v.kind, v.cin, v.cout, v.D, v.H, v.W, nnames[k], bad, rounds);
      if (bad) printf("  (%llu wrong values)", wrong);
      printf("\n");
      any |= bad > 0;
    }
    hipFree(dx); hipFree(dpk); hipFree(dsk); hipFree(dout); hipFree(dref);
  }
  printf(any ? "REPRODUCED with the library's kernels as victims\n" : "not reproduced: every victim launch equals its solo run\n");
  return any ? 1 : 0;
}
