// Phase timeline of conv11_prob_zfused_kernel (csrc/conv11_prob_zfused.hip, all waves in step) from the shader clock: needs a profiling build of the library
//   python tools/build_variant.py zftrace -DCASMVS_ZF_TRACE=1 -DCASMVS_ZF_WS=0 ;  LD_PRELOAD=casmvsnet_pl_amd/libcasmvs_zftrace.so tools/probes/bin/zf_trace
// Waves 0 (5 matrix units) and 7 (4) of workgroup (0, 0), six stamps per plane step: t0 top, t1 matrix instructions issued, t2 epilogue done (skip values
// landed, slot written), t3 next loads issued + box ring work, t4 past the barrier, t5 `prob`'s multiply phase done; the next t0 follows the cost store.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "casmvs.h"

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, Di = 16, Hi = 128, Wi = 160;
  typedef int (*read_fn)(unsigned long long *);
  read_fn rd = (read_fn)dlsym(RTLD_DEFAULT, "casmvs_zf_trace_read");
  if (!rd) { printf("no casmvs_zf_trace_read in the loaded library (build with -DCASMVS_ZF_TRACE=1 -DCASMVS_ZF_WS=0 and LD_PRELOAD it)\n"); return 2; }
  const int D = 2 * Di, H = 2 * Hi, W = 2 * Wi;
  const size_t ni = (size_t)Di * Hi * Wi, no = (size_t)D * H * W;
  std::vector<float> x((size_t)B * 16 * ni), w11(16 * 8 * 27, 0.02f), wp(8 * 27, 0.05f), bias(1, 0.1f);
  for (size_t i = 0; i < x.size(); ++i) x[i] = (float)((i * 2654435761u) >> 20 & 1023) * (1.0f / 512.0f) - 1.0f;
  std::vector<unsigned char> dpk(casmvs_deconv11_splitf16_packed_bytes());
  if (casmvs_deconv11_splitf16_pack(w11.data(), nullptr, nullptr, dpk.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
  std::vector<float> ppk(casmvs_conv3d_packed_floats(CASMVS_CONV_S1, 8, 1));
  if (casmvs_conv3d_pack_f32(CASMVS_CONV_S1, 8, 1, wp.data(), nullptr, bias.data(), ppk.data())) { printf("pack prob: %s\n", casmvs_last_error()); return 3; }
  float *dx, *dsk, *ddv, *dcost, *ddepth, *dconf, *dppk;
  void *ddpk, *dirty;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dsk, (size_t)B * 8 * no * 4); hipMalloc(&ddv, (size_t)B * no * 4); hipMalloc(&dcost, (size_t)B * no * 4);
  hipMalloc(&ddepth, (size_t)B * H * W * 4); hipMalloc(&dconf, (size_t)B * H * W * 4); hipMalloc(&dppk, ppk.size() * 4); hipMalloc(&ddpk, dpk.size());
  hipMalloc(&dirty, (size_t)512 << 20);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dsk, 0, (size_t)B * 8 * no * 4); hipMemset(ddv, 0, (size_t)B * no * 4);
  hipMemcpy(dppk, ppk.data(), ppk.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(ddpk, dpk.data(), dpk.size(), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(dirty, rep, (size_t)512 << 20);
    if (casmvs_conv11_prob_zfused_f32(ddpk, dppk, dx, dsk, ddv, dcost, ddepth, dconf, nullptr, B, Di, Hi, Wi, 0.01f, 1.0f, nullptr)) { printf("forward: %s\n", casmvs_last_error()); return 3; }
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> t(2 * 512);
  if (rd(t.data())) { printf("trace read failed\n"); return 3; }
  for (int wv = 0; wv < 2; ++wv) {
    const unsigned long long *c = t.data() + 512 * wv;
    for (int par = 0; par < 2; ++par) {
      double s[7] = {0, 0, 0, 0, 0, 0, 0};
      int cnt = 0;
      for (int u = 4 + par; u < D - 2; u += 2, ++cnt) {
        const unsigned long long *q = c + 6 * u;
        s[0] += (double)(q[6] - q[0]);
        for (int i = 0; i < 6; ++i) s[1 + i] += (double)(q[i + 1] - q[i]);
      }
      printf("wave %d, %s planes 4..%d (cycles of the stamp clock): step %.0f | matrix phase %.0f | epilogue %.0f | loads + box ring %.0f | barrier %.0f | prob multiply %.0f | cost store + rotate %.0f\n",
             wv ? 7 : 0, par ? "odd " : "even", D - 3, s[0] / cnt, s[1] / cnt, s[2] / cnt, s[3] / cnt, s[4] / cnt, s[5] / cnt, s[6] / cnt);
    }
  }
  printf("(s_memtime may run at 100 MHz on this GPU: then the numbers are units of 10 ns = 24 shader cycles at 2.4 GHz)\n");
  return 0;
}
