// Phase timeline of conv0_zw_kernel (csrc/conv0_zmarch.hip) from the shader clock: needs a -DCASMVS_ZW_TRACE build of the library
//   python tools/build_variant.py zwtrace -DCASMVS_ZW_TRACE=1 ;  LD_PRELOAD=casmvsnet_pl_amd/libcasmvs_zwtrace.so tools/probes/bin/zw_trace
// Workgroup 0, wave 0 of each role: consumer stamps c0 (unit top) c1 (36 of 108 matrix instructions issued) c2 (past barrier A) c3 (all issued)
// c4 (folds done) per unit; producer stamps p0 (top) p1 (loads landed, maximum published) p2 (past A) p3 (split + LDS writes issued) p4 (before B).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "casmvs.h"

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, cin = 32, D = 48, H = 128, W = 160;
  typedef int (*read_fn)(unsigned long long *);
  read_fn rd = (read_fn)dlsym(RTLD_DEFAULT, "casmvs_zw_trace_read");
  if (!rd) { printf("no casmvs_zw_trace_read in the loaded library (build with -DCASMVS_ZW_TRACE=1 and LD_PRELOAD it)\n"); return 2; }
  const size_t n = (size_t)D * H * W;
  std::vector<float> x((size_t)B * cin * n, 0.25f), w((size_t)8 * cin * 27, 0.01f);
  for (size_t i = 0; i < x.size(); ++i) x[i] = (float)((i * 2654435761u) >> 20 & 1023) * (1.0f / 512.0f) - 1.0f;
  std::vector<unsigned char> pk(casmvs_conv0_splitf16_packed_bytes(cin));
  if (casmvs_conv0_splitf16_pack(cin, w.data(), nullptr, nullptr, pk.data())) { printf("pack: %s\n", casmvs_last_error()); return 3; }
  float *dx, *dy;
  void *dp, *dirty;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dy, (size_t)B * 8 * n * 4); hipMalloc(&dp, pk.size()); hipMalloc(&dirty, (size_t)512 << 20);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dp, pk.data(), pk.size(), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(dirty, rep, (size_t)512 << 20);
    if (casmvs_conv0_zmarch_forward_f32(dp, dx, dy, B, cin, D, H, W, 0.01f, nullptr)) { printf("forward: %s\n", casmvs_last_error()); return 3; }
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> t(2 * 512);
  if (rd(t.data())) { printf("trace read failed\n"); return 3; }
  const unsigned long long *c = t.data(), *p = t.data() + 512;
  // consumer: 3 stamps per unit (c0 top, c1 all 108 matrix instructions issued, c2 folds done); producer: 4 per iteration (p0 top, p1 split + LDS writes
  // issued, p2 the next unit's loads landed + its maximum published, p3 the loads of unit n + NSET issued)
  double s[4] = {0, 0, 0, 0}, r[5] = {0, 0, 0, 0, 0};
  int cnt = 0;
  for (int u = 4; u < 44; ++u, ++cnt) {
    const unsigned long long *q = c + 3 * u;
    s[0] += (double)(q[3] - q[0]); s[1] += (double)(q[1] - q[0]); s[2] += (double)(q[2] - q[1]); s[3] += (double)(q[3] - q[2]);
    const unsigned long long *w = p + 4 * u;
    r[0] += (double)(w[4] - w[0]); r[1] += (double)(w[1] - w[0]); r[2] += (double)(w[2] - w[1]); r[3] += (double)(w[3] - w[2]); r[4] += (double)(w[4] - w[3]);
  }
  printf("consumer, units 4..43 (cycles): unit total %.0f | c0->c1 row reads + 108 MFMA %.0f | c1->c2 folds %.0f | c2->next c0 barrier (+ epilogue) %.0f\n", s[0] / cnt, s[1] / cnt, s[2] / cnt, s[3] / cnt);
  printf("producer, iterations 4..43 (cycles): total %.0f | p0->p1 scale, split, LDS writes %.0f | p1->p2 next unit's loads + maximum %.0f | p2->p3 issue loads %.0f | p3->next p0 barrier %.0f\n",
         r[0] / cnt, r[1] / cnt, r[2] / cnt, r[3] / cnt, r[4] / cnt);
  printf("(the shader clock of s_memtime runs at 100 MHz on this GPU if the numbers look 20x too small)\n");
  return 0;
}
