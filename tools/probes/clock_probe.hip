// Effective shader clock during short kernels: s_memtime (shader cycles) vs the 100 MHz wall clock inside one kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long *out, int iters) {
  float a = threadIdx.x;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)a; }
}
int main() {
  long long *d; hipMalloc(&d, 64);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", rate);
  for (int blocks : {1, 256, 2048}) for (int iters : {2000, 20000, 200000}) {
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipDeviceSynchronize();
      long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      if (rep == 4) printf("blocks %5d iters %7d: %9lld shader ticks, %8lld wall ticks -> %.0f MHz if s_memtime counts shader cycles; %.2f ticks/iter\n",
                           blocks, iters, h[0], h[1], (double)h[0] / ((double)h[1] / (rate * 1e3)) / 1e6, (double)h[0] / iters);
    }
  }
  return 0;
}
