// Issue rate of the float32 vector FMA forms on the MI355X: v_fma_f32, v_pk_fma_f32 with vector operands, v_pk_fma_f32 with a scalar (SGPR pair)
// multiplier - the form prob_zwalk.h uses - at 1, 2 and 4 waves per SIMD.  Twelve independent accumulator chains per wave (no dependency stalls);
// time from HIP events over a grid that fills every SIMD with exactly `waves` waves.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o tools/probes/bin/valu_rate && tools/probes/bin/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NCHAIN = 12>   // 0 v_fma_f32, 1 v_pk_fma_f32 (vector operands), 2 v_pk_fma_f32 (scalar multiplier); NCHAIN independent accumulators
__global__ __launch_bounds__(256) void rate_kernel(float *sink, const float *wsrc, int iters) {
  f32x2 acc[NCHAIN];
#pragma unroll
  for (int i = 0; i < NCHAIN; ++i) acc[i] = f32x2{(float)(threadIdx.x + i), 1.0f};
  const f32x2 mv = {1.0f + threadIdx.x * 1e-7f, 1.0f - threadIdx.x * 1e-7f}, c = {1e-3f, -1e-3f};
  const f32x2 ms = {wsrc[0], wsrc[1]};   // wave-uniform: SGPR pair
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 96 / NCHAIN; ++rep)
#pragma unroll
      for (int i = 0; i < NCHAIN; ++i) {
        if (KIND == 0) { acc[i][0] = fmaf(acc[i][0], mv[0], c[0]); acc[i][1] = fmaf(acc[i][1], mv[1], c[1]); }
        else if (KIND == 1) acc[i] = __builtin_elementwise_fma(acc[i], mv, c);
        else acc[i] = __builtin_elementwise_fma(acc[i], ms, c);
      }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NCHAIN; ++i) s += acc[i][0] + acc[i][1];
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float *sink, *w;
  hipMalloc(&sink, 64); hipMalloc(&w, 64);
  const float hw[2] = {1.0000001f, 0.9999999f};
  hipMemcpy(w, hw, 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[3] = {"v_fma_f32 (2 per pair)", "v_pk_fma_f32, vector operands", "v_pk_fma_f32, scalar multiplier"};
  const int iters = 2000;
  printf("%s, %d CUs, clock %d kHz; 12 independent chains x 8 x %d iterations per wave\n", prop.gcnArchName, cus, prop.clockRate, iters);
  for (int kind = 0; kind < 3; ++kind)
    for (int waves = 1; waves <= 4; waves *= 2) {
      const dim3 grid(cus * waves), blk(256);   // a 256-thread workgroup = one wave per SIMD of its CU
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, grid, blk, 0, 0, sink, w, iters);
        else if (kind == 1) hipLaunchKernelGGL(rate_kernel<1>, grid, blk, 0, 0, sink, w, iters);
        else hipLaunchKernelGGL(rate_kernel<2>, grid, blk, 0, 0, sink, w, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double pair_fmas = 12.0 * 8 * iters * waves;   // per SIMD: wave-instructions' worth of PAIRS
      const double cycles = best * 1e-3 * prop.clockRate * 1e3;
      const double tflops = 2.0 * 2 * 64 * pair_fmas * 4 * cus / (best * 1e-3) / 1e12;
      printf("%-34s %d wave(s) per SIMD: %.3f ms, %.2f cycles per pair-of-FMAs wave-instruction slot, %.1f TFLOP/s\n", names[kind], waves, best, cycles / pair_fmas, tflops);
    }
  // dependent-issue latency: the scalar-multiplier form with 1 .. 12 independent accumulator chains, one wave per SIMD (prob_zwalk.h runs six)
  printf("v_pk_fma_f32 (scalar multiplier), one wave per SIMD, by the number of independent chains:\n");
  auto chains = [&](auto k, int n) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(cus), dim3(256), 0, 0, sink, w, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("  %2d chains: %.2f cycles per instruction\n", n, best * 1e-3 * prop.clockRate * 1e3 / (96.0 * iters));
  };
  chains(rate_kernel<2, 1>, 1); chains(rate_kernel<2, 2>, 2); chains(rate_kernel<2, 3>, 3); chains(rate_kernel<2, 4>, 4); chains(rate_kernel<2, 6>, 6);
  chains(rate_kernel<2, 8>, 8); chains(rate_kernel<2, 12>, 12);
  return 0;
}
