// VALU issue-rate probe (wave64 on gfx950): cycles per instruction for the op classes of the plane-sweep kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_probe.hip -o tools/probes/bin/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP>
__global__ __launch_bounds__(256) void probe(float *out, long long *cycles, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 1.0001f, c = 0.5f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) {  // v_fma_f32, 8 independent chains
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (OP == 1) {  // v_pk_fma_f32 on 4 register pairs (counts as 4 instructions = 8 FMAs per lane)
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(const double *)&m), "v"(*(const double *)&c));)
    } else if (OP == 2) {  // v_rcp_f32
      REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (OP == 3) {  // v_mul_lo_u32
      REP16(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (OP == 4) {  // v_cndmask_b32 (vcc)
      REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
    } else if (OP == 5) {  // v_add_f32
      REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (OP == 6) {  // v_floor_f32
      REP16(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (OP == 7) {  // v_mad_u32_u24
      REP16(asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n"
                         "v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    }
  }
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char *name, int waves_per_simd) {
  float *d_out; long long *d_cyc;
  const int blocks = 256 * waves_per_simd, iters = 500;
  hipMalloc(&d_out, blocks * 256 * 4); hipMalloc(&d_cyc, blocks * 4 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 1.0f);
    hipDeviceSynchronize();
  }
  std::vector<long long> c(blocks * 4);
  hipMemcpy(c.data(), d_cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : c) mean += v;
  mean /= c.size();
  printf("%-14s %d wave(s)/SIMD: %6.2f ticks per instruction per wave  => %5.2f ticks per instruction on the SIMD\n", name, waves_per_simd,
         mean / (128.0 * iters), mean / (128.0 * iters) / waves_per_simd);
  hipFree(d_out); hipFree(d_cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<5>("v_add_f32", w); run<4>("v_cndmask_b32", w);
    run<2>("v_rcp_f32", w); run<3>("v_mul_lo_u32", w); run<6>("v_floor_f32", w); run<7>("v_mad_u32_u24", w);
  }
  return 0;
}
