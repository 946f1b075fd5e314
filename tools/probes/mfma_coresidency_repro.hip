// Stand-alone reproducer of DESIGN.md 2.0's co-residency observation - no library, no torch, builtins only (no inline assembly: every
// hazard the ISA asks software to cover is the compiler's):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_coresidency_repro.hip -o tools/probes/bin/mfma_coresidency_repro
//   tools/probes/bin/mfma_coresidency_repro [rounds = 300]
// Stream A runs a NEIGHBOUR kernel (a loop of one instruction kind, 2 workgroups per CU, ~1 ms); stream B runs a deterministic VICTIM kernel
// whose waves share SIMDs with it.  Every victim launch is compared bit for bit with the victim's output when it runs ALONE.
//   victims:    f32mfma  LDS-fed chains of v_mfma_f32_16x16x4_f32 + a float32 epilogue (the shape of the engine's float32 layer kernels)
//               pkfma    chains of v_pk_fma_f32 (the plane loop of the LDS cost-volume kernel)
//               fma      chains of v_fma_f32 (control)
//   neighbours: none, f32mfma (v_mfma_f32_16x16x4_f32), f16mfma (v_mfma_f32_16x16x32_f16), bf16mfma (v_mfma_f32_16x16x32_bf16), valu, lds
// Round 3 saw, inside the engine: float32-MFMA and packed-float32 victims corrupted (lanes 48-63) beside f16 / bf16 MFMA neighbours only.
// Exit status: 0 = no victim launch differed (nothing reproduced here), 1 = at least one differed (the table says which pair).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int KIND>   // 1 f32 MFMA, 2 f16 MFMA, 3 bf16 MFMA, 4 VALU, 5 LDS
__global__ __launch_bounds__(256) void neighbour(float *sink, int iters) {
  __shared__ u32x4 lds[1024];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  const u32x4 ua = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  if (KIND == 5) { for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u}; __syncthreads(); }
  for (int it = 0; it < iters; ++it) {
    if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    else if (KIND == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, ua), acc, 0, 0, 0);
    else if (KIND == 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ua), acc, 0, 0, 0);
    else if (KIND == 4) { acc[0] = fmaf(acc[0], a, b); acc[1] = fmaf(acc[1], a, b); }
    else { const u32x4 v = lds[(threadIdx.x + it * 263) & 1023]; lds[(threadIdx.x + it * 97 + 5) & 1023] = v; acc[0] += (float)v[0]; }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// victim 0: 4 waves, each 4 accumulator tiles over `steps` K-steps; B operands from an LDS tile the workgroup fills first, A from registers
__global__ __launch_bounds__(256) void victim_f32mfma(float *out, int steps) {
  __shared__ float tile[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) tile[i] = (float)((i * 37 + blockIdx.x * 11) & 255) * (1.0f / 64.0f) - 2.0f;
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int s = 0; s < steps; ++s) {
    const float a = (float)(((lane >> 4) * 5 + s * 3 + (lane & 15)) & 31) * 0.0625f - 1.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = tile[(s * 64 + lane + t * 1031 + (tid >> 6) * 257) & 4095];
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
  float *o = out + ((size_t)blockIdx.x * 256 + tid) * 16;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float v = acc[t][r] * 0.75f + 0.125f; o[t * 4 + r] = v > 0 ? v : v * 0.01f; }
}

template <bool PACKED>
__global__ __launch_bounds__(256) void victim_fma(float *out, int steps) {
  const int tid = threadIdx.x;
  f32x2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)(tid + i) * 0.001f, (float)(blockIdx.x + i) * 0.002f};
  const f32x2 m = {0.999f, 1.001f}, c = {0.01f, -0.01f};
  for (int s = 0; s < steps; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PACKED) acc[i] = __builtin_elementwise_fma(acc[i], m, c);
      else { acc[i][0] = fmaf(acc[i][0], m[0], c[0]); acc[i][1] = fmaf(acc[i][1], m[1], c[1]); }
    }
  float *o = out + ((size_t)blockIdx.x * 256 + tid) * 16;
#pragma unroll
  for (int i = 0; i < 8; ++i) { o[2 * i] = acc[i][0]; o[2 * i + 1] = acc[i][1]; }
}

__global__ void compare(const unsigned *got, const unsigned *ref, size_t n, unsigned *counts) {   // counts[0]: differing values, [1]: of them in lanes 48-63
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (got[i] != ref[i]) { atomicAdd(&counts[0], 1u); if (((i / 16) & 63) >= 48) atomicAdd(&counts[1], 1u); }
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 300;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipStream_t sa, sb;
  CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
  const int vblocks = 4 * cus;
  const size_t n = (size_t)vblocks * 256 * 16;
  float *dout, *dref, *dsink;
  unsigned *dcounts;
  CHECK(hipMalloc(&dout, n * 4)); CHECK(hipMalloc(&dref, n * 4)); CHECK(hipMalloc(&dsink, 64)); CHECK(hipMalloc(&dcounts, 8));
  const char *vnames[3] = {"f32mfma", "pkfma", "fma"}, *nnames[6] = {"none", "f32mfma", "f16mfma", "bf16mfma", "valu", "lds"};
  auto launch_victim = [&](int v) {
    if (v == 0) hipLaunchKernelGGL(victim_f32mfma, dim3(vblocks), dim3(256), 0, sb, dout, 600);
    else if (v == 1) hipLaunchKernelGGL(victim_fma<true>, dim3(vblocks), dim3(256), 0, sb, dout, 1500);
    else hipLaunchKernelGGL(victim_fma<false>, dim3(vblocks), dim3(256), 0, sb, dout, 1500);
  };
  auto launch_neighbour = [&](int k) {
    const dim3 g(2 * cus), b(256);
    const int it = 40000;
    if (k == 1) hipLaunchKernelGGL(neighbour<1>, g, b, 0, sa, dsink, it / 2);
    else if (k == 2) hipLaunchKernelGGL(neighbour<2>, g, b, 0, sa, dsink, it);
    else if (k == 3) hipLaunchKernelGGL(neighbour<3>, g, b, 0, sa, dsink, it);
    else if (k == 4) hipLaunchKernelGGL(neighbour<4>, g, b, 0, sa, dsink, it * 2);
    else if (k == 5) hipLaunchKernelGGL(neighbour<5>, g, b, 0, sa, dsink, it / 2);
  };
  printf("%s, %d CUs; %d rounds per (victim, neighbour) pair; a round = one neighbour launch on stream A (~0.5-1 ms) + one victim launch on stream B (~0.15 ms), compared with the victim alone\n",
         prop.gcnArchName, cus, rounds);
  int any = 0;
  for (int v = 0; v < 3; ++v) {
    launch_victim(v);
    CHECK(hipStreamSynchronize(sb));
    CHECK(hipMemcpy(dref, dout, n * 4, hipMemcpyDeviceToDevice));
    for (int k = 0; k < 6; ++k) {
      int bad_launches = 0;
      unsigned long long wrong = 0, wrong_hi = 0;
      for (int r = 0; r < rounds; ++r) {
        CHECK(hipMemsetAsync(dcounts, 0, 8, sb));
        launch_neighbour(k);
        launch_victim(v);
        hipLaunchKernelGGL(compare, dim3(1024), dim3(256), 0, sb, (const unsigned *)dout, (const unsigned *)dref, n, dcounts);
        unsigned c[2];
        CHECK(hipMemcpyAsync(c, dcounts, 8, hipMemcpyDeviceToHost, sb));
        CHECK(hipStreamSynchronize(sb));
        CHECK(hipStreamSynchronize(sa));
        bad_launches += c[0] > 0;
        wrong += c[0];
        wrong_hi += c[1];
      }
      printf("victim %-8s beside %-8s: %4d of %4d launches differ from the solo run", vnames[v], nnames[k], bad_launches, rounds);
      if (bad_launches) printf("  (%llu wrong values, %llu of them in lanes 48-63)", wrong, wrong_hi);
      printf("\n");
      any |= bad_launches > 0;
    }
  }
  printf(any ? "REPRODUCED: a victim's results depend on what runs beside it\n" : "not reproduced: every victim launch equals its solo run\n");
  return any ? 1 : 0;
}
