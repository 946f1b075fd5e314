// LDS bank-conflict probe for ds_read_b128 gathers: lane i reads the 16-byte unit A(i) (+ r) - which per-lane
// unit strides are conflict-free?   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_probe.hip -o gpurun_out/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WIDTH>  // 16: ds_read_b128, 8: ds_read_b64
__global__ __launch_bounds__(256) void probe(const int *lane_unit, float *out, long long *cycles, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];  // 64 KiB
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int a = lane_unit[lane] * 4;  // float index
  f32x4 acc = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (a + r * 256 * 0 + (it & 3) * 4) & 16380;
      if (WIDTH == 16) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(lds + o + r * 1024);
        acc += v;
      } else {
        float2 v = *reinterpret_cast<const float2 *>(lds + o + r * 1024);
        acc[0] += v.x; acc[1] += v.y;
      }
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  struct Pat { const char *name; int (*f)(int); };
  Pat pats[] = {
      {"stride 1 unit (contiguous)", [](int i) { return i; }},
      {"stride 2 (CS=8 unpadded)", [](int i) { return 2 * i; }},
      {"stride 3 (CS=8 padded)", [](int i) { return 3 * i; }},
      {"stride 4 (CS=16 unpadded)", [](int i) { return 4 * i; }},
      {"stride 5 (CS=16 padded)", [](int i) { return 5 * i; }},
      {"stride 8 (CS=32 unpadded)", [](int i) { return 8 * i; }},
      {"stride 9 (CS=32 padded)", [](int i) { return 9 * i; }},
      {"stride 4 xor (i>>2)&3", [](int i) { return 4 * i + ((i >> 2) & 3); }},
      {"stride 4 xor (i>>1)&3", [](int i) { return 4 * i + ((i >> 1) & 3); }},
      {"stride 4 + (i>>3)&3", [](int i) { return 4 * i + ((i >> 3) & 3); }},
      {"stride 2 + (i>>3)&1", [](int i) { return 2 * i + ((i >> 3) & 1); }},
      {"stride 2 + (i>>2)&1", [](int i) { return 2 * i + ((i >> 2) & 1); }},
      {"stride 2 + (i>>4)&1", [](int i) { return 2 * i + ((i >> 4) & 1); }},
      {"stride 17", [](int i) { return 17 * i; }},
      {"stride 7", [](int i) { return 7 * i; }},
      {"all same (broadcast)", [](int i) { return 0 * i; }},
  };
  int *d_unit; float *d_out; long long *d_cyc;
  const int blocks = 256, iters = 2000;
  hipMalloc(&d_unit, 64 * 4); hipMalloc(&d_out, blocks * 256 * 4); hipMalloc(&d_cyc, blocks * 8);
  for (int width : {16, 8}) {
    for (auto &p : pats) {
      int h[64];
      for (int i = 0; i < 64; ++i) h[i] = p.f(i);
      hipMemcpy(d_unit, h, sizeof(h), hipMemcpyHostToDevice);
      for (int rep = 0; rep < 2; ++rep) {
        if (width == 16) hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, d_unit, d_out, d_cyc, iters);
        else hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, d_unit, d_out, d_cyc, iters);
        hipDeviceSynchronize();
      }
      std::vector<long long> c(blocks);
      hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
      double mean = 0;
      for (auto v : c) mean += v;
      mean /= blocks;
      // 4 waves per CU (1 block/CU) x 16 reads x iters; clock64 = s_memtime ticks
      printf("b%-3d %-30s %8.2f ticks per wave-read (4 waves/CU issuing)\n", width * 8, p.name, mean / (16.0 * iters));
    }
  }
  return 0;
}
