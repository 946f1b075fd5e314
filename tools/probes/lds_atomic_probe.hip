// LDS read-modify-write probe: what does a wave-instruction of ds_add_f32 cost next to ds_add_u32 / ds_add_u64 / a plain read + add + write, as a function of
// the number of active lanes and of the address pattern?  (The variance-volume backward of train.hip scatters through LDS float atomics.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_atomic_probe.hip -o tools/probes/bin/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum { F32_ATOMIC = 0, U32_ATOMIC = 1, U64_ATOMIC = 2, F32_RMW = 3, F32_ATOMIC_RTN = 4, F32_READ_ONLY = 5, F32_WRITE_ONLY = 6 };

template <int KIND>
__global__ __launch_bounds__(256) void probe(const int *lane_word, float *out, long long *cycles, int iters, unsigned long long active) {
  __shared__ __attribute__((aligned(16))) float lds[16384];   // 64 KiB: two workgroups per CU
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = lane_word[lane] + wave * 4096;   // every wave its own quarter: no cross-wave sharing of words
  const bool on = (active >> lane) & 1;
  float acc = 0.0f;
  const float v = 1.0f + lane;
  long long t0 = clock64();
  if (on) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float *p = lds + ((a + r * 200 + (it & 7) * 8) & 4095) + wave * 4096 - (a & ~4095) * 0;
        if (KIND == F32_ATOMIC) atomicAdd(p, v);
        else if (KIND == U32_ATOMIC) atomicAdd(reinterpret_cast<unsigned *>(p), 3u + lane);
        else if (KIND == U64_ATOMIC) atomicAdd(reinterpret_cast<unsigned long long *>(lds + ((((a & 2047) + r * 100 + (it & 7) * 4) & 2047) * 2 + wave * 4096)), 3ull + lane);
        else if (KIND == F32_RMW) { *p = *p + v; }
        else if (KIND == F32_ATOMIC_RTN) acc += atomicAdd(p, v);
        else if (KIND == F32_READ_ONLY) acc += *p;
        else *p = v + it;
      }
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int KIND>
double run(const int *d_word, float *d_out, long long *d_cyc, int blocks, int iters, unsigned long long active) {
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_word, d_out, d_cyc, iters, active);
    hipDeviceSynchronize();
  }
  std::vector<long long> c(blocks);
  hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : c) mean += v;
  return mean / blocks / (16.0 * iters);
}

int main() {
  struct Pat { const char *name; int (*f)(int); };
  Pat pats[] = {
      {"consecutive words", [](int i) { return i; }},
      {"stride 2", [](int i) { return 2 * i; }},
      {"two rows of 32 (row stride 48)", [](int i) { return (i & 31) + (i >> 5) * 48; }},
      {"pairs share a word", [](int i) { return i >> 1; }},
      {"stride 32 (one bank)", [](int i) { return 32 * i; }},
      {"all lanes one word", [](int i) { return 0 * i; }},
  };
  struct Act { const char *name; unsigned long long mask; };
  Act acts[] = {{"64 lanes", ~0ull}, {"32 lanes (low half)", 0xffffffffull}, {"every 2nd lane", 0x5555555555555555ull}, {"8 lanes", 0xffull}, {"1 lane", 1ull}};
  int *d_word; float *d_out; long long *d_cyc;
  const int iters = 500;
  hipMalloc(&d_word, 64 * 4); hipMalloc(&d_out, 512 * 256 * 4); hipMalloc(&d_cyc, 512 * 8);
  const char *kinds[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "read+add+write", "ds_add_rtn_f32", "ds_read_b32", "ds_write_b32"};
  for (int blocks : {256, 512}) {
    printf("== %d workgroups of 4 waves (%d per CU), ticks (clock64) per wave-instruction, every wave issuing\n", blocks, blocks / 256);
    for (auto &p : pats) {
      int h[64];
      for (int i = 0; i < 64; ++i) h[i] = p.f(i);
      hipMemcpy(d_word, h, sizeof(h), hipMemcpyHostToDevice);
      for (auto &a : acts) {
        if (a.mask != ~0ull && p.f != pats[0].f) continue;
        double t[7];
        t[0] = run<0>(d_word, d_out, d_cyc, blocks, iters, a.mask); t[1] = run<1>(d_word, d_out, d_cyc, blocks, iters, a.mask);
        t[2] = run<2>(d_word, d_out, d_cyc, blocks, iters, a.mask); t[3] = run<3>(d_word, d_out, d_cyc, blocks, iters, a.mask);
        t[4] = run<4>(d_word, d_out, d_cyc, blocks, iters, a.mask); t[5] = run<5>(d_word, d_out, d_cyc, blocks, iters, a.mask);
        t[6] = run<6>(d_word, d_out, d_cyc, blocks, iters, a.mask);
        printf("%-32s %-20s", p.name, a.name);
        for (int k = 0; k < 7; ++k) printf("  %s %7.1f", kinds[k], t[k]);
        printf("\n");
      }
    }
  }
  return 0;
}
