// HBM write bandwidth of the cost-volume store pattern: a (C, D, h, w) fp32 volume written by waves that each
// own (CG channels) x (SEG consecutive floats of one row) of one plane per store round.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave item: (c group of 4, d, y, x-segment of SEG floats): lanes = (channel cw = lane / (SEG/4), 4 px)
template <int SEG>
__global__ __launch_bounds__(256) void vol_store(float *out, int C, int D, int h, int w, int planes_per_wave) {
  constexpr int LPC = SEG / 4;          // lanes per channel
  constexpr int CPI = 64 / LPC;         // channels per store instruction
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  const int segs = w / SEG;
  // wave -> (y, seg, dchunk): all C channels, planes_per_wave planes
  const int nd = D / planes_per_wave;
  int t = wave;
  const int dchunk = t % nd; t /= nd;
  const int seg = t % segs; t /= segs;
  const int y = t;
  if (y >= h) return;
  const int cw = lane / LPC, q = lane % LPC;
  const size_t hw = (size_t)h * w;
  f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
  for (int k = 0; k < planes_per_wave; ++k) {
    const int d = dchunk * planes_per_wave + k;
    for (int c0 = 0; c0 < C; c0 += CPI) {
      float *p = out + ((size_t)(c0 + cw) * D + d) * hw + (size_t)y * w + seg * SEG + 4 * q;
      if (c0 + cw < C) *reinterpret_cast<f32x4 *>(p) = v;
    }
  }
}

template <int SEG>
void run(float *d, int C, int D, int h, int w, int ppw, const char *name) {
  const int waves = h * (w / SEG) * (D / ppw);
  const int blocks = (waves + 3) / 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(vol_store<SEG>, dim3(blocks), dim3(256), 0, 0, d, C, D, h, w, ppw);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(vol_store<SEG>, dim3(blocks), dim3(256), 0, 0, d, C, D, h, w, ppw);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const double bytes = 4.0 * C * D * h * w;
  printf("%-40s C=%d D=%d %dx%d planes/wave %d: %7.1f us  %6.0f GB/s\n", name, C, D, h, w, ppw, ms * 1e3, bytes / ms / 1e6);
}

__global__ void fill(f32x4 *p, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) p[i] = f32x4{1, 2, 3, 4};
}

int main() {
  float *d; hipMalloc(&d, (size_t)768 << 20); setvbuf(stdout, nullptr, _IONBF, 0);
  struct { int C, D, h, w; } lv[] = {{32, 48, 128, 160}, {16, 32, 256, 320}, {8, 8, 512, 640}, {16, 32, 512, 640}};
  for (auto &l : lv) {
    size_t n4 = (size_t)l.C * l.D * l.h * l.w / 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill, dim3((n4 + 255) / 256), dim3(256), 0, 0, (f32x4 *)d, n4);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fill, dim3((n4 + 255) / 256), dim3(256), 0, 0, (f32x4 *)d, n4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-40s C=%d D=%d %dx%d: %7.1f us  %6.0f GB/s\n", "linear fill (16 B/lane, 1 KB/wave)", l.C, l.D, l.h, l.w, ms * 1e3, n4 * 16.0 / ms / 1e6);
    for (int ppw : {1, 8}) {
      run<16>(d, l.C, l.D, l.h, l.w, ppw, "64 B segments (16 ch / store)");
      run<32>(d, l.C, l.D, l.h, l.w, ppw, "128 B segments (8 ch / store)");
      run<64>(d, l.C, l.D, l.h, l.w, ppw, "256 B segments (4 ch / store)");

    }
  }
  return 0;
}
