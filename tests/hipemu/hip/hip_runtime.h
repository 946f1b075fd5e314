// A stand-in for <hip/hip_runtime.h> that lets the kernels of casmvsnet_pl_amd/csrc/*.hip be compiled as plain C++ for the HOST (ROCm's clang++,
// x86-64) and executed on the CPU: one std::thread per GPU thread of a workgroup, workgroups one after the other.  Test infrastructure only
// (tests/test_hip_emulation.py): it exists so that a kernel written without access to a GPU can be RUN - its own source, not a transcription -
// against a float64 reference before it ever meets the hardware.
//
// What is emulated (the subset those kernels use):
//   * __global__ / __device__ / __shared__ (one dynamic LDS array `smem_raw` per workgroup), threadIdx / blockIdx / gridDim, __syncthreads,
//     hipLaunchKernelGGL, the handful of host API calls the launch wrappers make;
//   * raw buffer loads / stores with hardware range checking (an offset at or beyond num_records reads zeros / drops the store);
//   * the wave-collective operations - v_mfma_f32_16x16x32_f16 (lane layout as casmvs_selftest_mfma_f16 verified it on the MI355X), the DPP
//     exchanges and v_readlane of casmvs::wave_max_bits - as rendezvous of the wave's 64 threads;
//   * scheduling builtins as no-ops.
// What is NOT emulated: timing, bank conflicts, memory ordering beyond barriers, and the co-residency hazard of DESIGN.md 2.0.
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <barrier>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#ifndef __shared__   // a translation unit whose kernels hold function-scope static LDS arrays defines it as `static` before this header
#define __shared__
#endif
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

namespace hipemu {
struct Dim3 {
  unsigned x = 1, y = 1, z = 1;
  Dim3() = default;
  Dim3(unsigned x_, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct Block {
  std::unique_ptr<std::barrier<>> bar;                 // the workgroup barrier
  std::vector<std::unique_ptr<std::barrier<>>> wave;   // one rendezvous per wave of 64
  // exchange buffers of the wave collectives
  std::vector<std::array<_Float16, 64 * 8>> mfma_a, mfma_b;
  std::vector<std::array<float, 64>> mfma_fa, mfma_fb;
  std::vector<std::array<uint32_t, 64>> lane_u32;
};
inline thread_local Dim3 t_threadIdx, t_blockIdx;
inline Dim3 g_gridDim, g_blockDim;
inline Block *g_block = nullptr;
inline thread_local int t_lane = 0, t_wave = 0;
}  // namespace hipemu

#define threadIdx hipemu::t_threadIdx
#define blockIdx hipemu::t_blockIdx
#define gridDim hipemu::g_gridDim
#define blockDim hipemu::g_blockDim
typedef hipemu::Dim3 dim3;
struct float2 {
  float x, y;
};

// The workgroup's dynamic LDS: the kernels declare `extern __shared__ ... smem_raw[]` inside their anonymous namespace, so the translation unit that
// includes them defines `namespace { alignas(64) unsigned char smem_raw[HIPEMU_LDS_BYTES]; }` and registers it with hipemu::g_lds.
#define HIPEMU_LDS_BYTES (160 * 1024 + 64)
namespace hipemu { inline unsigned char *g_lds = nullptr; inline bool g_reverse_blocks = false; }

// LDS bank profile (tests/hipemu/lds_profile.cpp; built with -DHIPEMU_LDS_PROFILE -fsanitize=thread but linked against that file instead of the sanitizer's
// runtime): the compiler's memory-access hooks record every access that falls into the workgroup's LDS array, these calls tell the recorder who is running
#ifdef HIPEMU_LDS_PROFILE
extern "C" void hipemu_prof_block_begin(const void *lds, size_t bytes, int nthreads);
extern "C" void hipemu_prof_thread(int tid);
extern "C" void hipemu_prof_barrier(void);
extern "C" void hipemu_prof_bufop(int on);
extern "C" void hipemu_prof_block_end(void);
#define HIPEMU_PROF(call) call
#else
#define HIPEMU_PROF(call) ((void)0)
#endif

inline void __syncthreads() {
  hipemu::g_block->bar->arrive_and_wait();
  HIPEMU_PROF(hipemu_prof_barrier());
}
inline float unsafeAtomicAdd(float *p, float v) { return std::atomic_ref<float>(*p).fetch_add(v, std::memory_order_relaxed); }
inline float atomicAdd(float *p, float v) { return std::atomic_ref<float>(*p).fetch_add(v, std::memory_order_relaxed); }
inline int atomicAdd(int *p, int v) { return std::atomic_ref<int>(*p).fetch_add(v, std::memory_order_relaxed); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return std::atomic_ref<unsigned long long>(*p).fetch_add(v, std::memory_order_relaxed); }
inline unsigned atomicMax(unsigned *p, unsigned v) {
  std::atomic_ref<unsigned> a(*p);
  unsigned cur = a.load(std::memory_order_relaxed);
  while (v > cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
  return cur;
}
inline int atomicMin(int *p, int v) {
  std::atomic_ref<int> a(*p);
  int cur = a.load(std::memory_order_relaxed);
  while (v < cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
  return cur;
}
inline int atomicMax(int *p, int v) {
  std::atomic_ref<int> a(*p);
  int cur = a.load(std::memory_order_relaxed);
  while (v > cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
  return cur;
}
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }

// ---- host API subset ----
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
template <class T>
inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)std::malloc(n); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int c, size_t n, void *) { std::memset(d, c, n); return 0; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }

template <class K, class... Args>
void hipemu_launch(K kernel, dim3 grid, dim3 block, Args... args) {
  using namespace hipemu;
  g_gridDim = grid;
  g_blockDim = block;
  const int nthreads = (int)(block.x * block.y * block.z), nwaves = (nthreads + 63) / 64;
  // g_reverse_blocks: the workgroups run in the opposite order - a result that depends on the order in which workgroups add (float atomics) changes
  // its last bits, an order-independent one (integer atomics, fixed-order reductions) does not
  for (unsigned bzi = 0; bzi < grid.z; ++bzi)
    for (unsigned byi = 0; byi < grid.y; ++byi)
      for (unsigned bxi = 0; bxi < grid.x; ++bxi) {
        const unsigned bz = g_reverse_blocks ? grid.z - 1 - bzi : bzi, by = g_reverse_blocks ? grid.y - 1 - byi : byi, bx = g_reverse_blocks ? grid.x - 1 - bxi : bxi;
        Block blk;
        blk.bar = std::make_unique<std::barrier<>>(nthreads);
        blk.mfma_a.resize(nwaves);
        blk.mfma_b.resize(nwaves);
        blk.lane_u32.resize(nwaves);
        blk.mfma_fa.resize(nwaves);
        blk.mfma_fb.resize(nwaves);
        for (int w = 0; w < nwaves; ++w) blk.wave.push_back(std::make_unique<std::barrier<>>(std::min(64, nthreads - 64 * w)));
        g_block = &blk;
        if (g_lds) std::memset(g_lds, 0xCD, HIPEMU_LDS_BYTES);   // uninitialised LDS is garbage on the GPU too
        HIPEMU_PROF(hipemu_prof_block_begin(g_lds, HIPEMU_LDS_BYTES, nthreads));
        std::vector<std::thread> threads;
        std::atomic<int> early{0};
        for (int t = 0; t < nthreads; ++t)
          threads.emplace_back([&, t] {
            t_threadIdx = Dim3((unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y));
            t_blockIdx = Dim3(bx, by, bz);
            t_lane = t & 63;
            t_wave = t >> 6;
            HIPEMU_PROF(hipemu_prof_thread(t));
            kernel(args...);
          });
        for (auto &th : threads) th.join();
        HIPEMU_PROF(hipemu_prof_block_end());
      }
  g_block = nullptr;
}
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hipemu_launch(kernel, grid, block, __VA_ARGS__)

// ---- raw buffer addressing ----
struct hipemu_rsrc {
  char *base;
  uint32_t bytes;
};
#define __amdgpu_buffer_rsrc_t hipemu_rsrc
inline hipemu_rsrc hipemu_make_rsrc(void *p, int /*stride*/, int num_records, int /*flags*/) { return hipemu_rsrc{(char *)p, (uint32_t)num_records}; }
#define __builtin_amdgcn_make_buffer_rsrc hipemu_make_rsrc
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
template <class T>
inline T hipemu_buf_load(hipemu_rsrc r, int voff, int soff) {
  T v{};
  const uint64_t off = (uint64_t)(uint32_t)voff + (uint64_t)(uint32_t)soff;
  HIPEMU_PROF(hipemu_prof_bufop(1));
  if ((uint32_t)voff < r.bytes && off + sizeof(T) <= r.bytes) std::memcpy(&v, r.base + off, sizeof(T));   // out of range: zeros
  HIPEMU_PROF(hipemu_prof_bufop(0));
  return v;
}
template <class T>
inline void hipemu_buf_store(T v, hipemu_rsrc r, int voff, int soff) {
  const uint64_t off = (uint64_t)(uint32_t)voff + (uint64_t)(uint32_t)soff;
  HIPEMU_PROF(hipemu_prof_bufop(1));
  if ((uint32_t)voff < r.bytes && off + sizeof(T) <= r.bytes) std::memcpy(r.base + off, &v, sizeof(T));   // out of range: dropped
  HIPEMU_PROF(hipemu_prof_bufop(0));
}
#define __builtin_amdgcn_raw_buffer_load_b32(r, v, s, aux) hipemu_buf_load<unsigned>(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b64(r, v, s, aux) hipemu_buf_load<hipemu_u32x2>(r, v, s)
#define __builtin_amdgcn_raw_buffer_load_b128(r, v, s, aux) hipemu_buf_load<hipemu_u32x4>(r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b32(val, r, v, s, aux) hipemu_buf_store<unsigned>(val, r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b64(val, r, v, s, aux) hipemu_buf_store<hipemu_u32x2>(val, r, v, s)
#define __builtin_amdgcn_raw_buffer_store_b128(val, r, v, s, aux) hipemu_buf_store<hipemu_u32x4>(val, r, v, s)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)

// ---- wave collectives (every lane of the wave must arrive: the kernels call them under wave-uniform control flow) ----
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
inline hipemu_f32x4 hipemu_mfma_16x16x32_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x4 c, int, int, int) {
  using namespace hipemu;
  Block &blk = *g_block;
  auto &A = blk.mfma_a[t_wave], &Bm = blk.mfma_b[t_wave];
  for (int e = 0; e < 8; ++e) {
    A[t_lane * 8 + e] = a[e];
    Bm[t_lane * 8 + e] = b[e];
  }
  blk.wave[t_wave]->arrive_and_wait();
  // lane l: column j = l & 15, rows i = 4 (l >> 4) + r;  A row i, k = 8 kb + e lives in lane i + 16 kb; B column j, the same k, in lane j + 16 kb
  const int j = t_lane & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (t_lane >> 4) + r;
    double sum = 0.0;
    for (int kb = 0; kb < 4; ++kb)
      for (int e = 0; e < 8; ++e) sum += (double)(float)A[(i + 16 * kb) * 8 + e] * (double)(float)Bm[(j + 16 * kb) * 8 + e];
    d[r] = (float)((double)c[r] + sum);
  }
  blk.wave[t_wave]->arrive_and_wait();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 hipemu_mfma_16x16x32_f16

// v_mfma_f32_16x16x4_f32: A lane l holds A[i = l & 15][k = l >> 4], B lane l holds B[k = l >> 4][j = l & 15]; D as above (column j = l & 15, rows 4 (l >> 4) + r)
inline hipemu_f32x4 hipemu_mfma_16x16x4_f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  using namespace hipemu;
  Block &blk = *g_block;
  auto &A = blk.mfma_fa[t_wave], &Bm = blk.mfma_fb[t_wave];
  A[t_lane] = a;
  Bm[t_lane] = b;
  blk.wave[t_wave]->arrive_and_wait();
  const int j = t_lane & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (t_lane >> 4) + r;
    double sum = 0.0;
    for (int k = 0; k < 4; ++k) sum += (double)A[i + 16 * k] * (double)Bm[j + 16 * k];
    d[r] = (float)((double)c[r] + sum);
  }
  blk.wave[t_wave]->arrive_and_wait();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4_f32

inline int hipemu_lane_exchange(int v, int src_lane) {
  using namespace hipemu;
  Block &blk = *g_block;
  blk.lane_u32[t_wave][t_lane] = (uint32_t)v;
  blk.wave[t_wave]->arrive_and_wait();
  const int out = (int)blk.lane_u32[t_wave][src_lane];
  blk.wave[t_wave]->arrive_and_wait();
  return out;
}
// the DPP controls casmvs::wave_max_bits uses (bound_ctrl / masks irrelevant for them: every lane has a source inside its row)
inline int hipemu_update_dpp(int old, int src, int ctrl, int, int, bool) {
  const int l = hipemu::t_lane, row = l & ~15, q = l & ~3;
  int from;
  if (ctrl == 0x138 || ctrl == 0x130) {   // wave_shr:1 / wave_shl:1 (bound_ctrl off): the lane without a source keeps `old`
    const int f = ctrl == 0x138 ? l - 1 : l + 1;
    const int got = hipemu_lane_exchange(src, f < 0 || f > 63 ? l : f);
    return f < 0 || f > 63 ? old : got;
  }
  if (ctrl == 0xB1) from = q + ((l & 3) ^ 1);            // quad_perm [1,0,3,2]
  else if (ctrl == 0x4E) from = q + ((l & 3) ^ 2);       // quad_perm [2,3,0,1]
  else if (ctrl == 0x141) from = (l & ~7) + (7 - (l & 7));   // row_half_mirror
  else if (ctrl == 0x140) from = row + (15 - (l & 15));  // row_mirror
  else { std::fprintf(stderr, "hipemu: DPP control 0x%x is not emulated\n", ctrl); std::abort(); }
  return hipemu_lane_exchange(src, from);
}
#define __builtin_amdgcn_update_dpp hipemu_update_dpp
inline unsigned hipemu_readlane(unsigned v, int lane) { return (unsigned)hipemu_lane_exchange((int)v, lane); }
#define __builtin_amdgcn_readlane hipemu_readlane
// v_readfirstlane under wave-uniform control flow: lane 0's value (a value that is NOT uniform shows, as on the hardware)
inline unsigned hipemu_readfirstlane(unsigned v) { return (unsigned)hipemu_lane_exchange((int)v, 0); }
#define __builtin_amdgcn_readfirstlane hipemu_readfirstlane
inline uint64_t hipemu_ballot(bool p) {
  using namespace hipemu;
  Block &blk = *g_block;
  blk.lane_u32[t_wave][t_lane] = p ? 1u : 0u;
  blk.wave[t_wave]->arrive_and_wait();
  uint64_t m = 0;
  for (int l = 0; l < 64; ++l) m |= (uint64_t)(blk.lane_u32[t_wave][l] & 1u) << l;
  blk.wave[t_wave]->arrive_and_wait();
  return m;
}
#define __builtin_amdgcn_ballot_w64 hipemu_ballot
inline void hipemu_wave_barrier() { hipemu::g_block->wave[hipemu::t_wave]->arrive_and_wait(); }   // the wave's 64 threads are not in lock step here: a real rendezvous
#define __builtin_amdgcn_wave_barrier hipemu_wave_barrier
#define __builtin_amdgcn_fence(order, scope) std::atomic_thread_fence(std::memory_order_seq_cst)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))   // v_rcp_f32 is within 1 ulp of this
