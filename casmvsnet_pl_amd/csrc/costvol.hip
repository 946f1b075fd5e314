// Plane-sweep kernels: homo_warp, fused warp+variance, fused warp+group-wise correlation.
//
// Reference semantics (cited per function in include/casmvs.h):
//   models/modules.py:52-92  homo_warp        models/mvsnet.py:134-172  aggregation
//
// Design (gfx950): one thread per (ref pixel, depth plane); a 64-lane wavefront covers 64
// consecutive pixels of one image row, so the depth read, every bilinear tap (a near-affine
// image of that row segment in the source view) and every volume write is a coalesced 256 B
// access.  All views and all C channels of a voxel are accumulated in registers (sum / sum of
// squares / correlation), so the (C, D, h, w) volume is written exactly once and no warped
// volume is ever materialised.  Blocks are mapped so that each XCD owns a contiguous band of
// image rows: the per-XCD L2 (4 MiB) then only has to hold 1/8 of every source feature map.
#include "common.h"
#include "plane_sweep.h"

namespace {

using namespace casmvs_dev;

constexpr int kThreads = 256;

// The 2x2 bilinear footprint is fetched as two 8-byte row pairs (x, x+1): half the vector-memory
// instructions of four scalar taps.
__device__ __forceinline__ float sample(const float *__restrict__ plane, const Taps &t, int W) {
  const f32x2u n = *reinterpret_cast<const f32x2u *>(plane + t.yn * W + t.xl);
  const f32x2u s = *reinterpret_cast<const f32x2u *>(plane + t.ys * W + t.xl);
  return fmaf(s[1], t.w_sr, fmaf(s[0], t.w_sl, fmaf(n[1], t.w_nr, n[0] * t.w_nl)));
}

// XCD-aware block -> (pixel tile, depth plane) map.  Block b runs on XCD b % 8 (observed
// dispatch order; speed only, never correctness): XCD k owns pixel tiles
// [k * tiles_per_xcd, (k + 1) * tiles_per_xcd), depth plane fastest inside a tile.
__device__ __forceinline__ bool block_to_tile(int D, int tiles, int tiles_per_xcd, int &tile,
                                              int &d) {
  int blk = blockIdx.x;
  int xcd = blk & 7;
  int i = blk >> 3;
  int tile_local = i / D;
  d = i - tile_local * D;
  tile = xcd * tiles_per_xcd + tile_local;
  return tile < tiles;
}

// ---- homo_warp (un-fused op, modules.py:52-92) ---------------------------------------------
__global__ __launch_bounds__(kThreads) void homo_warp_kernel(
    const float *__restrict__ src, const float *__restrict__ proj, const float *__restrict__ depth,
    float *__restrict__ out, int C, int H, int W, int D, int tiles, int tiles_per_xcd) {
  int tile, d;
  if (!block_to_tile(D, tiles, tiles_per_xcd, tile, d)) return;
  const int b = blockIdx.y;
  const int hw = H * W;
  const int p = tile * kThreads + threadIdx.x;
  if (p >= hw) return;
  const int y = p / W, x = p - y * W;
  const float dv = depth[((size_t)b * D + d) * hw + p];
  const Taps t = plane_sweep_taps(proj + (size_t)b * 12, (float)x, (float)y, dv, W, H);
  const float *sp = src + (size_t)b * C * hw;
  float *op = out + ((size_t)b * C * D + d) * hw + p;
  for (int c = 0; c < C; ++c) op[(size_t)c * D * hw] = sample(sp + (size_t)c * hw, t, W);
}

// ---- fused warp + aggregation ---------------------------------------------------------------
// MODE 0: variance (mvsnet.py:139-141,150-156,167), output (B, C, D, h, w)
// MODE 1: group-wise correlation (mvsnet.py:143-144,158-162,170-171), output (B, G, D, h, w)
// CH channels are held in registers per thread; blockIdx.z selects the channel chunk (MODE 0).
template <int CH, int MODE>
__global__ __launch_bounds__(kThreads) void costvol_kernel(
    const float *__restrict__ feats, const float *__restrict__ proj,
    const float *__restrict__ depth, float *__restrict__ out, int V, int C, int G, int h, int w,
    int D, int tiles, int tiles_per_xcd) {
  int tile, d;
  if (!block_to_tile(D, tiles, tiles_per_xcd, tile, d)) return;
  const int b = blockIdx.y;
  const int c0 = blockIdx.z * CH;
  const int hw = h * w;
  const int p = tile * kThreads + threadIdx.x;
  if (p >= hw) return;
  const int y = p / w, x = p - y * w;
  const float dv = depth[((size_t)b * D + d) * hw + p];
  const float *fb = feats + ((size_t)b * V * C + c0) * hw;  // view 0 = reference view
  float ref[CH], s[CH], q[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    ref[c] = fb[(size_t)c * hw + p];
    if (MODE == 0) {
      s[c] = ref[c];                    // volume_sum = ref_volume            (mvsnet.py:140)
      q[c] = ref[c] * ref[c];           // volume_sq_sum = ref_volume ** 2    (mvsnet.py:141)
    } else {
      s[c] = 0.0f;                      // volume_sum = 0                     (mvsnet.py:144)
    }
  }
  for (int v = 1; v < V; ++v) {
    const Taps t =
        plane_sweep_taps(proj + ((size_t)b * (V - 1) + (v - 1)) * 12, (float)x, (float)y, dv, w, h);
    const float *sp = fb + (size_t)v * C * hw;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float val = sample(sp + (size_t)c * hw, t, w);
      s[c] = s[c] + val;
      if (MODE == 0) q[c] = fmaf(val, val, q[c]);
    }
  }
  if (MODE == 0) {
    // sq/V - (sum/V)^2 (mvsnet.py:167); x/V is evaluated as x * (1/V): <= 1 ulp from the
    // reference's division and ~10 VALU instructions cheaper per channel (64 divisions/thread)
    const float rV = 1.0f / (float)V;
    float *op = out + (((size_t)b * C + c0) * D + d) * hw + p;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float m = s[c] * rV;
      op[(size_t)c * D * hw] = q[c] * rV - m * m;
    }
  } else {
    const int cpg = C / G;  // CH == C here
    float *op = out + ((size_t)b * G * D + d) * hw + p;
    const float fn = (float)cpg, fv = (float)(V - 1);
    float acc = 0.0f;
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {  // static register indexing; group boundaries are runtime
      acc = acc + s[c] * ref[c];  // volume_sum * ref_volume (mvsnet.py:170)
      if (++cnt == cpg) {
        *op = (acc / fn) / fv;  // mean over C/G, then / (V-1)
        op += (size_t)D * hw;
        acc = 0.0f;
        cnt = 0;
      }
    }
  }
}

// ---- channel-last variant ---------------------------------------------------------------------
// Measured on MI355X (tools/gpu_costvol_probe.py, profiles/): the NCHW kernel above is bound by the
// rate at which the texture addresser retires gather instructions (~20 cycles per wave-instruction
// whatever its width, with perfect tap locality as well as with noisy depth), not by HBM.  With the
// feature maps stored pixel-major (B, V, h, w, C) one bilinear tap of 4 channels is ONE 16-byte load,
// so a voxel needs C loads per source view instead of 2 C.  Same taps, same weights, same order of
// operations per channel as costvol_kernel: results are bit-identical.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Lane roles.  A WAVEFRONT owns 64 consecutive pixels of one depth plane and all C channels; the four
// wavefronts of a workgroup never synchronise with each other:
//   taps:    lane = pixel computes the source view's taps once;
//   gather:  lane = (pixel, group of 4 channels), the C/4 lanes of a pixel adjacent, so that one
//            wave-wide 16-byte gather reads 64/(C/4) pixels x C contiguous floats = whole cache
//            lines (the addresser's cost grows with the number of lines an instruction touches:
//            measured ~1 cycle/line on top of the ~16-cycle floor).  The taps reach the gather
//            lanes by ds_bpermute (no LDS storage, no barrier); sums stay in registers over views;
//   store:   the variance (or the per-channel products of the group-wise correlation) goes through
//            a wave-private LDS transpose so that the volume is written with lane = (channel,
//            4 pixels): 16-byte stores, 256 B contiguous per channel.
__device__ __forceinline__ int lane_bcast_i(int src_lane, int v) {
  return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}
__device__ __forceinline__ float lane_bcast_f(int src_lane, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

template <int C, int MODE, int OCC = 1>  // OCC: minimum waves per SIMD the register allocation must allow
__global__ __launch_bounds__(kThreads, OCC) void costvol_nhwc_kernel(
    const float *__restrict__ feats, const float *__restrict__ proj,
    const float *__restrict__ depth, float *__restrict__ out, int V, int G, int h, int w, int D,
    int tiles, int tiles_per_xcd) {
  constexpr int GL = C / 4;     // lanes per pixel
  constexpr int PPI = 64 / GL;  // pixels per gather iteration
  constexpr int NI = GL;        // iterations covering the wave's 64 pixels
  constexpr int RS = 65;        // row stride of the transpose buffer (odd: conflict-free)
  __shared__ float tr_all[4 * C * RS];
  int tile, d;
  if (!block_to_tile(D, tiles, tiles_per_xcd, tile, d)) return;
  const int b = blockIdx.y;
  const int hw = h * w;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *tr = tr_all + wave * (C * RS);
  const int pw0 = tile * kThreads + wave * 64;
  if (pw0 >= hw) return;  // wave-uniform
  const int pa = pw0 + lane;  // taps role
  const int ya = pa / w, xa = pa - ya * w;
  const float dv = depth[((size_t)b * D + d) * hw + (pa < hw ? pa : hw - 1)];
  const int g = lane % GL, pxi = lane / GL;
  const size_t view_floats = (size_t)hw * C;
  const float *fb = uniform_ptr(feats + (size_t)b * V * view_floats);  // view 0 = reference view
  float ref[NI][4], s[NI][4], q[NI][4];
  {
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(fb), 0, __builtin_amdgcn_readfirstlane((int)(view_floats * 4)), 0x00020000);
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const f32x4 r = buf_load4(r0, ((pw0 + it * PPI + pxi) * C + 4 * g) * 4, 0);  // beyond the map: 0
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ref[it][i] = r[i];
        s[it][i] = MODE == 0 ? r[i] : 0.0f;   // mvsnet.py:140 / :144
        q[it][i] = r[i] * r[i];                // mvsnet.py:141
      }
    }
  }
  for (int v = 1; v < V; ++v) {
    Taps t = plane_sweep_taps(proj + ((size_t)b * (V - 1) + (v - 1)) * 12, (float)xa, (float)ya, dv, w, h);
    if (pa >= hw) t = Taps{0, 0, 0, 0.0f, 0.0f, 0.0f, 0.0f};
    const int t_on = t.yn * w + t.xl, t_os = t.ys * w + t.xl;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(uniform_ptr(fb + (size_t)v * view_floats)), 0,
        __builtin_amdgcn_readfirstlane((int)(view_floats * 4)), 0x00020000);
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int pl = it * PPI + pxi;  // the wave-local pixel this lane gathers for
      const int on0 = (lane_bcast_i(pl, t_on) * C + 4 * g) * 4, os0 = (lane_bcast_i(pl, t_os) * C + 4 * g) * 4;
      const float w_nl = lane_bcast_f(pl, t.w_nl), w_nr = lane_bcast_f(pl, t.w_nr);
      const float w_sl = lane_bcast_f(pl, t.w_sl), w_sr = lane_bcast_f(pl, t.w_sr);
      const f32x4 n0 = buf_load4(src, on0, 0), n1 = buf_load4(src, on0, C * 4);
      const f32x4 s0 = buf_load4(src, os0, 0), s1 = buf_load4(src, os0, C * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float val = fmaf(s1[i], w_sr, fmaf(s0[i], w_sl, fmaf(n1[i], w_nr, n0[i] * w_nl)));
        s[it][i] = s[it][i] + val;
        if (MODE == 0) q[it][i] = fmaf(val, val, q[it][i]);
      }
    }
  }
  // transpose through the wave's LDS rows (same-wave LDS operations execute in order)
  const float rV = 1.0f / (float)V;
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int pl = it * PPI + pxi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o;
      if (MODE == 0) {
        const float m = s[it][i] * rV;  // sq/V - (sum/V)^2 (mvsnet.py:167), x/V as x * (1/V)
        o = q[it][i] * rV - m * m;
      } else {
        o = s[it][i] * ref[it][i];      // volume_sum * ref_volume (mvsnet.py:170)
      }
      tr[(4 * g + i) * RS + pl] = o;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (MODE == 0) {
    const int q4 = lane & 15, cw = lane >> 4;  // 4 pixels x channels cw, cw + 4, ...
    const int p = pw0 + 4 * q4;
    float *ob = out + ((size_t)b * C * D + d) * hw + p;
#pragma unroll
    for (int j = 0; j < C / 4; ++j) {
      const int c = cw + 4 * j;
      const float *row = tr + c * RS + 4 * q4;
      const f32x4 o{row[0], row[1], row[2], row[3]};
      float *op = ob + (size_t)c * D * hw;
      if ((hw & 3) == 0) {
        if (p < hw) *reinterpret_cast<f32x4 *>(op) = o;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (p + i < hw) op[i] = o[i];
      }
    }
  } else {
    if (pa < hw) {
      const int cpg = C / G;
      const float fn = (float)cpg, fv = (float)(V - 1);
      float *op = out + ((size_t)b * G * D + d) * hw + pa;
      float acc = 0.0f;
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        acc = acc + tr[c * RS + lane];
        if (++cnt == cpg) {
          *op = (acc / fn) / fv;  // mean over C/G, then / (V-1)
          op += (size_t)D * hw;
          acc = 0.0f;
          cnt = 0;
        }
      }
    }
  }
}

// (N, C, h, w) -> (N, h, w, C): thread = pixel, C coalesced dword loads, C/4 16-byte stores
template <int C>
__global__ __launch_bounds__(kThreads) void nchw_to_nhwc_kernel(const float *__restrict__ in,
                                                                float *__restrict__ out, int hw) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  const float *ip = in + (size_t)n * C * hw + p;
  float *op = out + ((size_t)n * hw + p) * C;
#pragma unroll
  for (int g = 0; g < C / 4; ++g) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ip[(size_t)(4 * g + i) * hw];
    *reinterpret_cast<f32x4 *>(op + 4 * g) = v;
  }
}

struct Grid {
  int tiles, tiles_per_xcd;
  dim3 grid;
};

Grid make_grid(int B, int hw, int D, int zchunks) {
  Grid g;
  g.tiles = casmvs::ceil_div(hw, kThreads);
  g.tiles_per_xcd = casmvs::ceil_div(g.tiles, 8);
  g.grid = dim3((unsigned)(8 * g.tiles_per_xcd * D), (unsigned)B, (unsigned)zchunks);
  return g;
}

}  // namespace

extern "C" int casmvs_homo_warp_f32(const float *src, const float *proj, const float *depth,
                                    float *out, int B, int C, int H, int W, int D, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(src && proj && depth && out, "homo_warp: null pointer");
  CASMVS_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1 && D > 0, "homo_warp: bad shape B=%d C=%d H=%d W=%d D=%d", B, C, H, W, D);
  CASMVS_REQUIRE(B <= 65535, "homo_warp: B > 65535");
  Grid g = make_grid(B, H * W, D, 1);
  hipLaunchKernelGGL(homo_warp_kernel, g.grid, dim3(kThreads), 0, (hipStream_t)stream, src, proj,
                     depth, out, C, H, W, D, g.tiles, g.tiles_per_xcd);
  return casmvs::check_launch("homo_warp_kernel");
}

template <int CH>
static int launch_var(const float *feats, const float *proj, const float *depth, float *out, int B,
                      int V, int C, int h, int w, int D, void *stream) {
  Grid g = make_grid(B, h * w, D, C / CH);
  hipLaunchKernelGGL((costvol_kernel<CH, 0>), g.grid, dim3(kThreads), 0, (hipStream_t)stream,
                     feats, proj, depth, out, V, C, 1, h, w, D, g.tiles, g.tiles_per_xcd);
  return casmvs::check_launch("costvol_var_kernel");
}

extern "C" int casmvs_costvol_var_f32(const float *feats, const float *proj, const float *depth,
                                      float *out, int B, int V, int C, int h, int w, int D,
                                      void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(feats && proj && depth && out, "costvol_var: null pointer");
  CASMVS_REQUIRE(B > 0 && V >= 2 && C > 0 && h > 1 && w > 1 && D > 0,
                 "costvol_var: bad shape B=%d V=%d C=%d h=%d w=%d D=%d", B, V, C, h, w, D);
  CASMVS_REQUIRE(B <= 65535, "costvol_var: B > 65535");
  if (C % 32 == 0) return launch_var<32>(feats, proj, depth, out, B, V, C, h, w, D, stream);
  if (C % 16 == 0) return launch_var<16>(feats, proj, depth, out, B, V, C, h, w, D, stream);
  if (C % 8 == 0) return launch_var<8>(feats, proj, depth, out, B, V, C, h, w, D, stream);
  if (C % 4 == 0) return launch_var<4>(feats, proj, depth, out, B, V, C, h, w, D, stream);
  return launch_var<1>(feats, proj, depth, out, B, V, C, h, w, D, stream);
}

extern "C" int casmvs_costvol_gwc_f32(const float *feats, const float *proj, const float *depth,
                                      float *out, int B, int V, int C, int G, int h, int w, int D,
                                      void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(feats && proj && depth && out, "costvol_gwc: null pointer");
  CASMVS_REQUIRE(B > 0 && V >= 2 && h > 1 && w > 1 && D > 0 && G > 0,
                 "costvol_gwc: bad shape B=%d V=%d G=%d h=%d w=%d D=%d", B, V, G, h, w, D);
  CASMVS_REQUIRE(B <= 65535, "costvol_gwc: B > 65535");
  if ((C != 8 && C != 16 && C != 32) || C % G != 0)
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "costvol_gwc: C=%d G=%d (need C in {8,16,32}, G | C)", C, G);
  Grid g = make_grid(B, h * w, D, 1);
  dim3 blk(kThreads);
  hipStream_t st = (hipStream_t)stream;
  if (C == 32)
    hipLaunchKernelGGL((costvol_kernel<32, 1>), g.grid, blk, 0, st, feats, proj, depth, out, V, C, G, h, w, D, g.tiles, g.tiles_per_xcd);
  else if (C == 16)
    hipLaunchKernelGGL((costvol_kernel<16, 1>), g.grid, blk, 0, st, feats, proj, depth, out, V, C, G, h, w, D, g.tiles, g.tiles_per_xcd);
  else
    hipLaunchKernelGGL((costvol_kernel<8, 1>), g.grid, blk, 0, st, feats, proj, depth, out, V, C, G, h, w, D, g.tiles, g.tiles_per_xcd);
  return casmvs::check_launch("costvol_gwc_kernel");
}

extern "C" int casmvs_nchw_to_nhwc_f32(const float *in, float *out, int N, int C, int h, int w, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(in && out, "nchw_to_nhwc: null pointer");
  CASMVS_REQUIRE(N > 0 && N <= 65535 && h > 0 && w > 0, "nchw_to_nhwc: bad shape N=%d h=%d w=%d", N, h, w);
  const int hw = h * w;
  dim3 grid((unsigned)casmvs::ceil_div(hw, kThreads), (unsigned)N), blk(kThreads);
  hipStream_t st = (hipStream_t)stream;
  if (C == 8) hipLaunchKernelGGL(nchw_to_nhwc_kernel<8>, grid, blk, 0, st, in, out, hw);
  else if (C == 16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<16>, grid, blk, 0, st, in, out, hw);
  else if (C == 32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<32>, grid, blk, 0, st, in, out, hw);
  else return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "nchw_to_nhwc: C=%d (need 8, 16 or 32)", C);
  return casmvs::check_launch("nchw_to_nhwc_kernel");
}

extern "C" int casmvs_costvol_var_nhwc_f32(const float *feats, const float *proj, const float *depth,
                                           float *out, int B, int V, int C, int h, int w, int D,
                                           void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(feats && proj && depth && out, "costvol_var_nhwc: null pointer");
  CASMVS_REQUIRE(B > 0 && V >= 2 && h > 1 && w > 1 && D > 0,
                 "costvol_var_nhwc: bad shape B=%d V=%d C=%d h=%d w=%d D=%d", B, V, C, h, w, D);
  CASMVS_REQUIRE(B <= 65535, "costvol_var_nhwc: B > 65535");
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(feats) | reinterpret_cast<size_t>(out)) & 15) == 0, "costvol_var_nhwc: feats and out must be 16-byte aligned");
  CASMVS_REQUIRE((size_t)h * w * C < ((size_t)1 << 29), "costvol_var_nhwc: one view's map must hold < 2^29 floats");
  Grid g = make_grid(B, h * w, D, 1);
  dim3 blk(kThreads);
  hipStream_t st = (hipStream_t)stream;
  // register budget (minimum waves per SIMD) A/B-tested per channel count: default / 6 / 8
#define CASMVS_CV(CC, OO) hipLaunchKernelGGL((costvol_nhwc_kernel<CC, 0, OO>), g.grid, blk, 0, st, feats, proj, depth, out, V, 1, h, w, D, g.tiles, g.tiles_per_xcd)
  if (C == 32) CASMVS_CV(32, 1);
  else if (C == 16) CASMVS_CV(16, 6);
  else if (C == 8) CASMVS_CV(8, 8);
  else
#undef CASMVS_CV
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "costvol_var_nhwc: C=%d (need 8, 16 or 32)", C);
  return casmvs::check_launch("costvol_var_nhwc_kernel");
}

extern "C" int casmvs_costvol_gwc_nhwc_f32(const float *feats, const float *proj, const float *depth,
                                           float *out, int B, int V, int C, int G, int h, int w, int D,
                                           void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(feats && proj && depth && out, "costvol_gwc_nhwc: null pointer");
  CASMVS_REQUIRE(B > 0 && V >= 2 && h > 1 && w > 1 && D > 0 && G > 0,
                 "costvol_gwc_nhwc: bad shape B=%d V=%d G=%d h=%d w=%d D=%d", B, V, G, h, w, D);
  CASMVS_REQUIRE(B <= 65535, "costvol_gwc_nhwc: B > 65535");
  CASMVS_REQUIRE((reinterpret_cast<size_t>(feats) & 15) == 0, "costvol_gwc_nhwc: feats must be 16-byte aligned");
  CASMVS_REQUIRE((size_t)h * w * C < ((size_t)1 << 29), "costvol_gwc_nhwc: one view's map must hold < 2^29 floats");
  if ((C != 8 && C != 16 && C != 32) || C % G != 0)
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "costvol_gwc_nhwc: C=%d G=%d (need C in {8,16,32}, G | C)", C, G);
  Grid g = make_grid(B, h * w, D, 1);
  dim3 blk(kThreads);
  hipStream_t st = (hipStream_t)stream;
  if (C == 32)
    hipLaunchKernelGGL((costvol_nhwc_kernel<32, 1>), g.grid, blk, 0, st, feats, proj, depth, out, V, G, h, w, D, g.tiles, g.tiles_per_xcd);
  else if (C == 16)
    hipLaunchKernelGGL((costvol_nhwc_kernel<16, 1>), g.grid, blk, 0, st, feats, proj, depth, out, V, G, h, w, D, g.tiles, g.tiles_per_xcd);
  else
    hipLaunchKernelGGL((costvol_nhwc_kernel<8, 1>), g.grid, blk, 0, st, feats, proj, depth, out, V, G, h, w, D, g.tiles, g.tiles_per_xcd);
  return casmvs::check_launch("costvol_gwc_nhwc_kernel");
}
