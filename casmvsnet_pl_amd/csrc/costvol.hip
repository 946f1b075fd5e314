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

namespace {

constexpr int kThreads = 256;

// The 2x2 bilinear footprint is fetched as two 8-byte row pairs (x, x+1): half the vector-memory
// instructions of four scalar taps.  o_n / o_s are the element offsets of the LEFT element of
// the north / south pair inside one (h, w) channel plane (clamped so both elements exist);
// the four weights already carry ATen's zeros padding (weight 0 for a tap outside the image).
struct Taps {
  int o_n, o_s;
  float w_nl, w_nr, w_sl, w_sr;  // north-left, north-right, south-left, south-right
};

typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));  // 4-byte aligned pair

// Coordinates + bilinear taps of ref pixel (x, y) at depth dv in the source view whose
// (P_src @ inv(P_ref))[:3] is P (12 floats, row-major 3x4).  Follows modules.py:59-89 and
// ATen's grid_sampler (bilinear, zeros padding, align_corners=True) operation by operation.
__device__ __forceinline__ Taps plane_sweep_taps(const float *__restrict__ P, float xf, float yf,
                                                 float dv, int W, int H) {
  // src_grid_d = R @ (x, y, 1)^T + T / depth                       (modules.py:72)
  float rx = fmaf(P[2], 1.0f, fmaf(P[1], yf, P[0] * xf));
  float ry = fmaf(P[6], 1.0f, fmaf(P[5], yf, P[4] * xf));
  float rz = fmaf(P[10], 1.0f, fmaf(P[9], yf, P[8] * xf));
  float qx = rx + P[3] / dv;
  float qy = ry + P[7] / dv;
  float qz = rz + P[11] / dv;
  // negative depth -> somewhere outside the image                  (modules.py:76-79)
  if (qz <= 1e-7f) {
    qx = (float)W;
    qy = (float)H;
    qz = 1.0f;
  }
  float u = qx / qz;  // modules.py:81
  float v = qy / qz;
  // scale to [-1, 1] (modules.py:83-84) and ATen's un-normalisation (align_corners=True)
  float gx = u / ((float)(W - 1) * 0.5f) - 1.0f;
  float gy = v / ((float)(H - 1) * 0.5f) - 1.0f;
  float ix = ((gx + 1.0f) * 0.5f) * (float)(W - 1);
  float iy = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
  float x0 = floorf(ix), y0 = floorf(iy);
  float tw = ix - x0, te = 1.0f - tw;  // ATen CPU kernel: w = x - x_w, e = 1 - w
  float tn = iy - y0, ts = 1.0f - tn;
  // Bounds tests in float: NaN / +-inf / huge coordinates fail every comparison, so the tap is
  // dropped exactly like ATen's zeros padding and is never converted to an int index.
  // x: left element of the pair is column xl = clamp(x0, 0, W-2); columns x0 and x0+1 carry
  // weights te and tw when they exist.
  const float fW = (float)W, fH = (float)H;
  float wl, wr;
  int xl;
  if (x0 >= 0.0f && x0 <= fW - 2.0f) {        // both columns inside
    xl = (int)x0; wl = te; wr = tw;
  } else if (x0 == -1.0f) {                   // only column x0+1 = 0 inside
    xl = 0; wl = tw; wr = 0.0f;
  } else if (x0 == fW - 1.0f) {               // only column x0 = W-1 inside
    xl = W - 2; wl = 0.0f; wr = te;
  } else {
    xl = 0; wl = 0.0f; wr = 0.0f;
  }
  const bool y0_in = (y0 >= 0.0f) && (y0 <= fH - 1.0f);
  const bool y1_in = (y0 >= -1.0f) && (y0 <= fH - 2.0f);
  const int yi0 = y0_in ? (int)y0 : 0;
  const int yi1 = y1_in ? (int)y0 + 1 : 0;
  const float wn = y0_in ? ts : 0.0f, wsth = y1_in ? tn : 0.0f;
  Taps t;
  t.o_n = yi0 * W + xl;
  t.o_s = yi1 * W + xl;
  t.w_nl = wl * wn;   // ATen: nw = e * s, ne = w * s, sw = e * n, se = w * n
  t.w_nr = wr * wn;
  t.w_sl = wl * wsth;
  t.w_sr = wr * wsth;
  return t;
}

__device__ __forceinline__ float sample(const float *__restrict__ plane, const Taps &t) {
  const f32x2u n = *reinterpret_cast<const f32x2u *>(plane + t.o_n);
  const f32x2u s = *reinterpret_cast<const f32x2u *>(plane + t.o_s);
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
  for (int c = 0; c < C; ++c) op[(size_t)c * D * hw] = sample(sp + (size_t)c * hw, t);
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
      float val = sample(sp + (size_t)c * hw, t);
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
