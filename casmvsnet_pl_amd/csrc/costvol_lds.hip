// Plane sweep with the source tiles staged in LDS: homo_warp, fused warp + variance, fused warp + group-wise
// correlation, and the partial sums of the view-sharded build.
//
// Reference semantics: models/modules.py:52-92 (homo_warp), models/mvsnet.py:134-172 (aggregation).
//
// Why (measured in round 1, profiles/r01_pmc_costvol_sq_counters.md): the gather kernels of costvol.hip move
// (V-1) * 4 taps * C * 4 bytes per voxel through the CU's texture path (64 B/clk/CU): 8x the bytes of the volume
// they write at V = 3, and that path - not HBM - bounds them at 0.3 of the HBM roof.  The footprint of a tile of
// reference pixels in a source view, over a chunk of depth planes, is a small box (the epipolar segment slides
// ~0.6 px per plane): here that box is read ONCE with coalesced 16-byte loads into LDS (256 B/clk/CU for
// ds_read_b128) and every bilinear tap is an LDS read.
//
// Work item = (tile of TW x TH = 256 reference pixels, chunk of DC depth planes, CS of the C channels):
//   1. every thread (= pixel) computes its taps at the chunk's first and last plane for every staged source
//      view; the union over the workgroup is the view's box (the taps move monotonically along the epipolar
//      line between the two planes).  The box is clipped to the LDS capacity of a view;
//   2. the boxes are staged: pixel-major, CS channels per pixel, pixel stride padded to an ODD number of
//      16-byte units so that 16 lanes reading the same channel group of 16 consecutive pixels hit 16 different
//      bank quads (conflict-free ds_read_b128);
//   3. plane by plane, view by view: taps (same arithmetic as costvol.hip: plane_sweep.h) -> 4 * CS/4 LDS reads
//      -> accumulate in registers.  A tap outside its box (noise-like depth, clipped box) is fetched from the
//      global map by that lane alone: the result never depends on the box;
//   4. the plane's C values per pixel leave through a wave-private LDS transpose as 16-byte stores.
// Results are bit-identical to the gather kernels (same operations in the same order per channel).
#include <climits>

#include "common.h"
#include "plane_sweep.h"

namespace {

using namespace casmvs_dev;

constexpr int kThreads = 256;
constexpr int kMaxViews = 8;   // source views staged at once
constexpr int RS = 65;         // row stride of the transpose buffer (odd: conflict-free)
constexpr int kFixedLds = kMaxViews * 8 * 4 + 4 * kMaxViews * 4 * 4 + 4 * 4 * RS * 4;  // prm + red + tr = 4928 B
static_assert(kFixedLds % 16 == 0, "the boxes must start 16-byte aligned");

enum { MODE_VAR = 0, MODE_GWC = 1, MODE_WARP = 2, MODE_VAR_PART = 3, MODE_GWC_PART = 4 };

struct SweepArgs {
  const float *feats;  // (B, Vtot, h, w, C) pixel-major feature maps; view 0 = reference view (unless MODE_WARP)
  const float *proj;   // (B, pstride, 3, 4)
  const float *depth;  // (B, D, h, w)
  float *out;          // VAR / WARP (B, C, D, h, w); GWC (B, G, D, h, w); VAR_PART: sum volume; GWC_PART (B, G, D, h, w)
  float *out2;         // VAR_PART: sum-of-squares volume
  int Vtot;            // views in `feats`
  int v0, nv;          // staged source views [v0, v0 + nv)
  int pv0, pstride;    // their matrices: proj[b][pv0 + i]
  int with_ref;        // VAR_PART: sums start from ref / ref^2 (1) or from 0 (0)
  int nviews_total;    // V of the variance formula / V - 1 of the correlation (final modes)
  int G, h, w, D;
  int tiles_x, tiles, tiles_per_xcd;
  int cap_units;       // LDS capacity of one view's box in 16-byte units
};

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One plane of CS values per pixel (lane = pixel) -> global (.., c, d, y, x) with lane = (channel, 4 pixels):
// 16-byte stores, TW * 4 bytes contiguous per channel and row.  `tr` = this wave's [4][RS] transpose rows.
template <int CS, int TW>
__device__ __forceinline__ void store_plane_transposed(const float (&vals)[CS], float *tr, float *plane_base,
                                                       size_t chan_stride, int wave, int lane, int tx, int ty, int h, int w) {
  constexpr int TH = kThreads / TW;
  const int q4 = lane & 15, cw = lane >> 4;
  const int tw_ = wave * 64 + 4 * q4;          // workgroup-local pixel index of the first of the 4 pixels
  const int py = ty * TH + tw_ / TW, px = tx * TW + tw_ % TW;
  float *op = plane_base + (size_t)py * w + px;
#pragma unroll
  for (int j = 0; j < CS / 4; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tr[i * RS + lane] = vals[4 * j + i];
    wave_lds_fence();
    const float *row = tr + cw * RS + 4 * q4;
    const f32x4 o{row[0], row[1], row[2], row[3]};
    float *oc = op + (size_t)(4 * j + cw) * chan_stride;
    if (py < h) {
      if ((w & 3) == 0) {
        if (px < w) *reinterpret_cast<f32x4 *>(oc) = o;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (px + i < w) oc[i] = o[i];
      }
    }
    wave_lds_fence();  // the rows are rewritten by the next channel group
  }
}

template <int C, int CS, int MODE, int TW, int DC>
__global__ __launch_bounds__(kThreads) void costvol_lds_kernel(const SweepArgs a) {
  constexpr int TH = kThreads / TW;
  constexpr int GL = CS / 4;                     // 16-byte units per staged pixel
  constexpr int GLP = GL == 1 ? 1 : GL + 1;      // padded (odd) pixel stride in units
  constexpr int NSPLIT = C / CS;
  constexpr int NUB = 8;                         // staging loads in flight per thread
  constexpr bool SQ = MODE == MODE_VAR || MODE == MODE_VAR_PART;
  constexpr bool NEED_REF = MODE != MODE_WARP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int *prm = reinterpret_cast<int *>(smem);                   // [kMaxViews][8]: bx0, by0, bw, bh, q256, r256
  int *red = prm + kMaxViews * 8;                             // [4 waves][kMaxViews][4]
  float *tr_all = reinterpret_cast<float *>(red + 4 * kMaxViews * 4);
  f32x4 *box = reinterpret_cast<f32x4 *>(smem + kFixedLds);   // [nv][cap_units]

  // ---- work item ----------------------------------------------------------------------------------------
  // XCD-aware order (block b runs on XCD b % 8; speed only): XCD k owns the tiles [k, k + 1) * tiles_per_xcd,
  // and all depth chunks / channel splits of a tile - which read the same source region - are adjacent.
  const int nchunk = a.D / DC;
  const int inner = nchunk * NSPLIT;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int tile_local = seq / inner, rem = seq - tile_local * inner;
  const int tile = xcd * a.tiles_per_xcd + tile_local;
  if (tile >= a.tiles) return;
  const int d0 = (rem / NSPLIT) * DC, c0 = (rem % NSPLIT) * CS;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int b = blockIdx.y;
  const int h = a.h, w = a.w, hw = h * w, D = a.D, nv = a.nv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = tx * TW + tid % TW, py = ty * TH + tid / TW;
  const bool valid = px < w && py < h;
  const int pcl = valid ? py * w + px : 0;
  const float xf = (float)px, yf = (float)py;
  const size_t view_floats = (size_t)hw * C;
  const int view_bytes = uniform_int((int)(view_floats * 4));
  const float *fb = uniform_ptr(a.feats + (size_t)b * a.Vtot * view_floats);
  const float *pb = uniform_ptr(a.proj + ((size_t)b * a.pstride + a.pv0) * 12);

  float dv[DC];
  {
    const float *dp = a.depth + ((size_t)b * D + d0) * hw + pcl;
#pragma unroll
    for (int k = 0; k < DC; ++k) dv[k] = dp[(size_t)k * hw];
  }

  // ---- 1. boxes ---------------------------------------------------------------------------------------------
  for (int vi = 0; vi < nv; ++vi) {
    const float *P = pb + vi * 12;
    int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const Taps t = plane_sweep_taps(P, xf, yf, dv[e == 0 ? 0 : DC - 1], w, h);
      if (valid && taps_live(t)) {
        xmn = min(xmn, t.xl); xmx = max(xmx, t.xl + 1);
        ymn = min(ymn, t.yn); ymx = max(ymx, t.ys);
      }
    }
    xmn = wave_min(xmn); xmx = wave_max(xmx); ymn = wave_min(ymn); ymx = wave_max(ymx);
    if (lane == 0) {
      int *r = red + (wave * kMaxViews + vi) * 4;
      r[0] = xmn; r[1] = xmx; r[2] = ymn; r[3] = ymx;
    }
  }
  __syncthreads();
  if (tid < nv) {
    int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      const int *r = red + (wv * kMaxViews + tid) * 4;
      xmn = min(xmn, r[0]); xmx = max(xmx, r[1]); ymn = min(ymn, r[2]); ymx = max(ymx, r[3]);
    }
    if (xmn > xmx) { xmn = 0; xmx = 1; ymn = 0; ymx = 0; }   // nothing of this tile projects into the view
    int bw = xmx - xmn + 1, bh = ymx - ymn + 1;
    const int maxbw = a.cap_units / GLP;                      // host guarantees >= 2
    if (bw > maxbw) bw = maxbw;
    const int maxbh = a.cap_units / (bw * GLP);
    if (bh > maxbh) bh = maxbh;
    const int nsu = bw * GL, q256 = kThreads / nsu;
    int *p = prm + tid * 8;
    p[0] = xmn; p[1] = ymn; p[2] = bw; p[3] = bh; p[4] = q256; p[5] = kThreads - q256 * nsu;
  }
  __syncthreads();

  // ---- 2. staging -----------------------------------------------------------------------------------------------
  for (int vi = 0; vi < nv; ++vi) {
    const int *p = prm + vi * 8;
    const int bx0 = uniform_int(p[0]), by0 = uniform_int(p[1]), bw = uniform_int(p[2]), bh = uniform_int(p[3]);
    const int q256 = uniform_int(p[4]), r256 = uniform_int(p[5]);
    const int nsu = bw * GL, total = bh * nsu, rowu = bw * GLP;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(uniform_ptr(fb + (size_t)(a.v0 + vi) * view_floats)), 0, view_bytes, 0x00020000);
    f32x4 *bx = box + (size_t)vi * a.cap_units;
    int row = tid / nsu, ru = tid - row * nsu;
    for (int base = 0; base < total; base += kThreads * NUB) {
      f32x4 regs[NUB];
      int lo[NUB];
#pragma unroll
      for (int i = 0; i < NUB; ++i) {
        const bool ok = base + tid + kThreads * i < total;
        const int pxx = ru / GL, ch = ru % GL;
        lo[i] = ok ? row * rowu + pxx * GLP + ch : -1;
        if (ok) regs[i] = buf_load4(src, (((by0 + row) * w + bx0 + pxx) * C + c0 + 4 * ch) * 4, 0);
        ru += r256; row += q256;
        if (ru >= nsu) { ru -= nsu; ++row; }
      }
#pragma unroll
      for (int i = 0; i < NUB; ++i)
        if (lo[i] >= 0) bx[lo[i]] = regs[i];
    }
  }

  float ref[CS];
  if (NEED_REF) {
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(fb), 0, view_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < GL; ++j) {
      const f32x4 r = buf_load4(r0, (pcl * C + c0 + 4 * j) * 4, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) ref[4 * j + i] = r[i];
    }
  }
  __syncthreads();

  // ---- 3. plane by plane ----------------------------------------------------------------------------------------
  float *tr = tr_all + wave * (4 * RS);
  const float rV = 1.0f / (float)a.nviews_total;
#pragma unroll
  for (int k = 0; k < DC; ++k) {
    float s[CS], q[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) {
      if (MODE == MODE_VAR || (MODE == MODE_VAR_PART && a.with_ref)) {
        s[c] = ref[c];                 // volume_sum = ref_volume            (mvsnet.py:140)
        q[c] = ref[c] * ref[c];        // volume_sq_sum = ref_volume ** 2    (mvsnet.py:141)
      } else {
        s[c] = 0.0f;                   // volume_sum = 0                     (mvsnet.py:144)
        q[c] = 0.0f;
      }
    }
    for (int vi = 0; vi < nv; ++vi) {
      const int *p = prm + vi * 8;
      const int bx0 = uniform_int(p[0]), by0 = uniform_int(p[1]), bw = uniform_int(p[2]), bh = uniform_int(p[3]);
      const int rowu = bw * GLP;
      Taps t = plane_sweep_taps(pb + vi * 12, xf, yf, dv[k], w, h);
      if (!valid) t.w_nl = t.w_nr = t.w_sl = t.w_sr = 0.0f;
      const bool live = taps_live(t);
      const int rx = t.xl - bx0, ryn = t.yn - by0, rys = t.ys - by0;
      const bool in = (rx >= 0) & (rx + 1 < bw) & (ryn >= 0) & (rys < bh);   // yn <= ys
      const bool use_lds = live & in;
      // a dead voxel (all weights 0) reads the box origin: staged, finite data
      const int aN = use_lds ? ryn * rowu + rx * GLP : 0, aS = use_lds ? rys * rowu + rx * GLP : 0;
      const f32x4 *bx = box + (size_t)vi * a.cap_units;
      f32x4 n0[GL], n1[GL], s0[GL], s1[GL];
#pragma unroll
      for (int j = 0; j < GL; ++j) {
        n0[j] = bx[aN + j]; n1[j] = bx[aN + GLP + j];
        s0[j] = bx[aS + j]; s1[j] = bx[aS + GLP + j];
      }
      if (live & !in) {   // a tap outside the staged box: this lane gathers from the global map
        const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(uniform_ptr(fb + (size_t)(a.v0 + vi) * view_floats)), 0, view_bytes, 0x00020000);
        const int on0 = ((t.yn * w + t.xl) * C + c0) * 4, os0 = ((t.ys * w + t.xl) * C + c0) * 4;
#pragma unroll
        for (int j = 0; j < GL; ++j) {
          n0[j] = buf_load4(src, on0, 16 * j); n1[j] = buf_load4(src, on0, C * 4 + 16 * j);
          s0[j] = buf_load4(src, os0, 16 * j); s1[j] = buf_load4(src, os0, C * 4 + 16 * j);
        }
      }
#pragma unroll
      for (int j = 0; j < GL; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float val = fmaf(s1[j][i], t.w_sr, fmaf(s0[j][i], t.w_sl, fmaf(n1[j][i], t.w_nr, n0[j][i] * t.w_nl)));
          s[4 * j + i] = s[4 * j + i] + val;
          if (SQ) q[4 * j + i] = fmaf(val, val, q[4 * j + i]);
        }
      }
    }
    // ---- 4. the plane leaves -------------------------------------------------------------------------------------
    const int d = d0 + k;
    if (MODE == MODE_VAR || MODE == MODE_WARP || MODE == MODE_VAR_PART) {
      if (MODE == MODE_VAR) {
#pragma unroll
        for (int c = 0; c < CS; ++c) {
          const float m = s[c] * rV;   // sq/V - (sum/V)^2 (mvsnet.py:167), x/V as x * (1/V)
          s[c] = q[c] * rV - m * m;
        }
      }
      const size_t chan_stride = (size_t)D * hw;
      float *base = a.out + (((size_t)b * C + c0) * D + d) * hw;
      store_plane_transposed<CS, TW>(s, tr, base, chan_stride, wave, lane, tx, ty, h, w);
      if (MODE == MODE_VAR_PART) {
        float *base2 = a.out2 + (((size_t)b * C + c0) * D + d) * hw;
        store_plane_transposed<CS, TW>(q, tr, base2, chan_stride, wave, lane, tx, ty, h, w);
      }
    } else {
      // group-wise correlation (mvsnet.py:170-171): mean over the C/G channels of a group of
      // volume_sum * ref_volume, then / (V - 1); lane = pixel stores TW consecutive x per group
      const int cpg = C / a.G;
      const float fn = (float)cpg, fv = (float)a.nviews_total;
      float *op = a.out + (((size_t)b * a.G + c0 / cpg) * D + d) * hw + pcl;
      float acc = 0.0f;
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < CS; ++c) {   // static register indexing; group boundaries are runtime
        acc = acc + s[c] * ref[c];
        if (++cnt == cpg) {
          const float m = acc / fn;
          if (valid) *op = MODE == MODE_GWC ? m / fv : m;
          op += (size_t)D * hw;
          acc = 0.0f;
          cnt = 0;
        }
      }
    }
  }
}

// sum / sum-of-squares -> variance (mvsnet.py:167), and the scaling of the all-reduced correlation
__global__ __launch_bounds__(kThreads) void var_finalize_kernel(const f32x4 *__restrict__ sum, const f32x4 *__restrict__ sq,
                                                               f32x4 *__restrict__ out, size_t n4, float rV) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n4) return;
  const f32x4 s = sum[i], q = sq[i];
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float m = s[j] * rV;
    o[j] = q[j] * rV - m * m;
  }
  out[i] = o;
}

__global__ __launch_bounds__(kThreads) void gwc_finalize_kernel(const f32x4 *__restrict__ in, f32x4 *__restrict__ out,
                                                               size_t n4, float fv) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n4) return;
  const f32x4 s = in[i];
  out[i] = f32x4{s[0] / fv, s[1] / fv, s[2] / fv, s[3] / fv};
}

struct Plan {
  int cs, tw, dc, cap_units, lds_bytes;
};

// LDS plan: the highest occupancy (3, 2, 1 workgroups per CU) whose per-view capacity still holds the box a
// tile is expected to need ((TW + DC + 4) x (TH + 2) pixels: ~0.6 px of epipolar slide per plane + slack).
bool make_plan(int C, int w, int D, int nv, int G, int mode, Plan &p) {
  if (nv < 1 || nv > kMaxViews) return false;
  p.dc = 8;
  if (D % p.dc != 0) return false;
  p.tw = (w % 64 == 0) ? 64 : 32;
  p.cs = C == 32 ? 16 : C;
  const bool gwc = mode == MODE_GWC || mode == MODE_GWC_PART;
  if (gwc && (C / G) > p.cs) p.cs = C;     // a group must not straddle two channel splits
#ifdef CASMVS_TRACE   // profiling build only: A/B of the tile shape / channel split
  if (const char *e = getenv("CASMVS_CV_TW")) p.tw = atoi(e);
  if (const char *e = getenv("CASMVS_CV_CS")) p.cs = atoi(e);
#endif
  if (C % p.cs != 0 || (p.cs != 4 && p.cs != 8 && p.cs != 16 && p.cs != 32)) return false;
  const int gl = p.cs / 4, glp = gl == 1 ? 1 : gl + 1, th = kThreads / p.tw;
  const int need = (p.tw + p.dc + 4) * (th + 2) * glp;
  static const int budgets[3] = {52 * 1024, 79 * 1024, 158 * 1024};
  for (int i = 0; i < 3; ++i) {
    const int cap = (budgets[i] - kFixedLds) / (nv * 16);
    if (cap >= need) {
      p.cap_units = cap;
      p.lds_bytes = kFixedLds + nv * cap * 16;
      return true;
    }
  }
  return false;
}

template <int C, int CS, int MODE, int TW>
int launch_cfg(const SweepArgs &a, const Plan &p, int B, hipStream_t st) {
  auto kernel = costvol_lds_kernel<C, CS, MODE, TW, 8>;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), 158 * 1024, "costvol_lds_kernel")) return rc;
  const int inner = (a.D / 8) * (C / CS);
  dim3 grid((unsigned)(8 * a.tiles_per_xcd * inner), (unsigned)B);
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads), (size_t)p.lds_bytes, st, a);
  return casmvs::check_launch("costvol_lds_kernel");
}

template <int C, int MODE>
int launch_c(const SweepArgs &a, const Plan &p, int B, hipStream_t st) {
#define CASMVS_TRY(CSV, TWV) \
  if (p.cs == CSV && p.tw == TWV) return launch_cfg<C, CSV, MODE, TWV>(a, p, B, st);
  if constexpr (C >= 32) { CASMVS_TRY(32, 64) CASMVS_TRY(32, 32) }
  if constexpr (C >= 16) { CASMVS_TRY(16, 64) CASMVS_TRY(16, 32) }
  CASMVS_TRY(8, 64) CASMVS_TRY(8, 32)
#undef CASMVS_TRY
  return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "costvol_lds: no kernel for C=%d CS=%d TW=%d", C, p.cs, p.tw);
}

template <int MODE>
int launch_mode(SweepArgs a, int C, int B, hipStream_t st, const char *what) {
  Plan p;
  if (!make_plan(C, a.w, a.D, a.nv, a.G, MODE, p))
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "%s: no LDS plan for C=%d w=%d D=%d views=%d", what, C, a.w, a.D, a.nv);
  const int th = kThreads / p.tw;
  a.tiles_x = casmvs::ceil_div(a.w, p.tw);
  a.tiles = a.tiles_x * casmvs::ceil_div(a.h, th);
  a.tiles_per_xcd = casmvs::ceil_div(a.tiles, 8);
  a.cap_units = p.cap_units;
  if (C == 32) return launch_c<32, MODE>(a, p, B, st);
  if (C == 16) return launch_c<16, MODE>(a, p, B, st);
  if (C == 8) return launch_c<8, MODE>(a, p, B, st);
  return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "%s: C=%d (need 8, 16 or 32)", what, C);
}

int check_common(const char *what, const void *feats, const void *proj, const void *depth, const void *out, int B, int V,
                 int C, int h, int w, int D) {
  CASMVS_REQUIRE(feats && proj && depth && out, "%s: null pointer", what);
  CASMVS_REQUIRE(B > 0 && B <= 65535 && V >= 1 && h > 1 && w > 1 && D > 0, "%s: bad shape B=%d V=%d C=%d h=%d w=%d D=%d", what, B, V, C, h, w, D);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(feats) | reinterpret_cast<size_t>(out)) & 15) == 0, "%s: feats and out must be 16-byte aligned", what);
  CASMVS_REQUIRE((size_t)h * w * C < ((size_t)1 << 29), "%s: one view's map must hold < 2^29 floats", what);
  return CASMVS_OK;
}

}  // namespace

extern "C" int casmvs_costvol_lds_supported(int C, int w, int D, int n_src_views, int G) {
  Plan p;
  if (C != 8 && C != 16 && C != 32) return 0;
  if (G > 1 && C % G != 0) return 0;
  return make_plan(C, w, D, n_src_views, G > 1 ? G : 1, G > 1 ? MODE_GWC : MODE_VAR, p) ? 1 : 0;
}

extern "C" int casmvs_costvol_var_lds_f32(const float *feats, const float *proj, const float *depth, float *out, int B,
                                          int V, int C, int h, int w, int D, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("costvol_var_lds", feats, proj, depth, out, B, V, C, h, w, D)) return rc;
  CASMVS_REQUIRE(V >= 2, "costvol_var_lds: V=%d", V);
  SweepArgs a{feats, proj, depth, out, nullptr, V, 1, V - 1, 0, V - 1, 1, V, 1, h, w, D, 0, 0, 0, 0};
  return launch_mode<MODE_VAR>(a, C, B, (hipStream_t)stream, "costvol_var_lds");
}

extern "C" int casmvs_costvol_gwc_lds_f32(const float *feats, const float *proj, const float *depth, float *out, int B,
                                          int V, int C, int G, int h, int w, int D, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("costvol_gwc_lds", feats, proj, depth, out, B, V, C, h, w, D)) return rc;
  CASMVS_REQUIRE(V >= 2 && G >= 1 && C % G == 0, "costvol_gwc_lds: V=%d C=%d G=%d", V, C, G);
  SweepArgs a{feats, proj, depth, out, nullptr, V, 1, V - 1, 0, V - 1, 0, V - 1, G, h, w, D, 0, 0, 0, 0};
  return launch_mode<MODE_GWC>(a, C, B, (hipStream_t)stream, "costvol_gwc_lds");
}

extern "C" int casmvs_homo_warp_nhwc_f32(const float *src, const float *proj, const float *depth, float *out, int B,
                                         int C, int H, int W, int D, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("homo_warp_nhwc", src, proj, depth, out, B, 1, C, H, W, D)) return rc;
  SweepArgs a{src, proj, depth, out, nullptr, 1, 0, 1, 0, 1, 0, 1, 1, H, W, D, 0, 0, 0, 0};
  return launch_mode<MODE_WARP>(a, C, B, (hipStream_t)stream, "homo_warp_nhwc");
}

extern "C" int casmvs_costvol_partial_var_f32(const float *feats, const float *proj, const float *depth, float *sum,
                                              float *sq, int B, int V, int C, int h, int w, int D, int view_begin,
                                              int view_end, int include_ref, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("costvol_partial_var", feats, proj, depth, sum, B, V, C, h, w, D)) return rc;
  CASMVS_REQUIRE(sq && (reinterpret_cast<size_t>(sq) & 15) == 0, "costvol_partial_var: sq must be a 16-byte aligned pointer");
  CASMVS_REQUIRE(1 <= view_begin && view_begin < view_end && view_end <= V, "costvol_partial_var: source views [%d, %d) of V=%d", view_begin, view_end, V);
  SweepArgs a{feats, proj, depth, sum, sq, V, view_begin, view_end - view_begin, view_begin - 1, V - 1, include_ref ? 1 : 0, V, 1, h, w, D, 0, 0, 0, 0};
  return launch_mode<MODE_VAR_PART>(a, C, B, (hipStream_t)stream, "costvol_partial_var");
}

extern "C" int casmvs_costvol_partial_gwc_f32(const float *feats, const float *proj, const float *depth, float *out,
                                              int B, int V, int C, int G, int h, int w, int D, int view_begin,
                                              int view_end, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("costvol_partial_gwc", feats, proj, depth, out, B, V, C, h, w, D)) return rc;
  CASMVS_REQUIRE(G >= 1 && C % G == 0, "costvol_partial_gwc: C=%d G=%d", C, G);
  CASMVS_REQUIRE(1 <= view_begin && view_begin < view_end && view_end <= V, "costvol_partial_gwc: source views [%d, %d) of V=%d", view_begin, view_end, V);
  SweepArgs a{feats, proj, depth, out, nullptr, V, view_begin, view_end - view_begin, view_begin - 1, V - 1, 0, V - 1, G, h, w, D, 0, 0, 0, 0};
  return launch_mode<MODE_GWC_PART>(a, C, B, (hipStream_t)stream, "costvol_partial_gwc");
}

extern "C" int casmvs_costvol_var_finalize_f32(const float *sum, const float *sq, float *out, size_t n, int V, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(sum && sq && out && V >= 1, "costvol_var_finalize: null pointer / V=%d", V);
  CASMVS_REQUIRE(n % 4 == 0 && ((reinterpret_cast<size_t>(sum) | reinterpret_cast<size_t>(sq) | reinterpret_cast<size_t>(out)) & 15) == 0,
                 "costvol_var_finalize: n must be a multiple of 4 and the pointers 16-byte aligned");
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(var_finalize_kernel, dim3((unsigned)((n4 + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4 *>(sum), reinterpret_cast<const f32x4 *>(sq), reinterpret_cast<f32x4 *>(out), n4,
                     1.0f / (float)V);
  return casmvs::check_launch("var_finalize_kernel");
}

extern "C" int casmvs_costvol_gwc_finalize_f32(const float *in, float *out, size_t n, int V, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(in && out && V >= 2, "costvol_gwc_finalize: null pointer / V=%d", V);
  CASMVS_REQUIRE(n % 4 == 0 && ((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out)) & 15) == 0,
                 "costvol_gwc_finalize: n must be a multiple of 4 and the pointers 16-byte aligned");
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(gwc_finalize_kernel, dim3((unsigned)((n4 + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4 *>(in), reinterpret_cast<f32x4 *>(out), n4, (float)(V - 1));
  return casmvs::check_launch("gwc_finalize_kernel");
}
