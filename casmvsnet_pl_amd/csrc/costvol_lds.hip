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
//      line between the two planes), in PADDED image coordinates (columns -1 .. W, rows -1 .. H: positions outside the
//      image are staged as zeros, which is what ATen's zeros padding reads there).  The box is clipped to the LDS capacity of a view;
//   2. the boxes are staged: pixel-major, CS channels per pixel, pixel stride padded to an ODD number of
//      16-byte units so that 16 lanes reading the same channel group of 16 consecutive pixels hit 16 different
//      bank quads (conflict-free ds_read_b128);
//   3. plane by plane, view by view: taps (same arithmetic as costvol.hip: plane_sweep.h) -> 4 * CS/4 LDS reads
//      -> accumulate in registers.  A tap outside its box (noise-like depth, clipped box) is fetched from the
//      global map by that lane alone: the result never depends on the box;
//   4. the plane's C values per pixel leave through a wave-private LDS transpose as 16-byte stores.
// Results are bit-identical to the gather kernels (same operations in the same order per channel).
#include <algorithm>
#include <climits>

#include "common.h"
#include "plane_sweep.h"

namespace {

using namespace casmvs_dev;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int kMaxViews = 8;   // source views staged at once
#ifndef CASMVS_CV_RS
#define CASMVS_CV_RS 68
#endif
constexpr int RS = CASMVS_CV_RS;   // row stride of the transpose buffer.  68: rows stay 16-byte aligned, so a lane's four pixels are ONE
                                   // ds_read_b128 at consecutive 16-byte units (conflict-free); the writes (lane = consecutive floats) are
                                   // conflict-free for any stride.  (65, round 2: misaligned rows -> two ds_read2_b32 with a 16-byte lane
                                   // stride = bank conflicts: SQ_LDS_BANK_CONFLICT was 44 % of the LDS-active cycles.)
constexpr int kMaxPG = 2;      // plane groups: a tile's 256 pixels are worked on by PG x 4 waves, each group its own planes
// prm + red + pmat + tr (one [4][RS] transpose buffer per wave)
constexpr int fixed_lds(int pg) { return kMaxViews * 8 * 4 + 4 * kMaxViews * 4 * 4 + kMaxViews * 12 * 4 + pg * 4 * 4 * RS * 4; }
static_assert(fixed_lds(1) % 16 == 0 && fixed_lds(2) % 16 == 0, "the boxes must start 16-byte aligned");

#ifdef CASMVS_TRACE
// Profiling build only (tools/gpu_cv_trace.py): thread 0 of every 32nd workgroup (batch element 0) stamps the shader
// clock at the phase boundaries of costvol_lds_kernel; read back by casmvs_cv_trace_read.
__device__ unsigned long long g_cv_trace[64 * 32];
#define CV_STAMP()                                                                                      \
  do {                                                                                                  \
    if (threadIdx.x == 0 && (blockIdx.x & 31) == 0 && blockIdx.y == 0 && (blockIdx.x >> 5) < 64 && tr_n < 32) \
      g_cv_trace[(blockIdx.x >> 5) * 32 + tr_n++] = __builtin_readcyclecounter();                       \
  } while (0)
#else
#define CV_STAMP() do {} while (0)
#endif

enum { MODE_VAR = 0, MODE_GWC = 1, MODE_WARP = 2, MODE_VAR_PART = 3, MODE_GWC_PART = 4,
       MODE_WARP_NCHW = 5 };   // homo_warp on the reference's own (B, C, H, W) source: the box is staged from the channel planes (no layout pass)
constexpr bool is_warp(int mode) { return mode == MODE_WARP || mode == MODE_WARP_NCHW; }

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
  int B;               // batch elements (the persistent workgroups walk (batch element, tile, chunk, split) items)
  int cap_units;       // LDS capacity of one view's box in 16-byte units
  int ablate;          // profiling build only (-DCASMVS_TRACE): bit 0 no volume stores, 1 no LDS tap reads, 2 taps of plane 0, 3 stores only
};

// Wave-wide min / max with a wave-uniform result.  DPP inside the rows of 16 lanes (lane ^ 1, lane ^ 2, mirror of 8,
// mirror of 16: four VALU instructions with a lane-permuting operand), then the four rows through v_readlane / s_min.
// (Round 2 used six __shfl_xor steps = six dependent ds_bpermute_b32 per value: 24 LDS-crossbar round trips per view in
// the prologue of every workgroup.)
template <int CTRL>
__device__ __forceinline__ int dpp_perm(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_min(int v) {
  v = min(v, dpp_perm<0xB1>(v));    // quad_perm [1,0,3,2]
  v = min(v, dpp_perm<0x4E>(v));    // quad_perm [2,3,0,1]
  v = min(v, dpp_perm<0x141>(v));   // row_half_mirror
  v = min(v, dpp_perm<0x140>(v));   // row_mirror
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max(int v) {
  v = max(v, dpp_perm<0xB1>(v));
  v = max(v, dpp_perm<0x4E>(v));
  v = max(v, dpp_perm<0x141>(v));
  v = max(v, dpp_perm<0x140>(v));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One plane of CS values per pixel (lane = pixel) -> global (.., c, d, y, x) with lane = (channel, 4 pixels):
// 16-byte stores, TW * 4 bytes contiguous per channel and row.  `tr` = this wave's [4][RS] transpose rows.
// Buffer stores: the per-lane offset `voff` ((channel lane>>4 of the split, pixel) - or beyond the buffer for a
// pixel outside the image: the hardware drops the store) is computed ONCE per kernel; what changes per plane and
// channel group is a scalar offset.  No per-store address arithmetic, no 64-bit pointers in VGPRs.
struct PlaneStore {
  __amdgpu_buffer_rsrc_t rsrc;   // the batch element's whole volume
  int voff;                      // bytes; past the buffer for lanes that store nothing
};

template <int TW>
__device__ __forceinline__ PlaneStore make_plane_store(float *batch_base, size_t batch_bytes, int c0, int D, int hw, int h, int w,
                                                       int wave, int lane, int tx, int ty) {
  constexpr int TH = kThreads / TW;
  const int q4 = lane & 15, cw = lane >> 4;
  const int tw_ = wave * 64 + 4 * q4;          // workgroup-local pixel index of the first of the 4 pixels
  PlaneStore ps;
  const int py = ty * TH + tw_ / TW, px = tx * TW + tw_ % TW;
  ps.rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(batch_base), 0, uniform_int((int)batch_bytes), 0x00020000);
  const bool ok = py < h && px < w;   // w % 4 == 0: the 4 pixels of a lane are inside or outside together
  ps.voff = ok ? ((c0 + cw) * D * hw + py * w + px) * 4 : -16;   // -16 = 0xfffffff0: past any buffer
  return ps;
}

#ifndef CASMVS_CV_STORE_AUX
#define CASMVS_CV_STORE_AUX 2   // cache-policy bits of the volume stores (gfx94x: 1 = sc0, 2 = nt, 16 = sc1).  nt = streaming: the
                                // volume does not displace the boxes / depth maps the kernel re-reads from L2 (level 1, batch 2:
                                // 133 -> 104 us, profiles/r02_s3_costvol_store_policy.txt)
#endif
template <int CS>
__device__ __forceinline__ void store_plane_transposed(const float (&vals)[CS], float *tr, const PlaneStore &ps, int lane,
                                                       int D, int hw, int d, int w) {
  const int q4 = lane & 15, cw = lane >> 4;
#pragma unroll
  for (int j = 0; j < CS / 4; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tr[i * RS + lane] = vals[4 * j + i];
    wave_lds_fence();
    const float *row = tr + cw * RS + 4 * q4;
    const f32x4 o = RS % 4 == 0 ? *reinterpret_cast<const f32x4 *>(row) : f32x4{row[0], row[1], row[2], row[3]};
    const int soff = uniform_int((4 * j * D + d) * hw * 4);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ps.rsrc, ps.voff, soff, CASMVS_CV_STORE_AUX);   // w % 4 == 0 (host)
    // STORE-DATA HAZARD (found in round 5, profiles/r05_store_data_hazard.md): the store reads its 16 bytes per lane over several cycles after issue, and a
    // VALU write of one of its data registers in the very next issue slot lands in some lanes of the stored value (here: the depth register rotation put a
    // hypothesis into element 1 of lanes 12-15 of every 16 - channels 12-15 of four pixels per wave and plane, different ones every run).  The ISA lists
    // the pair (wide store, then a write of its data registers) as needing wait states and LLVM pads it - except for a store whose soffset is a scalar
    // REGISTER (GCNHazardRecognizer::createsVALUHazard), the form used here.  Two wait states that no write of `o` can move in front of (the statement
    // reads the registers; the memory clobber keeps it behind the store); tools/store_hazard_lint.py checks the whole library's device code for the pair.
#ifndef HIPEMU_LDS_BYTES   // (tests/hipemu: the host has no such instruction, and needs none)
    asm volatile("s_nop 1" ::"v"(o) : "memory");
#endif
    wave_lds_fence();  // the rows are rewritten by the next channel group
  }
}

// The plane's CS values per pixel -> global.  Default: through the wave-private transpose (16-byte stores).
// -DCASMVS_CV_DIRECT (A/B build): lane = pixel stores one dword per channel, TW * 4 contiguous bytes per
// instruction, no LDS round trip.
template <int CS>
__device__ __forceinline__ void store_plane(const float (&vals)[CS], float *tr, const PlaneStore &ps, int lane, int D, int hw,
                                            int d, int w, int direct_voff) {
#ifdef CASMVS_CV_DIRECT
  (void)tr; (void)lane; (void)w;
#pragma unroll
  for (int c = 0; c < CS; ++c)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vals[c]), ps.rsrc, direct_voff,
                                          uniform_int((c * D + d) * hw * 4), 0);
#else
  (void)direct_voff;
  store_plane_transposed<CS>(vals, tr, ps, lane, D, hw, d, w);
#endif
}

// LDS layout of a staged box row, in 16-byte units: pixel px (box-relative) starts at unit(px), its CS/4 channel
// groups follow contiguously, a row takes row_units(bw).  Measured with tools/probes/lds_probe.hip (ds_read_b128,
// lane i reads pixel p + i): a pixel stride of 2 or 4 units is 2- / 4-way bank conflicted, any ODD stride is
// conflict free, and so is stride 2 with one extra unit every 8 pixels / stride 4 with one extra unit every 4 pixels
// (pixel 4 a + r starts at unit 17 a + 4 r: 16 consecutive pixels cover the 16 bank quads; MI355X_MICROARCH.md: a ds_read_b128 is served in four groups
// of 16 lanes).
//   CS = 8 : unit(px) = 2 px + (px >> 3)    (6 % padding: what lets four workgroups share a CU's LDS)
//   CS = 16: unit(px) = 4 px + (px >> 2)    (6 % padding - round 5; the odd stride 5 px was 25 %: the difference is what lets a workgroup sweep 16
//                                            planes of a tile instead of 8 on the same 2 x 35 KiB, i.e. half the prologues per volume)
//   CS = 32: unit(px) = 9 px                (odd stride: 12.5 % padding)
template <int CS>
struct BoxLayout {
  static constexpr int GL = CS / 4;
  static __device__ __forceinline__ int unit(int px) { return CS == 8 ? 2 * px + (px >> 3) : (CS == 16 ? 4 * px + (px >> 2) : px * (GL + 1)); }
  static __device__ __forceinline__ int row_units(int bw) { return CS == 8 ? 2 * bw + (bw >> 3) + 1 : (CS == 16 ? 4 * bw + (bw >> 2) + 1 : bw * (GL + 1)); }
  // widest box of a given capacity (units) that still has one row; rows of a box of width bw
  static __host__ __device__ constexpr int units_per_px_bound() { return CS == 8 ? 3 : GL + 1; }
  static constexpr bool kConstantNeighbour = CS == 32;   // unit(px + 1) - unit(px) is a constant (an immediate offset of the right column's reads)
};

// Box of one staged view, wave-uniform (SGPRs).
struct Box {
  int bx0, by0, bw, bh;
};

__device__ __forceinline__ Box read_box(const int *prm, int vi) {
  const int *p = prm + vi * 8;
  return Box{uniform_int(p[0]), uniform_int(p[1]), uniform_int(p[2]), uniform_int(p[3])};
}

// One source view's contribution to one plane of this thread's pixel: coordinates -> 4 * CS/4 LDS reads (or, for a lane
// whose footprint is not inside the staged box, global gathers) -> s += val, q += val^2.
//
// ZERO-PADDED BOXES (round 5).  ATen's grid_sample drops a tap outside the image (modules.py:87-89, zeros padding): the same value as a tap that reads
// 0 - so the staged box lives in PADDED image coordinates (columns -1 .. W, rows -1 .. H; positions outside the image are staged as zeros) and the
// plane loop needs no per-tap bounds logic at all: a footprint (x0, x0 + 1) x (y0, y0 + 1) is either inside the box - then its four taps are read with
// the plain weights (1 - tw)(1 - tn), tw (1 - tn), (1 - tw) tn, tw tn, and the taps outside the image contribute 0 * w = 0 - or it is not, which is
// wave-uniformly rare.  The per-tap form (plane_sweep.h: taps_from_coords: ~40 compares / selects / clamps per (pixel, plane, view), every one a 4.3-cycle
// instruction) cost more vector issue than the interpolation of 16 channels; it survives in the fallback, where a lane whose footprint is inside the
// IMAGE but not inside the box gathers from global memory.  Results: the same products in the same order for every tap inside the image (the weight of an
// in-image tap is the same product of the same operands), a zero term where the per-tap form has a zero-weight term - equal values, as asserted against
// the gather kernels.
//
// gfx9 has ONE in-order counter (vmcnt) for vector-memory loads AND stores: a wait for a load that was issued
// after the previous plane's volume stores also waits for those stores to reach HBM.  The first version of this
// kernel paid that ~1-2 us once per plane and view (the compiler waits for the rarely taken gather branch's
// registers at the merge point, unconditionally) and was slower than the gather kernels.  Hence: the gather
// branch is wave-uniform (skipped by a scalar branch) and drains its own loads with an explicit s_waitcnt
// INSIDE the branch, so the steady state carries no vector-memory wait at all.
template <int C, int CS, bool SQ, int JBM = 2, bool NCHW = false>
__device__ __forceinline__ void accumulate_view(const float *P, float xf, float yf, float dvk, int w, int h, bool valid,
                                                const Box &bx_, const f32x4 *bx, const float *view_map, int view_bytes,
                                                int c0, f32x2 (&s)[CS / 2], f32x2 (&q)[CS / 2], int abl) {
  constexpr int GL = CS / 4;
  using L = BoxLayout<CS>;
  const int rowu = L::row_units(bx_.bw);
  const SweepCoords sc = plane_sweep_coords(P, xf, yf, dvk, w, h);
  // box-relative position of the footprint's north-west corner.  NaN / -inf -> -8 in x, -2 in y: left of / above every box (by0 >= -1; bx0 >= -1, and >= -4
  // where the channel-plane staging rounds the box down to a quad of x: a clamp at -2 would put a NaN column INSIDE such a box with NaN weights); +inf / huge
  // saturate: the unsigned compares fail for all of them.  A finite x0 in [-8, -2) of a quad-rounded box reads its staged zero padding, as before
  const int rx = (int)fmaxf(sc.x0, -8.0f) - bx_.bx0, ry = (int)fmaxf(sc.y0, -2.0f) - bx_.by0;
  const bool inbox = ((unsigned)rx < (unsigned)(bx_.bw - 1)) & ((unsigned)ry < (unsigned)(bx_.bh - 1)) & valid;   // columns rx, rx + 1 and rows ry, ry + 1 staged
  // a lane without a staged footprint reads the box origin (staged, finite data) with zero weights
  // (left, right) column weights as ONE register pair: the two rows' weights are two packed multiplies with a broadcast operand.  Without a staged
  // footprint all four weights are exactly 0 - also when the coordinates are NaN / inf (0 * NaN is NaN: the row fraction is replaced as well)
  const f32x2 tlr{inbox ? 1.0f - sc.tw : 0.0f, inbox ? sc.tw : 0.0f};
  const float tn_ = inbox ? sc.tn : 0.0f, ts_ = 1.0f - tn_;
  f32x2 wN = tlr * f32x2{ts_, ts_}, wS = tlr * f32x2{tn_, tn_};   // ATen: nw = e * s, ne = w * s | sw = e * n, se = w * n
  const int uL = inbox ? ry * rowu + L::unit(rx) : 0;
  const int uR = L::kConstantNeighbour ? uL + (GL + 1) : (inbox ? ry * rowu + L::unit(rx + 1) : L::unit(1));   // (odd pixel stride: the neighbour is a constant away)
  const int dS = rowu;   // the south row is staged whenever the north row is (ry + 1 < bh); the origin's south neighbour exists (bh >= 2)
  // channel groups in batches of JB (8 channels): at most 8 ds_read_b128 = 32 registers of tap data live at a time -
  // what keeps the 8-wave (PG = 2) form inside 128 VGPRs; the scheduling barriers stop the compiler from hoisting the
  // next batch's reads above this batch's arithmetic
  constexpr int JB = GL < JBM ? GL : JBM;
#pragma unroll
  for (int jb = 0; jb < GL; jb += JB) {
    f32x4 n0[JB], n1[JB], s0[JB], s1[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      if (abl & 2) {   // ablation: the timing without the LDS tap reads (wrong results)
        n0[j] = f32x4{wN.x, xf, yf, dvk}; n1[j] = n0[j]; s0[j] = n0[j]; s1[j] = n0[j];
        continue;
      }
      n0[j] = bx[uL + jb + j]; n1[j] = bx[uR + jb + j];
      s0[j] = bx[uL + dS + jb + j]; s1[j] = bx[uR + dS + jb + j];
    }
    if (__builtin_amdgcn_ballot_w64(valid & !inbox) != 0) {   // a footprint outside its box: mostly outside the image as well (tiles whose frustum leaves the view)
      asm volatile("" ::: "memory");   // (a side effect: keeps the ~35 instructions of the bounds logic INSIDE the branch - the compiler speculated them into the plane loop)
      const Taps t = taps_from_coords(sc, w, h);
      const bool outside = valid & !inbox & taps_live(t);
      if (__builtin_amdgcn_ballot_w64(outside) != 0) {   // rare: noise-like depth, or a box clipped by the LDS capacity
        const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(view_map), 0, view_bytes, 0x00020000);
        // every lane loads (a valid image address either way), only the lanes outside their box keep the result - together with ATen's
        // bounds-checked weights; one tap at a time: the rare path must not raise the register count of the common one
        // pixel-major map: the 4 channels of a group are one 16-byte load, the right column is C floats on; channel-plane map
        // (NCHW): four dword loads h * w floats apart, the right column one float on
        const int chs = NCHW ? h * w * 4 : 4, pxs = NCHW ? 4 : C * 4;   // byte strides of a channel / of a pixel
        const int on0 = NCHW ? ((c0 + 4 * jb) * h * w + t.yn * w + t.xl) * 4 : ((t.yn * w + t.xl) * C + c0 + 4 * jb) * 4;
        const int os0 = NCHW ? ((c0 + 4 * jb) * h * w + t.ys * w + t.xl) * 4 : ((t.ys * w + t.xl) * C + c0 + 4 * jb) * 4;
#define CASMVS_GATHER_TAP(dst, voff, imm)                                        \
        {                                                                          \
          f32x4 g[JB];                                                             \
          _Pragma("unroll") for (int j = 0; j < JB; ++j) {                         \
            if (NCHW) {                                                            \
              _Pragma("unroll") for (int i = 0; i < 4; ++i)                        \
                g[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src, voff + (imm) + (4 * j + i) * chs, 0, 0)); \
            } else {                                                               \
              g[j] = buf_load4(src, voff, (imm) + 16 * j);                         \
            }                                                                      \
          }                                                                        \
          __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0) only, INSIDE the branch */ \
          _Pragma("unroll") for (int j = 0; j < JB; ++j)                           \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) dst[j][i] = outside ? g[j][i] : dst[j][i]; \
        }
        CASMVS_GATHER_TAP(n0, on0, 0)
        CASMVS_GATHER_TAP(n1, on0, pxs)
        CASMVS_GATHER_TAP(s0, os0, 0)
        CASMVS_GATHER_TAP(s1, os0, pxs)
#undef CASMVS_GATHER_TAP
        wN = f32x2{outside ? t.w_nl : wN.x, outside ? t.w_nr : wN.y};
        wS = f32x2{outside ? t.w_sl : wS.x, outside ? t.w_sr : wS.y};
      }
    }
    const f32x2 w_nl{wN.x, wN.x}, w_nr{wN.y, wN.y}, w_sl{wS.x, wS.x}, w_sr{wS.y, wS.y};
    // Two channels per instruction (v_pk_mul / v_pk_fma / v_pk_add_f32): a wave issues one instruction every ~5
    // cycles whatever it is, so the instruction COUNT is the wave's run time.  Per lane and channel the operations
    // and their order are those of the scalar form: val = fma(s1, w_sr, fma(s0, w_sl, fma(n1, w_nr, n0 * w_nl))),
    // s += val, q = fma(val, val, q).
#pragma unroll
    for (int j = 0; j < JB; ++j) {
#pragma unroll
      for (int hlf = 0; hlf < 2; ++hlf) {
        const f32x2 a0 = hlf ? n0[j].zw : n0[j].xy, a1 = hlf ? n1[j].zw : n1[j].xy;
        const f32x2 b0 = hlf ? s0[j].zw : s0[j].xy, b1 = hlf ? s1[j].zw : s1[j].xy;
        f32x2 val = a0 * w_nl;
        val = __builtin_elementwise_fma(a1, w_nr, val);
        val = __builtin_elementwise_fma(b0, w_sl, val);
        val = __builtin_elementwise_fma(b1, w_sr, val);
        s[2 * (jb + j) + hlf] = s[2 * (jb + j) + hlf] + val;
        if (SQ) q[2 * (jb + j) + hlf] = __builtin_elementwise_fma(val, val, q[2 * (jb + j) + hlf]);
      }
    }
    if (GL > JB) __builtin_amdgcn_sched_barrier(0);
  }
}

// NV: number of staged source views when known at compile time (their loop is unrolled and the boxes / matrices
// live in SGPRs), 0 = run-time count (rolled loop, box re-read from LDS per view).
// Register budget: the CS = 8 boxes are small enough for three workgroups per CU (168 VGPRs; at 128 - four per CU - the
// compiler spills inside the plane loop, and a scratch reload is a vector-memory load that drains the store queue);
// the larger splits are LDS-bound at two workgroups per CU anyway.
//
// PG (plane groups): the LDS boxes cap the workgroups per CU at 2 (CS = 16) - 8 waves per CU, 2 per SIMD - and a wave
// issues one instruction every ~5-7 cycles whatever it is (tools/probes/valu_probe.hip, clock_probe.hip), so two
// waves leave the SIMD's VALU (one instruction per ~1.7 cycles) mostly idle: measured, the kernel ran at the speed of
// its longest wave, not of any unit.  With PG = 2 the SAME tile and boxes are worked on by 8 waves, waves 4-7 taking
// the upper half of the chunk's planes: twice the waves per byte of LDS.
#ifndef CASMVS_WARP_OCC
#define CASMVS_WARP_OCC 3   // waves per SIMD of the one-view (homo_warp) kernels at CS = 16: 136 VGPRs, one box per workgroup (A/B: 2, 4)
#endif
#ifndef CASMVS_WARP_OCC8
#define CASMVS_WARP_OCC8 4   // the same at CS = 8 (cascade level 0): 111 VGPRs, four 39 KiB workgroups per CU.  Round 6, batch 8, dirtied caches: 268 -> 252 us through the reference signature (0.39 -> 0.42 of HBM), 214 -> 206 pixel-major; batch 1 and hot caches: equal (A/B: 3)
#endif
constexpr int waves_per_simd(int cs, int mode, int pg) {
  return pg == 2 ? 4 : (cs == 8 ? (is_warp(mode) ? CASMVS_WARP_OCC8 : 3) : (is_warp(mode) && cs == 16 ? CASMVS_WARP_OCC : 2));
}

template <int C, int CS, int MODE, int TW, int DC, int NV, int PG>
__global__ __launch_bounds__(kThreads * PG, waves_per_simd(CS, MODE, PG)) void costvol_lds_kernel(const SweepArgs a) {
  constexpr int NT = kThreads * PG;   // threads of the workgroup
  constexpr int kFixedLds = fixed_lds(PG);
  constexpr int TH = kThreads / TW;
  constexpr int GL = CS / 4;                     // 16-byte units per staged pixel
  using L = BoxLayout<CS>;
  constexpr int NSPLIT = C / CS;
  constexpr int NUB = CS == 8 ? 4 : 8;           // staging loads in flight per thread (the 128-VGPR budget of CS = 8)
  constexpr bool SQ = MODE == MODE_VAR || MODE == MODE_VAR_PART;
#ifndef CASMVS_JB2
#define CASMVS_JB2 4   // channel groups (of 4) whose tap data is in flight at once at 2 waves per SIMD: all 16 reads of a
                       // view are issued, the next view's taps are computed under their latency (A/B: 2 = round 2)
#endif
  // tap data in flight: 16 registers per channel group
  constexpr int kJB = waves_per_simd(CS, MODE, PG) >= 4 ? 1 : (waves_per_simd(CS, MODE, PG) == 2 ? CASMVS_JB2 : 2);
  constexpr bool NCHW = MODE == MODE_WARP_NCHW;   // `feats` is (B, 1, C, h, w): channel planes
  constexpr bool NEED_REF = !is_warp(MODE);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int *prm = reinterpret_cast<int *>(smem);                   // [kMaxViews][8]: bx0, by0, bw, bh, q256, r256
  int *red = prm + kMaxViews * 8;                             // [4 waves][kMaxViews][4]
  float *pmat = reinterpret_cast<float *>(red + 4 * kMaxViews * 4);   // [kMaxViews][12] the views' 3x4 matrices
  float *tr_all = pmat + kMaxViews * 12;
  f32x4 *box = reinterpret_cast<f32x4 *>(smem + kFixedLds);   // [nv][cap_units]

#ifdef CASMVS_TRACE
  int tr_n = 0;
#endif
  CV_STAMP();   // 0: kernel start
  // ---- work item ----------------------------------------------------------------------------------------
  // XCD-aware order (block b runs on XCD b % 8; speed only): XCD k owns the tiles [k, k + 1) * tiles_per_xcd,
  // and all depth chunks / channel splits of a tile - which read the same source region - are adjacent.
  // PERSISTENT workgroups (round 5): the launch holds as many workgroups as the chip keeps resident; each walks the items seq, seq + step, ... of its XCD and
  // loads the NEXT item's first / last hypotheses before it enters the plane loop of the current one: the box extents of the next item - the head of the
  // prologue's chain depth -> extents -> boxes -> staging, two dependent memory round trips, a third of an item's time - start without waiting for memory.
  const int nchunk = a.D / DC;
  const int inner = nchunk * NSPLIT;
  const int xcd = blockIdx.x & 7;
  const int per_xcd = a.tiles_per_xcd * inner;   // work items of one batch element on this XCD (tiles past a.tiles are empty slots)
  const int n_seq = per_xcd * a.B, seq_step = gridDim.x >> 3;
  const int h = a.h, w = a.w, hw = h * w, D = a.D;
  const int nv = NV > 0 ? NV : a.nv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tp = tid % kThreads, pg = tid / kThreads;   // pixel of the tile, plane group
  const size_t view_floats = (size_t)hw * C;
  const int view_bytes = uniform_int((int)(view_floats * 4));
  struct Item {
    int b, tile, d0, c0;
    bool live;
  };
  auto decode = [&](int sq) {
    Item it;
    it.live = sq < n_seq;
    const int sqc = it.live ? sq : 0;
    it.b = sqc / per_xcd;
    const int sl = sqc - it.b * per_xcd;
    const int tile_local = sl / inner, rem = sl - tile_local * inner;
    it.tile = xcd * a.tiles_per_xcd + tile_local;
    it.live = it.live && it.tile < a.tiles;
    it.d0 = (rem / NSPLIT) * DC;
    it.c0 = (rem % NSPLIT) * CS;
    return it;
  };
  // the chunk's first / last hypothesis of this thread's pixel in item `it` (zeros through an empty descriptor for an empty slot)
  auto load_end_depths = [&](const Item &it, float &first, float &last) {
    const int ty_ = it.tile / a.tiles_x, tx_ = it.tile - ty_ * a.tiles_x;
    const int px_ = tx_ * TW + tp % TW, py_ = ty_ * TH + tp / TW;
    const int pcl_ = (px_ < w && py_ < h) ? py_ * w + px_ : 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(uniform_ptr(a.depth + (size_t)(it.live ? it.b : 0) * D * hw)), 0,
                                                                        uniform_int(it.live ? D * hw * 4 : 0), 0x00020000);
    first = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, pcl_ * 4, uniform_int(it.d0 * hw * 4), 0));
    last = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, pcl_ * 4, uniform_int((it.d0 + DC - 1) * hw * 4), 0));
  };
  int seq = blockIdx.x >> 3;
  Item cur = decode(seq);
  float dv_first_pf, dv_last_pf;
  load_end_depths(cur, dv_first_pf, dv_last_pf);
  for (; seq < n_seq; seq += seq_step) {
  const Item nxt = decode(seq + seq_step);
  const float dv_first = dv_first_pf, dv_last = dv_last_pf;
  if (!cur.live) {   // an empty slot (workgroup-uniform): nothing to do, the next item's hypotheses still have to be requested
    load_end_depths(nxt, dv_first_pf, dv_last_pf);
    cur = nxt;
    continue;
  }
  const int tile = cur.tile, d0 = cur.d0, c0 = cur.c0, b = cur.b;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int px = tx * TW + tp % TW, py = ty * TH + tp / TW;
  const bool valid = px < w && py < h;
  const int pcl = valid ? py * w + px : 0;
  const float xf = (float)px, yf = (float)py;
  const float *fb = uniform_ptr(a.feats + (size_t)b * a.Vtot * view_floats);
  const float *pb = uniform_ptr(a.proj + ((size_t)b * a.pstride + a.pv0) * 12);

  // depth hypotheses of this pixel: the chunk's first and last plane were requested one item ahead (boxes), the others one plane ahead
  // inside the plane loop (a load issued BEFORE a plane's volume stores only waits for the stores of the plane
  // before: gfx9's vmcnt is in order - and it costs one register instead of DC)
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(uniform_ptr(a.depth + (size_t)b * D * hw)), 0, uniform_int(D * hw * 4), 0x00020000);

  // ---- every global load the prologue and the plane loop need is issued HERE, together: the matrices, the reference
  // features, the chunk's depths.  (Round 2 issued them where they were used: three exposed memory latencies -
  // matrices per view inside the extents loop, `ref` after the staging, the depths before the plane loop - a third
  // of a workgroup's life, tools/gpu_cv_trace.py.)
  constexpr int NVS = NV > 0 ? NV : 1;
  // NV > 0: lane l of every wave loads element l of the staged views' matrices (<= 24 floats; out-of-range lanes read
  // 0 from the buffer), v_readlane then puts them in SGPRs.  Run-time view count: through LDS (the compiler cannot
  // prove that `proj` is not written by the volume stores and would re-load it with VECTOR loads inside the plane
  // loop, where a wait for a vector load also waits for every older store: gfx9 has one in-order vmcnt).
  float pm_lane = 0.0f;
  if (NV > 0) {
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pb), 0, uniform_int(NVS * 12 * 4), 0x00020000);
    pm_lane = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, lane * 4, 0, 0));
  } else if (tid < nv * 12) {
    pmat[tid] = pb[tid];
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
  // CS >= 16 runs at two waves per SIMD whatever it does (LDS): registers are free there and the planes' depths
  // are all loaded up front, which keeps every vector-memory wait out of the plane loop.
#ifdef CASMVS_DV_REGS_ALL
  constexpr bool DV_REGS = true;
#else
  constexpr bool DV_REGS = CS >= 16;
#endif
  constexpr int KP = DC / PG;          // planes of a plane group
  const int k0 = uniform_int(pg * KP);
  float dvr[DV_REGS ? KP : 1];
  if (DV_REGS) {
#pragma unroll
    for (int i = 0; i < KP; ++i)
      dvr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, pcl * 4, uniform_int((d0 + k0 + i) * hw * 4), 0));
  } else {
    dvr[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, pcl * 4, uniform_int((d0 + k0) * hw * 4), 0));
  }
  float Pm[NVS][12];
  if (NV > 0) {
#pragma unroll
    for (int vi = 0; vi < NVS; ++vi)
#pragma unroll
      for (int i = 0; i < 12; ++i)
        Pm[vi][i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pm_lane), vi * 12 + i));
  }

  // ---- 1. boxes (by the waves of plane group 0) ---------------------------------------------------------------
  auto extents = [&](const float *P, int vi) {
    int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      // footprints that touch the image, in PADDED coordinates (x0 in [-1, W - 1], y0 in [-1, H - 1]; NaN / inf fail the compares)
      const SweepCoords c = plane_sweep_coords(P, xf, yf, e == 0 ? dv_first : dv_last, w, h);
      if (valid && c.x0 >= -1.0f && c.x0 <= (float)(w - 1) && c.y0 >= -1.0f && c.y0 <= (float)(h - 1)) {
        const int x0 = (int)c.x0, y0 = (int)c.y0;
        xmn = min(xmn, x0); xmx = max(xmx, x0 + 1);
        ymn = min(ymn, y0); ymx = max(ymx, y0 + 1);
      }
    }
    xmn = wave_min(xmn); xmx = wave_max(xmx); ymn = wave_min(ymn); ymx = wave_max(ymx);
    if (lane == 0) {
      int *r = red + (wave * kMaxViews + vi) * 4;
      r[0] = xmn; r[1] = xmx; r[2] = ymn; r[3] = ymx;
    }
  };
  if (pg == 0) {   // wave-uniform
    if (NV > 0) {
#pragma unroll
      for (int vi = 0; vi < NVS; ++vi) extents(Pm[vi], vi);
    } else {
#pragma unroll 1
      for (int vi = 0; vi < nv; ++vi) extents(pb + vi * 12, vi);
    }
  }
  CV_STAMP();   // 1: this wave's extents done (depth loads landed, taps at the chunk's first / last plane, wave reductions)
  __syncthreads();
  CV_STAMP();   // 2: all waves there
  if (tid < nv) {
    int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
      const int *r = red + (wv * kMaxViews + tid) * 4;
      xmn = min(xmn, r[0]); xmx = max(xmx, r[1]); ymn = min(ymn, r[2]); ymx = max(ymx, r[3]);
    }
    if (xmn > xmx) { xmn = 0; xmx = 1; ymn = 0; ymx = 1; }   // nothing of this tile projects into the view
    if (NCHW) { xmn &= ~3; xmx |= 3; }   // channel-plane staging moves 4 consecutive x per load: whole quads (w % 4 == 0: a quad is image or padding)
    // every box has >= 2 columns and >= 2 rows (a footprint spans two of each; lanes without a staged footprint read the origin's 2 x 2 pixels with
    // zero weights): the clipped width leaves room for two rows
    int bw = xmx - xmn + 1, bh = ymx - ymn + 1;
    int maxbw = a.cap_units / (2 * L::units_per_px_bound()) - 1;   // host guarantees >= 4
    if (NCHW) maxbw &= ~3;
    if (bw > maxbw) bw = maxbw;
    const int maxbh = a.cap_units / L::row_units(bw);          // >= 2
    if (bh > maxbh) bh = maxbh;
    const int nsu = bw * GL, q256 = NT / nsu;
    int *p = prm + tid * 8;
    p[0] = xmn; p[1] = ymn; p[2] = bw; p[3] = bh; p[4] = q256; p[5] = NT - q256 * nsu;
  }
  __syncthreads();

  CV_STAMP();   // 3: boxes published
  // ---- 2. staging -----------------------------------------------------------------------------------------------
  // One pass = NUB 16-byte loads per thread and view.  NV > 0: the first pass of EVERY view is issued before anything
  // is waited for (one memory latency for all the boxes; a box of the expected size is one pass).
  struct ViewStage {
    int nsu, total, rowu, base;
    int q256;
    int bx0, yrow;               // the box's first column / the image row of this thread's next unit (padded coordinates: either may be outside the image)
    int ru, lofs, gofs;          // this thread's next unit: index inside its box row, LDS unit of the row, global byte offset of the row
    int r256, dL, dG, gW;        // per step of NT units: ru += r256, the row advances by q256 (+ 1 when ru wraps)
    __amdgpu_buffer_rsrc_t src;
    f32x4 *bx;
    f32x4 regs[NUB];
    int lo[NUB];
  };
  auto stage_init = [&](ViewStage &s, int vi) {
    const int *p = prm + vi * 8;
    const int bx0 = uniform_int(p[0]), by0 = uniform_int(p[1]), bw = uniform_int(p[2]), bh = uniform_int(p[3]);
    const int q256 = uniform_int(p[4]);
    s.r256 = uniform_int(p[5]);
    s.nsu = bw * GL; s.total = bh * s.nsu; s.rowu = L::row_units(bw);
    s.src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(uniform_ptr(fb + (size_t)(a.v0 + vi) * view_floats)), 0, view_bytes, 0x00020000);
    s.bx = box + (size_t)vi * a.cap_units;
    const int row = tid / s.nsu;
    s.ru = tid - row * s.nsu; s.base = 0;
    s.bx0 = bx0; s.yrow = by0 + row; s.q256 = q256;
    s.gW = w * C * 4;
    s.lofs = row * s.rowu; s.gofs = ((by0 + row) * w + bx0) * (C * 4) + c0 * 4;
    s.dL = q256 * s.rowu; s.dG = q256 * s.gW;
  };
  // (all running offsets: no multiplication, no division per load - the round-2 form spent 28 instructions per load)
  auto stage_issue = [&](ViewStage &s) {
#pragma unroll
    for (int i = 0; i < NUB; ++i) {
      const bool ok = s.base + tid + NT * i < s.total;
      const int pxx = (int)((unsigned)s.ru / (unsigned)GL), ch = (int)((unsigned)s.ru % (unsigned)GL);
      s.lo[i] = ok ? s.lofs + L::unit(pxx) + ch : -1;
      // a lane past the end of the box, or at a box position outside the image (the zero padding), addresses beyond the buffer: the hardware
      // returns 0 without a memory access
      const bool in_img = ((unsigned)(s.bx0 + pxx) < (unsigned)w) & ((unsigned)s.yrow < (unsigned)h);
      s.regs[i] = buf_load4(s.src, (ok & in_img) ? s.gofs + (pxx * C + 4 * ch) * 4 : -16, 0);
      s.ru += s.r256; s.lofs += s.dL; s.gofs += s.dG; s.yrow += s.q256;
      const bool wrap = s.ru >= s.nsu;
      s.ru -= wrap ? s.nsu : 0; s.lofs += wrap ? s.rowu : 0; s.gofs += wrap ? s.gW : 0; s.yrow += wrap ? 1 : 0;
    }
    s.base += NT * NUB;
  };
  auto stage_commit = [&](ViewStage &s) {
#pragma unroll
    for (int i = 0; i < NUB; ++i)
      if (s.lo[i] >= 0) s.bx[s.lo[i]] = s.regs[i];
  };
  if constexpr (NCHW) {
    // One view (homo_warp).  Item = (box row, quad of 4 x, group of 4 channels): four 16-byte loads - the same 4 x of the group's 4 channel planes -
    // transposed in registers into four pixel-major 16-byte LDS units.  Two items (8 loads) in flight per thread.
    const int *p = prm;
    const int bx0 = uniform_int(p[0]), by0 = uniform_int(p[1]), bw = uniform_int(p[2]), bh = uniform_int(p[3]);
    const int rowu = L::row_units(bw), qpr = bw >> 2, ipr = qpr * GL, items = bh * ipr;   // quads / items per row
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(uniform_ptr(fb + (size_t)a.v0 * view_floats)), 0, view_bytes, 0x00020000);
    constexpr int NI = 2;
#pragma unroll 1
    for (int base = 0; base < items; base += NT * NI) {
      f32x4 v[NI][4];
      int lo[NI];
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int it = base + tid + NT * k;
        const bool ok = it < items;
        const int r = (int)((unsigned)it / (unsigned)ipr), rem = it - r * ipr;
        const int q = (int)((unsigned)rem / (unsigned)GL), cg = rem - q * GL;
        lo[k] = ok ? r * rowu + cg : -1;
        const int px0 = 4 * q;
        lo[k] = ok ? lo[k] + (px0 << 16) : -1;   // (LDS row / channel part, first pixel) packed: unit(px) is not linear in px for CS = 8
        // whole quads of x (bx0 and w are multiples of 4): a quad is inside the image or it is padding
        const bool in_img = ((unsigned)(bx0 + px0) < (unsigned)w) & ((unsigned)(by0 + r) < (unsigned)h);
        const int gofs = (ok & in_img) ? (((c0 + 4 * cg) * h + by0 + r) * w + bx0 + px0) * 4 : -16;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[k][i] = buf_load4(src, gofs >= 0 ? gofs + i * hw * 4 : -16, 0);
      }
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        if (lo[k] >= 0) {
          const int px0 = lo[k] >> 16, rc = lo[k] & 0xffff;
#pragma unroll
          for (int i = 0; i < 4; ++i) box[rc + L::unit(px0 + i)] = f32x4{v[k][0][i], v[k][1][i], v[k][2][i], v[k][3][i]};
        }
      }
    }
  } else if (NV > 0) {
    ViewStage st[NVS];
#pragma unroll
    for (int vi = 0; vi < NVS; ++vi) { stage_init(st[vi], vi); stage_issue(st[vi]); }
#pragma unroll
    for (int vi = 0; vi < NVS; ++vi) {
      stage_commit(st[vi]);
      while (st[vi].base < st[vi].total) { stage_issue(st[vi]); stage_commit(st[vi]); }   // a box larger than one pass
    }
  } else {
#pragma unroll 1
    for (int vi = 0; vi < nv; ++vi) {
      ViewStage s1;
      stage_init(s1, vi);
      while (s1.base < s1.total) { stage_issue(s1); stage_commit(s1); }
    }
  }
  CV_STAMP();   // 4: boxes staged (this wave's loads landed and were written to LDS)
  __syncthreads();

  CV_STAMP();   // 5: all waves staged
  // wave-uniform boxes of the unrolled form
  Box boxes[NVS];
  if (NV > 0) {
#pragma unroll
    for (int vi = 0; vi < NVS; ++vi) boxes[vi] = read_box(prm, vi);
  }

  const size_t vol_floats = (size_t)C * D * hw;   // one batch element of out (VAR / WARP / VAR_PART)
  const PlaneStore ps = make_plane_store<TW>(a.out + (size_t)b * vol_floats, vol_floats * 4, c0, D, hw, h, w, wave & 3, lane, tx, ty);
  PlaneStore ps2 = ps;
  if (MODE == MODE_VAR_PART)
    ps2.rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.out2 + (size_t)b * vol_floats), 0, uniform_int((int)(vol_floats * 4)), 0x00020000);
  const int direct_voff = valid ? (c0 * D * hw + pcl) * 4 : -16;

#ifdef CASMVS_TRACE
  const int abl = a.ablate;
#else
  constexpr int abl = 0;
#endif

  // ---- 3. plane by plane (a rolled loop: the body is ~1.5 KB of code per view, the 8 planes unrolled were 40 KB)
  float *tr = tr_all + wave * (4 * RS);
  const float rV = 1.0f / (float)a.nviews_total;
  float dvk = dvr[0];
#ifndef CASMVS_NO_PRELOOP_WAIT
  // Every load issued so far has landed after this: the compiler's wait-count pass then knows that nothing the plane
  // loop reads is pending and places no vmcnt wait inside it.  Without it (ISA, round 2) it waited vmcnt(5) ... vmcnt(0)
  // for the depth registers in EVERY iteration - and vmcnt(0) also waits for the previous plane's volume stores.
  __builtin_amdgcn_s_waitcnt(0x0f70);
#endif
  load_end_depths(nxt, dv_first_pf, dv_last_pf);   // the next item's box extents will not wait for memory (consumed behind this item's plane loop)
  CV_STAMP();   // 6: plane loop begins
#pragma unroll 1
  for (int k = k0; k < k0 + KP; ++k) {
    float dv_next = dv_last;
    if (DV_REGS) {
      // the chunk's depths rotate through the register array (KP - 1 moves; a select chain on the scalar plane
      // counter cost three instructions per plane of the chunk, dynamic register indexing is not available)
      dv_next = KP > 1 ? dvr[1] : dvr[0];
#pragma unroll
      for (int i = 1; i + 1 < KP; ++i) dvr[i] = dvr[i + 1];
    } else if (!(abl & 4)) {
      const int kn = k + 1 < DC ? k + 1 : DC - 1;
      dv_next = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(drs, pcl * 4, uniform_int((d0 + kn) * hw * 4), 0));
    }
    f32x2 s2[CS / 2], q2[CS / 2];
#pragma unroll
    for (int c = 0; c < CS / 2; ++c) {
      if (MODE == MODE_VAR || (MODE == MODE_VAR_PART && a.with_ref)) {
        s2[c] = f32x2{ref[2 * c], ref[2 * c + 1]};   // volume_sum = ref_volume            (mvsnet.py:140)
        q2[c] = s2[c] * s2[c];                       // volume_sq_sum = ref_volume ** 2    (mvsnet.py:141)
      } else {
        s2[c] = f32x2{0.0f, 0.0f};                   // volume_sum = 0                     (mvsnet.py:144)
        q2[c] = f32x2{0.0f, 0.0f};
      }
    }
    if (abl & 8) {   // ablation: the plane loop without taps, LDS reads and accumulation - what the volume stores alone cost
    } else if (NV > 0) {
#pragma unroll
      for (int vi = 0; vi < NVS; ++vi)
        accumulate_view<C, CS, SQ, kJB, NCHW>(Pm[vi], xf, yf, dvk, w, h, valid, boxes[vi], box + (size_t)vi * a.cap_units,
                                         fb + (size_t)(a.v0 + vi) * view_floats, view_bytes, c0, s2, q2, abl);
    } else {
#pragma unroll 1
      for (int vi = 0; vi < nv; ++vi) {
        const Box bxv = read_box(prm, vi);
        float Pv[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) Pv[i] = pmat[vi * 12 + i];
        accumulate_view<C, CS, SQ, kJB>(Pv, xf, yf, dvk, w, h, valid, bxv, box + (size_t)vi * a.cap_units,
                                   uniform_ptr(fb + (size_t)(a.v0 + vi) * view_floats), view_bytes, c0, s2, q2, abl);
      }
    }
    // ---- 4. the plane leaves -------------------------------------------------------------------------------------
    const int d = d0 + k;
    float s[CS], q[CS];
    if (MODE == MODE_VAR) {
      const f32x2 rV2{rV, rV};
#pragma unroll
      for (int c = 0; c < CS / 2; ++c) {
        const f32x2 m = s2[c] * rV2;   // sq/V - (sum/V)^2 (mvsnet.py:167), x/V as x * (1/V)
        s2[c] = q2[c] * rV2 - m * m;
      }
    }
#pragma unroll
    for (int c = 0; c < CS / 2; ++c) {
      s[2 * c] = s2[c].x; s[2 * c + 1] = s2[c].y;
      q[2 * c] = q2[c].x; q[2 * c + 1] = q2[c].y;
    }
    if (MODE == MODE_VAR || is_warp(MODE) || MODE == MODE_VAR_PART) {
      if (!(abl & 1)) store_plane<CS>(s, tr, ps, lane, D, hw, d, w, direct_voff);
      if (MODE == MODE_VAR_PART) store_plane<CS>(q, tr, ps2, lane, D, hw, d, w, direct_voff);
    } else {
      // group-wise correlation (mvsnet.py:170-171): mean over the C/G channels of a group of
      // volume_sum * ref_volume, then / (V - 1); lane = pixel stores TW consecutive x per group
      const int cpg = C / a.G;
      const float fn = (float)cpg, fv = (float)a.nviews_total;
      const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
          uniform_ptr(a.out + (size_t)b * a.G * D * hw), 0, uniform_int(a.G * D * hw * 4), 0x00020000);
      const int gvoff = valid ? pcl * 4 : -16;
      int gsoff = ((c0 / cpg) * D + d) * hw * 4;
      float acc = 0.0f;
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < CS; ++c) {   // static register indexing; group boundaries are runtime
        acc = acc + s[c] * ref[c];
        if (++cnt == cpg) {
          const float m = acc / fn;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, MODE == MODE_GWC ? m / fv : m), gr, gvoff,
                                                uniform_int(gsoff), 0);
          gsoff += D * hw * 4;
          acc = 0.0f;
          cnt = 0;
        }
      }
    }
    dvk = dv_next;
    CV_STAMP();   // 7 + k: plane k issued (taps, LDS reads, accumulation, stores issued)
  }
#ifdef CASMVS_TRACE
  __builtin_amdgcn_s_waitcnt(0x0f70);
  CV_STAMP();   // last: this wave's stores acknowledged
#endif
  cur = nxt;
  __syncthreads();   // every wave is out of the plane loop: the boxes and the box table may be rewritten
  }   // items
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

bool g_persistent = true;   // (internal linkage; false = one workgroup per item as before round 5: the A/B switch of the trace build, CASMVS_CV_PERSIST=0)
int compute_units() {
#ifdef HIPEMU_LDS_BYTES
  return 8;   // (tests/hipemu: one CU per XCD, so that its small volumes give every workgroup several items)
#else
  static const int n = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8) cu = 256;
    return cu;
  }();
  return n;
#endif
}

struct Plan {
  int cs, tw, dc, pg, cap_units, lds_bytes;
};

// LDS plan: the highest occupancy (3, 2, 1 workgroups per CU) whose per-view capacity still holds the box a
// tile is expected to need ((TW + DC + 4) x (TH + 2) pixels: ~0.6 px of epipolar slide per plane + slack).
// dc: planes per workgroup, 8 or 16 (16: half the prologues - box extents, staging, two memory round trips, 35 % of a workgroup's life at 8 planes,
// tools/gpu_cv_trace.py - on a box ~8 px wider; the kernels exist for the CS = 16 forms with a compile-time view count)
bool make_plan(int C, int w, int D, int nv, int G, int mode, Plan &p, int dc = 8) {
  if (nv < 1 || nv > kMaxViews) return false;
  p.dc = dc;
  if (D % p.dc != 0) return false;
  if (w % 4 != 0) return false;                 // 16-byte volume stores
  p.tw = (C == 8 && w % 64 == 0) ? 64 : 32;   // A/B (tools/gpu_cv_ab.sh): 32 x 8 tiles win at C = 16, no difference at C = 8
  p.cs = C >= 16 ? 16 : 8;
  const bool gwc = mode == MODE_GWC || mode == MODE_GWC_PART;
  while (gwc && (C / G) > p.cs) p.cs *= 2;     // a group must not straddle two channel splits
  p.pg = 1;   // two plane groups (8 waves on the same boxes) need <= 128 VGPRs: the compiler spills in the plane loop, 3x slower
#ifdef CASMVS_TRACE   // profiling build only: A/B of the tile shape / channel split
  if (const char *e = getenv("CASMVS_CV_PG")) p.pg = atoi(e);
  if (const char *e = getenv("CASMVS_CV_TW")) p.tw = atoi(e);
  if (const char *e = getenv("CASMVS_CV_CS")) p.cs = atoi(e);
#endif
  if (C % p.cs != 0 || (p.cs != 8 && p.cs != 16 && p.cs != 32)) return false;
  const int th = kThreads / p.tw;
  const int units_per_8px = p.cs == 8 ? 17 : (p.cs == 16 ? 34 : 8 * (p.cs / 4 + 1));   // BoxLayout
  const int need = ((p.tw + p.dc + 4 + (mode == MODE_WARP_NCHW ? 6 : 0)) * units_per_8px / 8 + 1) * (th + 2);   // NCHW staging: boxes of whole x quads
  // workgroups per CU the registers allow (waves_per_simd of the kernel that will run): a smaller LDS budget than
  // that buys nothing and only clips boxes earlier
  const int occ = waves_per_simd(p.cs, mode, p.pg) / p.pg;
  static const int budgets[4] = {39 * 1024, 52 * 1024, 79 * 1024, 158 * 1024};
  for (int i = occ >= 4 ? 0 : (occ == 3 ? 1 : 2); i < 4; ++i) {
    const int cap = (budgets[i] - fixed_lds(p.pg)) / (nv * 16);
    if (cap >= need) {
      p.cap_units = cap;
      p.lds_bytes = fixed_lds(p.pg) + nv * cap * 16;
      return true;
    }
  }
  return false;
}

template <int C, int CS, int MODE, int TW, int NV, int PG, int DC>
int launch_dc(const SweepArgs &a, const Plan &p, int B, hipStream_t st) {
  auto kernel = costvol_lds_kernel<C, CS, MODE, TW, DC, NV, PG>;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), 158 * 1024, "costvol_lds_kernel")) return rc;
  const int inner = (a.D / DC) * (C / CS);
  // persistent launch: per XCD as many workgroups as its CUs keep resident (registers: waves_per_simd; LDS: the plan's budget follows the same number),
  // or one per work item when there are fewer
  const long items_per_xcd = (long)a.tiles_per_xcd * inner * B;
  const long resident_per_xcd = (long)(compute_units() / 8) * (waves_per_simd(CS, MODE, PG) / PG);
  const long wg_per_xcd = g_persistent ? std::min(items_per_xcd, resident_per_xcd) : items_per_xcd;
  CASMVS_REQUIRE(8 * wg_per_xcd <= 0x7fffffffL, "costvol_lds: too many workgroups");
  dim3 grid((unsigned)(8 * wg_per_xcd), 1u);
  (void)B;
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads * PG), (size_t)p.lds_bytes, st, a);
  return casmvs::check_launch("costvol_lds_kernel");
}

long g_dc16_min_rounds = 4;   // (internal linkage; the CPU emulation's driver, which includes this file, sets 0 to run the 16-plane form on its small volumes)
constexpr bool has_dc16(int cs, int mode, int nv, int pg) { return cs == 16 && nv > 0 && pg == 1 && (mode == MODE_VAR || mode == MODE_GWC || is_warp(mode)); }

template <int C, int CS, int MODE, int TW, int NV, int PG>
int launch_pg(const SweepArgs &a, const Plan &p, int B, hipStream_t st) {
  if constexpr (has_dc16(CS, MODE, NV, PG)) {
    if (p.dc == 16) return launch_dc<C, CS, MODE, TW, NV, PG, 16>(a, p, B, st);
  }
  if (p.dc != 8) return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "costvol_lds: no kernel sweeps %d planes per workgroup in this form", p.dc);
  return launch_dc<C, CS, MODE, TW, NV, PG, 8>(a, p, B, st);
}

template <int C, int CS, int MODE, int TW, int NV>
int launch_nv(const SweepArgs &a, const Plan &p, int B, hipStream_t st) {
#ifdef CASMVS_TRACE   // the two-plane-group form is kept for A/B runs only (see make_plan)
  if constexpr (NV == 2 || NV == 1) {
    if (p.pg == 2) return launch_pg<C, CS, MODE, TW, NV, 2>(a, p, B, st);
  }
#endif
  return launch_pg<C, CS, MODE, TW, NV, 1>(a, p, B, st);
}

// Which view counts get an unrolled kernel: the un-fused warp has one view; the fused builders the headline
// V = 3 (2 source views); everything else (other V, the partial sums of the view-sharded build) the rolled form.
template <int C, int CS, int MODE, int TW>
int launch_cfg(const SweepArgs &a, const Plan &p, int B, hipStream_t st) {
  if constexpr (is_warp(MODE)) return launch_nv<C, CS, MODE, TW, 1>(a, p, B, st);
  if constexpr (MODE == MODE_VAR || MODE == MODE_GWC) {
    if (a.nv == 2) return launch_nv<C, CS, MODE, TW, 2>(a, p, B, st);
  }
#ifdef CASMVS_CV_UNROLL46   // A/B build (round 6): the variance volume of V = 5 / V = 7 with a compile-time view count (profiles/r06_costvol_v5_v7_unrolled.txt)
  if constexpr (MODE == MODE_VAR) {
    if (a.nv == 4) return launch_nv<C, CS, MODE, TW, 4>(a, p, B, st);
    if (a.nv == 6) return launch_nv<C, CS, MODE, TW, 6>(a, p, B, st);
  }
#endif
  return launch_nv<C, CS, MODE, TW, 0>(a, p, B, st);
}

template <int C, int MODE>
int launch_c(const SweepArgs &a, const Plan &p, int B, hipStream_t st) {
#define CASMVS_TRY(CSV, TWV) \
  if (p.cs == CSV && p.tw == TWV) return launch_cfg<C, CSV, MODE, TWV>(a, p, B, st);
  if constexpr (C == 32) { CASMVS_TRY(32, 64) CASMVS_TRY(32, 32) }
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
  {
    // 16 planes per workgroup where that form exists (has_dc16: the compile-time view counts - V = 3 fused, the un-fused warp) and the launch still
    // has >= 4 rounds of the chip's 512 resident workgroups (2 per CU): fewer, longer workgroups otherwise lose to the tail of the last round
    const int nv_static = is_warp(MODE) ? 1 : ((MODE == MODE_VAR || MODE == MODE_GWC) && a.nv == 2 ? 2 : 0);
    Plan p16;
    long min_rounds = g_dc16_min_rounds;
#ifdef CASMVS_TRACE
    if (const char *e = getenv("CASMVS_CV_DC16_ROUNDS")) min_rounds = atol(e);   // A/B: 0 = always, 1000000 = never
#endif
    if (p.cs == 16 && has_dc16(p.cs, MODE, nv_static, p.pg) && a.D % 16 == 0 && make_plan(C, a.w, a.D, a.nv, a.G, MODE, p16, 16) && p16.cs == p.cs && p16.tw == p.tw &&
        p16.lds_bytes <= 79 * 1024 && (long)a.tiles * (a.D / 16) * (C / p.cs) * B >= min_rounds * 512)
      p = p16;
  }
  a.tiles_per_xcd = casmvs::ceil_div(a.tiles, 8);
  a.B = B;
#ifdef CASMVS_TRACE
  if (const char *e = getenv("CASMVS_CV_PERSIST")) g_persistent = atoi(e) != 0;
#endif
  a.cap_units = p.cap_units;
  a.ablate = 0;
#ifdef CASMVS_TRACE
  if (const char *e = getenv("CASMVS_CV_ABLATE")) a.ablate = atoi(e);
#endif
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
  CASMVS_REQUIRE((size_t)h * w * C * D < ((size_t)1 << 29), "%s: one batch element's volume must hold < 2^29 floats", what);
  return CASMVS_OK;
}

}  // namespace

#ifdef CASMVS_TRACE
extern "C" int casmvs_cv_trace_read(unsigned long long *host, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cv_trace), sizeof(unsigned long long) * 64 * 32) != hipSuccess) return -3;
  if (clear) {
    static unsigned long long z[64 * 32];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_cv_trace), z, sizeof(z)) != hipSuccess) return -3;
  }
  return 0;
}
#endif

extern "C" int casmvs_costvol_lds_supported(int C, int w, int D, int n_src_views, int G) {
  Plan p;
  if (C != 8 && C != 16 && C != 32) return 0;
  if (G > 1 && C % G != 0) return 0;
  return make_plan(C, w, D, n_src_views, G > 1 ? G : 1, G > 1 ? MODE_GWC : MODE_VAR, p) ? 1 : 0;
}

// Measured on the MI355X (tools/gpu_cv_variants.sh, profiles/r02_s3_costvol_ab.txt): with two source views (V = 3) the
// LDS-staged variance build beats the gather kernel at every level shape (C = 32 / 16 / 8: 91 / 135 / 84 us against
// 108 / 227 / 95 at batch 2).  Round 4 (tools/gpu_costvol_probe.py with dirtied caches, profiles/r04_costvol_ab_v5_v7_gwc.txt):
//   * more source views: the gather kernel stays ahead at every level of the V = 5 (1152 x 864) and V = 7 (768 x 576) workloads - all views
//     resident leaves one workgroup per CU (524 vs 632 us at level 1, V = 5; V = 7 does not fit at C >= 16), and the views taken two at a time
//     through the partial-sum kernels (two resident boxes, the occupancy of V = 3: a LOWER bound for a kernel that streams view pairs through
//     LDS, whose tap / interpolation / accumulation work per (voxel, view) is the same) take 568 us there, 1.1-2.5x the gather kernel elsewhere;
//   * group-wise correlation, V = 3: the LDS kernel now wins or ties at every level (G = 8, batch 4, noise-like depth: 201 / 297 / 218 us
//     against 234 / 340 / 245) - the round-3 prologue work moved it past the gather kernel, which it trailed in round 2.
extern "C" int casmvs_costvol_lds_preferred(int C, int w, int D, int n_src_views, int G) {
  return (n_src_views <= 2 && casmvs_costvol_lds_supported(C, w, D, n_src_views, G)) ? 1 : 0;
}

extern "C" int casmvs_costvol_var_lds_f32(const float *feats, const float *proj, const float *depth, float *out, int B,
                                          int V, int C, int h, int w, int D, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("costvol_var_lds", feats, proj, depth, out, B, V, C, h, w, D)) return rc;
  CASMVS_REQUIRE(V >= 2, "costvol_var_lds: V=%d", V);
  SweepArgs a{feats, proj, depth, out, nullptr, V, 1, V - 1, 0, V - 1, 1, V, 1, h, w, D, 0, 0, 0, 0, 0};
  return launch_mode<MODE_VAR>(a, C, B, (hipStream_t)stream, "costvol_var_lds");
}

extern "C" int casmvs_costvol_gwc_lds_f32(const float *feats, const float *proj, const float *depth, float *out, int B,
                                          int V, int C, int G, int h, int w, int D, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("costvol_gwc_lds", feats, proj, depth, out, B, V, C, h, w, D)) return rc;
  CASMVS_REQUIRE(V >= 2 && G >= 1 && C % G == 0, "costvol_gwc_lds: V=%d C=%d G=%d", V, C, G);
  SweepArgs a{feats, proj, depth, out, nullptr, V, 1, V - 1, 0, V - 1, 0, V - 1, G, h, w, D, 0, 0, 0, 0, 0};
  return launch_mode<MODE_GWC>(a, C, B, (hipStream_t)stream, "costvol_gwc_lds");
}

extern "C" int casmvs_homo_warp_nhwc_f32(const float *src, const float *proj, const float *depth, float *out, int B,
                                         int C, int H, int W, int D, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("homo_warp_nhwc", src, proj, depth, out, B, 1, C, H, W, D)) return rc;
  SweepArgs a{src, proj, depth, out, nullptr, 1, 0, 1, 0, 1, 0, 1, 1, H, W, D, 0, 0, 0, 0, 0};
  return launch_mode<MODE_WARP>(a, C, B, (hipStream_t)stream, "homo_warp_nhwc");
}

// homo_warp on the reference's own layout (modules.py:52-92: src_feat (B, C, H, W)): the source box is staged from the channel planes, no pixel-major copy
extern "C" int casmvs_homo_warp_lds_supported(int C, int W, int D) {
  Plan p;
  return (C == 8 || C == 16 || C == 32) && make_plan(C, W, D, 1, 1, MODE_WARP_NCHW, p) ? 1 : 0;
}

extern "C" int casmvs_homo_warp_lds_f32(const float *src, const float *proj, const float *depth, float *out, int B,
                                        int C, int H, int W, int D, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("homo_warp_lds", src, proj, depth, out, B, 1, C, H, W, D)) return rc;
  SweepArgs a{src, proj, depth, out, nullptr, 1, 0, 1, 0, 1, 0, 1, 1, H, W, D, 0, 0, 0, 0, 0};
  return launch_mode<MODE_WARP_NCHW>(a, C, B, (hipStream_t)stream, "homo_warp_lds");
}

extern "C" int casmvs_costvol_partial_var_f32(const float *feats, const float *proj, const float *depth, float *sum,
                                              float *sq, int B, int V, int C, int h, int w, int D, int view_begin,
                                              int view_end, int include_ref, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("costvol_partial_var", feats, proj, depth, sum, B, V, C, h, w, D)) return rc;
  CASMVS_REQUIRE(sq && (reinterpret_cast<size_t>(sq) & 15) == 0, "costvol_partial_var: sq must be a 16-byte aligned pointer");
  CASMVS_REQUIRE(1 <= view_begin && view_begin < view_end && view_end <= V, "costvol_partial_var: source views [%d, %d) of V=%d", view_begin, view_end, V);
  SweepArgs a{feats, proj, depth, sum, sq, V, view_begin, view_end - view_begin, view_begin - 1, V - 1, include_ref ? 1 : 0, V, 1, h, w, D, 0, 0, 0, 0, 0};
  return launch_mode<MODE_VAR_PART>(a, C, B, (hipStream_t)stream, "costvol_partial_var");
}

extern "C" int casmvs_costvol_partial_gwc_f32(const float *feats, const float *proj, const float *depth, float *out,
                                              int B, int V, int C, int G, int h, int w, int D, int view_begin,
                                              int view_end, void *stream) {
  casmvs::clear_error();
  if (int rc = check_common("costvol_partial_gwc", feats, proj, depth, out, B, V, C, h, w, D)) return rc;
  CASMVS_REQUIRE(G >= 1 && C % G == 0, "costvol_partial_gwc: C=%d G=%d", C, G);
  CASMVS_REQUIRE(1 <= view_begin && view_begin < view_end && view_end <= V, "costvol_partial_gwc: source views [%d, %d) of V=%d", view_begin, view_end, V);
  SweepArgs a{feats, proj, depth, out, nullptr, V, view_begin, view_end - view_begin, view_begin - 1, V - 1, 0, V - 1, G, h, w, D, 0, 0, 0, 0, 0};
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
