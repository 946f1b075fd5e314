// CostRegNet's `prob` head (Conv3d 8 -> 1, k3 p1, bias, no activation) walking the depth axis, optionally fused with
// the softmax / soft-argmin regression / confidence that consumes it.
//
// Reference semantics: models/mvsnet.py:89,104 (`prob`), :174-193 + models/modules.py:95-104 (softmax, depth
// regression, confidence).
//
// One output channel is not a matrix problem: the head is a VALU kernel.  The tile kernels it replaces
// (prob_valu_kernel / prob_pk_kernel in conv3d_mfma.hip: a thread = 4 x of one (z, y), the 6 x 2 inputs of every
// (channel pair, kz, ky) re-read from LDS) were LDS-bandwidth bound at 2.8x their HBM time: 432 B of tap reads per
// output voxel.  Here a workgroup owns a 64 x 8 pixel tile and WALKS z:
//   * input plane z_in (8 channels, halo tile) is staged ONCE into one of two LDS slots (channel pairs interleaved
//     [x][2], so that a thread's 4 x 2 inputs of a (pair, ky) are two aligned ds_read_b128 and every FMA is a
//     v_pk_fma_f32 over the pair with the weight pair as a scalar operand);
//   * the plane's contribution to the THREE output planes z_in + 1, z_in, z_in - 1 (kz = 0, 1, 2) is accumulated
//     from the same registers into three rotating accumulators: 18 packed FMAs per 2 LDS reads, 144 B of tap reads
//     per output voxel, and the plane after next is in flight from HBM meanwhile (one barrier per plane);
//   * the depth axis may be cut into chunks of `zc` output planes (grid.x = tiles * chunks) when the pixel tiles
//     alone do not fill the chip (levels 2 / 1); a chunk re-stages one halo plane on either side.
// FUSE (one chunk = the whole depth range): a thread has produced every cost value of its two pixels itself; after
// the walk it re-reads them (its own stores: same-thread program order, L2 hits), and runs the reference's softmax /
// regression / confidence on them (softmax_regress.h, the same code as softmax_regress_kernel): the separate
// regression launch and its read of the cost volume from memory disappear.
//
// Bound: VALU issue (108 v_pk_fma_f32 per output voxel); nominal roof: HBM (8 input channels + 1 output per voxel).
#include "buffer_ops.h"
#include "common.h"
#include "softmax_regress.h"

namespace {

using namespace casmvs::buf;
constexpr int kThreads = 256;

struct ProbZCfg {
  static constexpr int TX = 64, TY = 8;          // output pixels per workgroup; a thread = 2 consecutive x of one row
  static constexpr int IY = TY + 2;              // staged rows y0 - 1 .. y0 + TY
  static constexpr int NG = TX / 4 + 2;          // 16-byte global groups per staged row: x0 - 4 .. x0 + TX + 3
  static constexpr int RS = 2 * (TX + 2);        // floats per LDS row: positions x0 - 1 .. x0 + TX, [position][channel of the pair]
  static constexpr int SP = IY * RS;             // floats per channel pair of one plane
  static constexpr int NPAIR = 4;                // 8 input channels
  static constexpr int SLOT = NPAIR * SP;        // floats per plane slot (+ 4: a dump pair for the unused staged columns)
  static constexpr int ITEMS = NPAIR * IY * NG;  // (pair, row, group) staging items per plane
  static constexpr int NK = (ITEMS + kThreads - 1) / kThreads;
  static constexpr size_t LDS_BYTES = 2 * (size_t)(SLOT + 4) * sizeof(float);
  // lane (xi = tid & 31, yi = tid >> 5) reads floats [4 xi, 4 xi + 8) of row yi + ky: 32 lanes x 16 B = 512 B per row
  // of lanes, two rows per wave RS floats apart - conflict-free for the 4 x 16-lane groups of ds_read_b128
  // (checked for RS = 132 by enumeration, tests/kernel_model.py: prob_zwalk_bank_cycles)
  static_assert(RS % 4 == 0, "rows start 16-byte aligned");
};

// Contribution of the staged plane to the accumulators: A[2 - kz] += sum_{pair, ky, kx} in * w[kz][ky][kx] for the
// kz in KZM (bit mask).  `rows`: this lane's first row / position inside the slot; wpk: the P1 weight image
// ([pair][tap (27 + 5 zeros)][channel of the pair], conv3d_mfma.hip pack_weight) read as wave-uniform scalars.
template <int KZM>
__device__ __forceinline__ void zwalk_plane(const float *rows, const float *__restrict__ wpk, f32x2 (&A)[3][2]) {
  using Cfg = ProbZCfg;
  // the rows of step i + 1 = (pair, ky) are read from LDS before the FMAs of step i are issued
  f32x4v lo = *reinterpret_cast<const f32x4v *>(rows), hi = *reinterpret_cast<const f32x4v *>(rows + 4);
#pragma unroll
  for (int i = 0; i < Cfg::NPAIR * 3; ++i) {
    const int p = i / 3, ky = i % 3;
    const f32x2 P[4] = {f32x2{lo[0], lo[1]}, f32x2{lo[2], lo[3]}, f32x2{hi[0], hi[1]}, f32x2{hi[2], hi[3]}};
    if (i + 1 < Cfg::NPAIR * 3) {
      const float *row = rows + ((i + 1) / 3) * Cfg::SP + ((i + 1) % 3) * Cfg::RS;
      lo = *reinterpret_cast<const f32x4v *>(row);
      hi = *reinterpret_cast<const f32x4v *>(row + 4);
    }
    const float *wq = wpk + p * 64 + ky * 6;  // taps (kz, ky, kx = 0..2) x (even, odd channel) at [kz * 18 + 2 kx + c]
    // tap by tap over the (up to) six independent accumulators (kz, pixel): no two consecutive FMAs depend on each other
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        if (!((KZM >> kz) & 1)) continue;
        const f32x2 W{wq[kz * 18 + 2 * kx], wq[kz * 18 + 2 * kx + 1]};
        A[2 - kz][0] = __builtin_elementwise_fma(P[kx], W, A[2 - kz][0]);
        A[2 - kz][1] = __builtin_elementwise_fma(P[kx + 1], W, A[2 - kz][1]);
      }
    }
  }
}

// grid: x = tiles_x * tiles_y * chunks (XCD-major, chunk fastest, then x, then y), y = batch.
// in (B, 8, Di, Hi, Wi), Wi % 4 == 0, 16-byte aligned; cost (B, Di, Hi, Wi) is always written.
// FUSE: chunks == 1; dvals (B, Di, Hi, Wi) -> depth, conf (B, Hi, Wi) [, index].  DT: compile-time Di of the fused phase (0: generic).
template <int DT, bool FUSE>
__global__ __launch_bounds__(kThreads, 3) void prob_zwalk_kernel(
    const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ dvals, float *cost,
    float *__restrict__ depth, float *__restrict__ conf, int32_t *__restrict__ index, int Di, int Hi, int Wi,
    int tiles_x, int tiles_y, int zc, float slope) {
  using Cfg = ProbZCfg;
  constexpr int NK = Cfg::NK, RS = Cfg::RS, SP = Cfg::SP, SLOT = Cfg::SLOT, SLOTS = Cfg::SLOT + 4, NG = Cfg::NG, IY = Cfg::IY;
  extern __shared__ float smem[];
  const int ntile = tiles_x * tiles_y;
  const int nchunk = gridDim.x / ntile;
  const int bid = xcd_major(blockIdx.x, gridDim.x);
  const int chunk = bid % nchunk, tl = bid / nchunk;
  const int tx0 = (tl % tiles_x) * Cfg::TX, ty0 = (tl / tiles_x) * Cfg::TY;
  const int b = blockIdx.y;
  const int z_lo = chunk * zc, z_hi = min(z_lo + zc, Di);
  const int xi = threadIdx.x & 31, yi = threadIdx.x >> 5;
  const int HiWi = Hi * Wi, in_cs = Di * HiWi;
  const rsrc_t src = make_rsrc(in + (size_t)b * 8 * in_cs, (size_t)8 * in_cs * 4);
  const rsrc_t dst = make_rsrc(cost + (size_t)b * in_cs, (size_t)in_cs * 4);
  const float *tail = wpk + 8 * 32;  // scale[4] | shift[4] after the [pair][64] weight rows (cin = 8)
  const float sc0 = tail[0], sh0 = tail[4];

  // staging plan (tile constants): item e = tid + 256 k -> (pair, staged row, 16-byte group of the row)
  int voff0[NK], voff1[NK], loff[NK][4];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int e = threadIdx.x + k * kThreads;
    const int p = e / (IY * NG), r = e - p * (IY * NG);
    const int iy = r / NG, g = r - iy * NG;
    const int gy = ty0 - 1 + iy, gx = tx0 - 4 + 4 * g;
    const bool ok = e < Cfg::ITEMS && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;  // Wi % 4 == 0: a group is inside or outside
    voff0[k] = ok ? ((2 * p) * in_cs + gy * Wi + gx) * 4 : kOOB;
    voff1[k] = ok ? ((2 * p + 1) * in_cs + gy * Wi + gx) * 4 : kOOB;
    // column j of the group is position q = 4 g + j - 3 of the LDS row (position 0 = x0 - 1); positions 0 .. TX + 1 are
    // kept, the others (3 columns of the first group, 3 of the last, items beyond the plane) go to a dump word pair
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = 4 * g + j - 3;
      loff[k][j] = (e < Cfg::ITEMS && q >= 0 && q <= Cfg::TX + 1) ? p * SP + iy * RS + 2 * q : SLOT;
    }
  }
  f32x4v v0[NK], v1[NK];
  auto load_plane = [&](int z) {  // z in [0, Di)
    const int soff = z * HiWi * 4;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      v0[k] = buf_load4(src, voff0[k], soff);
      v1[k] = buf_load4(src, voff1[k], soff);
    }
  };
  auto store_plane = [&](float *slot) {
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x2 *>(slot + loff[k][j]) = f32x2{v0[k][j], v1[k][j]};
  };

  const int oy = ty0 + yi, ox = tx0 + 2 * xi;
  const int out_voff = (oy < Hi && ox < Wi) ? (oy * Wi + ox) * 4 : kOOB;  // Wi even: the pixel pair is inside or outside
  f32x2 A[3][2];  // [0]: output plane z_in - 1, [1]: z_in, [2]: z_in + 1; [pixel]; (even, odd channels' partial sums)
#pragma unroll
  for (int i = 0; i < 3; ++i) A[i][0] = A[i][1] = f32x2{0.f, 0.f};

  const int nplanes = z_hi - z_lo + 2;  // input planes z_lo - 1 .. z_hi
  int zin = z_lo - 1;
  if (zin >= 0) {
    load_plane(zin);
    store_plane(smem);
  }
  __syncthreads();
  for (int it = 0; it < nplanes; ++it, ++zin) {
    const float *cur = smem + (it & 1) * SLOTS;
    const bool nxt = it + 1 < nplanes && zin + 1 < Di;  // the next plane exists (zin + 1 >= 0 always)
    if (nxt) load_plane(zin + 1);
    if (zin >= 0 && zin < Di) {
      const float *rows = cur + yi * RS + 4 * xi;
      if (it == 0) zwalk_plane<1>(rows, wpk, A);                 // only output plane z_lo takes from plane z_lo - 1
      else if (it == nplanes - 1) zwalk_plane<4>(rows, wpk, A);  // only output plane z_hi - 1 takes from plane z_hi
      else zwalk_plane<7>(rows, wpk, A);
    }
    if (it >= 2) {  // output plane zin - 1 in [z_lo, z_hi) is complete
      float o0 = fmaf(A[0][0][0] + A[0][0][1], sc0, sh0), o1 = fmaf(A[0][1][0] + A[0][1][1], sc0, sh0);
      o0 = o0 > 0.0f ? o0 : o0 * slope;
      o1 = o1 > 0.0f ? o1 : o1 * slope;
      buf_store2(f32x2{o0, o1}, dst, out_voff, (zin - 1) * HiWi * 4);
    }
    A[0][0] = A[1][0];
    A[0][1] = A[1][1];
    A[1][0] = A[2][0];
    A[1][1] = A[2][1];
    A[2][0] = A[2][1] = f32x2{0.f, 0.f};
    if (nxt) store_plane(smem + ((it + 1) & 1) * SLOTS);
    __syncthreads();  // the other slot is published, this one is free
  }

  if constexpr (FUSE) {
    // every cost value of this thread's two pixels was stored by this thread: wait for the stores, then read them
    // back (the "memory" clobber also keeps the compiler from moving the loads above the buffer stores)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (oy < Hi && ox < Wi) {
      const size_t pix = (size_t)oy * Wi + ox;
      const float *cp = cost + (size_t)b * in_cs + pix, *dp = dvals + (size_t)b * in_cs + pix;
      const size_t o = (size_t)b * HiWi + pix;
#pragma nounroll
      for (int j = 0; j < 2; ++j) {  // one pixel after the other: DT values in registers at a time
        float d, c;
        int ix;
        casmvs::softmax_regress_pixel<DT>(cp + j, dp + j, (size_t)HiWi, Di, d, c, ix);
        depth[o + j] = d;
        conf[o + j] = c;
        if (index) index[o + j] = ix;
      }
    }
  }
}

// Output planes per chunk: the largest divisor-like cut of D (multiples of 4, >= 4) that still gives the chip
// ~2.3 workgroups per CU; the whole range when the pixel tiles alone do.
int auto_zchunk(int tiles, int D) {
  const int want = 600;
  if (tiles >= want) return D;
  const int cand[] = {32, 24, 16, 12, 8, 4};
  int best = D;
  for (int zc : cand) {
    if (zc >= D || D % zc) continue;
    best = zc;
    if ((long)tiles * (D / zc) >= want) break;
  }
  return best;
}

}  // namespace

extern "C" int casmvs_prob_regress_supported(int cin, int W) { return cin == 8 && W % 4 == 0 && W >= 4; }

extern "C" int casmvs_prob_regress_f32(const float *packed, const float *in, const float *depth_values, float *cost,
                                       float *depth, float *confidence, int32_t *index, int B, int cin, int D, int h,
                                       int w, float slope, int zchunk, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && cost, "prob_regress: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && D > 0 && h > 0 && w > 0, "prob_regress: bad shape B=%d D=%d h=%d w=%d", B, D, h, w);
  CASMVS_REQUIRE(cin == 8 && w % 4 == 0, "prob_regress: cin=%d w=%d (needs cin == 8, w %% 4 == 0)", cin, w);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(cost)) & 15) == 0, "prob_regress: in / cost must be 16-byte aligned");
  CASMVS_REQUIRE((size_t)cin * D * h * w < ((size_t)1 << 29), "prob_regress: one sample's input tensor must hold < 2^29 floats");
  const bool regress = depth != nullptr;
  if (regress) {
    CASMVS_REQUIRE(depth_values && confidence, "prob_regress: depth_values / confidence are required with depth");
    CASMVS_REQUIRE(((reinterpret_cast<size_t>(depth) | reinterpret_cast<size_t>(confidence)) & 7) == 0, "prob_regress: depth / confidence must be 8-byte aligned");
  }
  CASMVS_REQUIRE(zchunk >= 0, "prob_regress: zchunk=%d", zchunk);
  using Cfg = ProbZCfg;
  const int tiles_x = casmvs::ceil_div(w, Cfg::TX), tiles_y = casmvs::ceil_div(h, Cfg::TY);
  const int zc = zchunk > 0 ? (zchunk < D ? zchunk : D) : auto_zchunk(tiles_x * tiles_y * B, D);
  const int nchunk = casmvs::ceil_div(D, zc);
  const bool fuse = regress && nchunk == 1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(tiles_x * tiles_y * nchunk), (unsigned)B), blk(kThreads);
#define CASMVS_PZ(DT, FUSE)                                                                                          \
  hipLaunchKernelGGL((prob_zwalk_kernel<DT, FUSE>), grid, blk, Cfg::LDS_BYTES, st, in, packed, depth_values, cost, \
                     depth, confidence, index, D, h, w, tiles_x, tiles_y, zc, slope)
  if (!fuse) {
    CASMVS_PZ(0, false);
  } else {
    switch (D) {
      case 8: CASMVS_PZ(8, true); break;
      case 16: CASMVS_PZ(16, true); break;
      case 32: CASMVS_PZ(32, true); break;
      case 48: CASMVS_PZ(48, true); break;
      default: CASMVS_PZ(0, true); break;
    }
  }
#undef CASMVS_PZ
  if (int rc = casmvs::check_launch("prob_zwalk_kernel")) return rc;
  if (regress && !fuse) return casmvs_softmax_regress_f32(cost, depth_values, depth, confidence, index, B, D, h, w, stream);
  return CASMVS_OK;
}
