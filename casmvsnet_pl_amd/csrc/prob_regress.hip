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
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"
#include "prob_zwalk.h"
#include "softmax_regress.h"

namespace {

using namespace casmvs::buf;
constexpr int kThreads = 256;

struct ProbZCfg {
  static constexpr int TX = 64, TY = 8;          // output pixels per workgroup; a thread = 2 consecutive x of one row
  static constexpr int IY = TY + 2;              // staged rows y0 - 1 .. y0 + TY
  static constexpr int NPP = TX / 2 + 1;         // position pairs per staged row: positions x0 - 1 .. x0 + TX as (x0 - 1 + 2 m, x0 + 2 m)
  static constexpr int RS = 4 * NPP;             // floats per LDS row: [position][channel of the pair]; rows are contiguous
  static constexpr int SP = IY * RS;             // floats per channel pair of one plane
  static constexpr int NPAIR = 4;                // 8 input channels
  static constexpr int SLOT = NPAIR * SP;        // floats per plane slot
  static constexpr int ITEMS = NPAIR * IY * NPP; // (channel pair, row, position pair) staging items per plane; item e -> floats [4 e, 4 e + 4)
  static constexpr int NK = (ITEMS + kThreads - 1) / kThreads;
  static constexpr size_t LDS_BYTES = 2 * (size_t)SLOT * sizeof(float);
  // Reads: lane (xi = tid & 31, yi = tid >> 5) reads floats [4 xi, 4 xi + 8) of row yi + ky: 32 lanes x 16 B = 512 B per row
  // of lanes, two rows per wave RS floats apart - conflict-free for the 4 x 16-lane groups of ds_read_b128 (RS = 132,
  // enumerated in tests/kernel_model.py: prob_zwalk_bank_cycles).
  // Writes: item e (= thread + 256 k) owns the 16 bytes at float 4 e: one ds_write_b128 per item, consecutive lanes write
  // consecutive 16-byte units - conflict-free.  (First version: a thread staged 4 x of a channel pair as four
  // ds_write_b64 32 bytes apart from its neighbour's = 4-way bank conflicts; PMC: SQ_LDS_BANK_CONFLICT 47 % of
  // SQ_LDS_IDX_ACTIVE, LDS-active time = the kernel's run time, VALU busy 29 % - profiles/r03_pmc_prob_zwalk.txt.)
  static_assert(RS % 4 == 0, "rows start 16-byte aligned");
};

// grid: x = tiles_x * tiles_y * chunks (XCD-major, chunk fastest, then x, then y), y = batch.
// in (B, 8, Di, Hi, Wi), Wi % 4 == 0, 16-byte aligned; cost (B, Di, Hi, Wi) is always written.
// FUSE: chunks == 1; dvals (B, Di, Hi, Wi) -> depth, conf (B, Hi, Wi) [, index].  DT: compile-time Di of the fused phase (0: generic).
template <int DT, bool FUSE>
__global__ __launch_bounds__(kThreads, 3) void prob_zwalk_kernel(
    const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ dvals, float *cost,
    float *__restrict__ depth, float *__restrict__ conf, int32_t *__restrict__ index, int Di, int Hi, int Wi,
    int tiles_x, int tiles_y, int zc, float slope) {
  using Cfg = ProbZCfg;
  constexpr int NK = Cfg::NK, RS = Cfg::RS, SLOTS = Cfg::SLOT, NPP = Cfg::NPP, IY = Cfg::IY;
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

  // staging plan (tile constants): item e = tid + 256 k -> (channel pair, staged row, position pair m): the two positions
  // x = x0 - 1 + 2 m, x + 1 of BOTH channels of the pair = two 8-byte loads -> (c0[x], c1[x], c0[x+1], c1[x+1]).
  // x is odd: a pair straddles the image's left edge (x = -1: m = 0 of the first tile column) or its right edge
  // (x + 1 = Wi); there the pair one position further inside is loaded and shifted (edge tiles only: wave-uniform branch).
  int voff0[NK], voff1[NK];
  bool edge_l[NK], edge_r[NK];
  const bool edge_tile = tx0 == 0 || tx0 + Cfg::TX >= Wi;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int e = threadIdx.x + k * kThreads;
    const int p = e / (IY * NPP), r = e - p * (IY * NPP);
    const int iy = r / NPP, m = r - iy * NPP;
    const int gy = ty0 - 1 + iy, x = tx0 - 1 + 2 * m;
    edge_l[k] = x < 0;
    edge_r[k] = x + 1 == Wi;
    const int xl = edge_l[k] ? 0 : (edge_r[k] ? x - 1 : x);   // first element of the loaded pair: inside the row
    const bool ok = e < Cfg::ITEMS && gy >= 0 && gy < Hi && x < Wi;
    voff0[k] = ok ? ((2 * p) * in_cs + gy * Wi + xl) * 4 : kOOB;
    voff1[k] = ok ? ((2 * p + 1) * in_cs + gy * Wi + xl) * 4 : kOOB;
  }
  // two register sets: the plane after next is in flight from memory while the next one waits in registers for its LDS
  // slot - every load has a whole plane step plus its own to land (with one set the load of plane z + 1 had to land within
  // the FMAs of plane z: ~1.3 us, about the memory latency once HBM is busy)
#ifndef CASMVS_PROB_PREFETCH
#define CASMVS_PROB_PREFETCH 2   // planes in flight ahead of the one being multiplied (1: A/B builds)
#endif
  constexpr int NSET = CASMVS_PROB_PREFETCH;
  static_assert(NSET == 1 || NSET == 2, "one or two staging register sets");
  f32x2 v0[NSET][NK], v1[NSET][NK];
  // Loads, LDS stores and the cost store of a step are issued UNCONDITIONALLY (a plane that does not exist is read through
  // an empty buffer descriptor = zeros; a cost plane outside the chunk is stored to an out-of-range offset = dropped): with
  // a vector-memory operation inside a branch the compiler's wait-count pass must assume it did not execute and waits
  // vmcnt(0) for the OLDER set - i.e. also for the loads it has just issued (seen in the ISA of the first version).
  const rsrc_t none = make_rsrc(in, 0);
  auto load_plane = [&](auto set_, int z, bool exists) {
    constexpr int S = decltype(set_)::value;
    const rsrc_t r = exists ? src : none;
    const int soff = exists ? z * HiWi * 4 : 0;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if (CASMVS_PZ_ABL & 4) {
        v0[S][k] = f32x2{(float)z, 1.f};
        v1[S][k] = v0[S][k];
        continue;
      }
      v0[S][k] = buf_load2(r, voff0[k], soff);
      v1[S][k] = buf_load2(r, voff1[k], soff);
    }
  };
  auto store_plane = [&](auto set_, float *slot) {
    constexpr int S = decltype(set_)::value;
    if (edge_tile) {   // (x = -1, x = 0) was loaded as (0, 1): keep (zero, [0]); (Wi - 1, Wi) as (Wi - 2, Wi - 1): keep ([1], zero)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const f32x2 a = v0[S][k], c = v1[S][k];
        v0[S][k] = edge_l[k] ? f32x2{0.f, a[0]} : (edge_r[k] ? f32x2{a[1], 0.f} : a);
        v1[S][k] = edge_l[k] ? f32x2{0.f, c[0]} : (edge_r[k] ? f32x2{c[1], 0.f} : c);
      }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int e = threadIdx.x + k * kThreads;
      if ((k < NK - 1 || e < Cfg::ITEMS) && !((CASMVS_PZ_ABL & 8) && v0[S][k][0] != 12345.f))
        *reinterpret_cast<f32x4v *>(slot + 4 * e) = f32x4v{v0[S][k][0], v1[S][k][0], v0[S][k][1], v1[S][k][1]};
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, NSET - 1>;

  const int oy = ty0 + yi, ox = tx0 + 2 * xi;
  const int out_voff = (oy < Hi && ox < Wi) ? (oy * Wi + ox) * 4 : kOOB;  // Wi even: the pixel pair is inside or outside
  f32x2 A[3][2];  // [0]: output plane z_in - 1, [1]: z_in, [2]: z_in + 1; [pixel]; (even, odd channels' partial sums)
#pragma unroll
  for (int i = 0; i < 3; ++i) A[i][0] = A[i][1] = f32x2{0.f, 0.f};

  const int nplanes = z_hi - z_lo + 2;  // input planes z_lo - 1 .. z_hi (step `it` multiplies plane z_lo - 1 + it)
  const int z0 = z_lo - 1;
  // step `it`, staging set PAR = it & 1 (NSET == 2): LDS slot PAR holds plane z0 + it; set 1 - PAR holds plane z0 + it + 1
  // (in flight since step it - 1 or the prologue); set PAR is free -> plane z0 + it + 2 is put in flight into it
  auto step = [&](int it, auto par_) {
    constexpr int PAR = decltype(par_)::value;
    using Mine = std::integral_constant<int, NSET == 2 ? PAR : 0>;
    using Other = std::integral_constant<int, NSET == 2 ? 1 - PAR : 0>;
    const int zin = z0 + it;
    const float *cur = smem + (it & 1) * SLOTS;
    if (NSET == 2) load_plane(Mine{}, zin + 2, it + 2 < nplanes && zin + 2 < Di);
    else load_plane(Mine{}, zin + 1, it + 1 < nplanes && zin + 1 < Di);
    if (zin >= 0 && zin < Di) {
      const float *rows = cur + yi * RS + 4 * xi;
      if (it == 0) casmvs::pz::zwalk_plane<1, Cfg::SP, Cfg::RS>(rows, wpk, A);                 // only output plane z_lo takes from plane z_lo - 1
      else if (it == nplanes - 1) casmvs::pz::zwalk_plane<4, Cfg::SP, Cfg::RS>(rows, wpk, A);  // only output plane z_hi - 1 takes from plane z_hi
      else casmvs::pz::zwalk_plane<7, Cfg::SP, Cfg::RS>(rows, wpk, A);
    }
    {  // output plane zin - 1 is complete; it lies in [z_lo, z_hi) from step 2 on
      float o0 = fmaf(A[0][0][0] + A[0][0][1], sc0, sh0), o1 = fmaf(A[0][1][0] + A[0][1][1], sc0, sh0);
      o0 = o0 > 0.0f ? o0 : o0 * slope;
      o1 = o1 > 0.0f ? o1 : o1 * slope;
      buf_store2(f32x2{o0, o1}, dst, it >= 2 ? out_voff : kOOB, it >= 2 ? (zin - 1) * HiWi * 4 : 0);
    }
    A[0][0] = A[1][0];
    A[0][1] = A[1][1];
    A[1][0] = A[2][0];
    A[1][1] = A[2][1];
    A[2][0] = A[2][1] = f32x2{0.f, 0.f};
    store_plane(Other{}, smem + ((it + 1) & 1) * SLOTS);   // (zeros when plane zin + 1 does not exist: never multiplied)
    if (!(CASMVS_PZ_ABL & 16)) __syncthreads();  // the other slot is published, this one is free
  };
  // prologue: plane z0 -> set 0 -> slot 0; plane z0 + 1 -> set 1 (in flight)
  load_plane(Set0{}, z0, z0 >= 0);
  if (NSET == 2) load_plane(Set1{}, z0 + 1, nplanes > 1 && z0 + 1 < Di);
  store_plane(Set0{}, smem);
  __syncthreads();
  for (int it = 0; it < nplanes; it += 2) {
    step(it, Set0{});
    if (it + 1 < nplanes) step(it + 1, std::integral_constant<int, 1>{});
  }

  if constexpr (FUSE) {
    // every cost value of this thread's two pixels was stored by this thread: wait for the stores, then read them
    // back (the "memory" clobber also keeps the compiler from moving the loads above the buffer stores)
#ifndef HIPEMU_LDS_BYTES   // (tests/hipemu: the host has no such instruction, and needs none)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
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
