// Weight gradient of CostRegNet's `prob` layer (Conv3d 8 -> 1, k3 s1 p1; models/mvsnet.py:89) - training path, SURVEY 8 f-2.
//
//   grad_weight[c][kz][ky][kx] = sum over (b, z, y, x) of grad_out[b][z][y][x] * in[b][c][z + kz - 1][y + ky - 1][x + kx - 1]
//
// The generic kernel (train.hip: conv_wgrad_kernel) puts output channels on the rows of a 16 x 16 MFMA tile: with ONE output
// channel 15 of 16 rows are padding, and this layer - 216 numbers - cost ~1.0 ms of a 15 ms training step.
// It is 216 dot products over the whole volume, 27 FMAs per (voxel, channel): vector-ALU work.  Here a thread owns four
// x-consecutive voxels of a 4 x 8 x 32 tile and the 54 accumulators of TWO input channels (blockIdx.y picks the channel pair:
// all 216 in one thread spilled, and at <= 128 registers four workgroups share a CU and hide each other's load latency -
// 108 accumulators / two workgroups per CU ran at 85 us for the 8 x 512 x 640 volume); per input channel the tile's halo box
// (6 x 10 x 34 floats) is staged in LDS once, a thread reads the six floats of a (kz, ky) row with one 16-byte and one 8-byte
// read and feeds 12 FMAs.  Persistent workgroups; at the end the accumulators are summed over rows of 16 lanes (DPP), the sixteen
// rows of the workgroup (LDS) and - by a second kernel, in a fixed order - the workgroups: deterministic, no atomics.
//
// Added at the end of round 3; validated through tools/native/prob_wgrad_check.cpp (against conv_wgrad_kernel through the C ABI).
#include "buffer_ops.h"
#include "common.h"

namespace {

using namespace casmvs::buf;

struct PwCfg {
  static constexpr int THREADS = 256, TZ = 4, TY = 8, TX = 32;
  static constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 2;
  static constexpr int RS = 36;                        // floats per staged row (16-byte aligned 6-float windows at 4 xg)
  static constexpr int PLANE = IY * RS, BOX = IZ * PLANE;   // 2160 floats = 8640 B per channel
  static constexpr int ITEMS = IZ * IY * IX;           // 2040 staged floats per (tile, channel)
  static constexpr int NR = (ITEMS + THREADS - 1) / THREADS;   // 8 loads per thread
  static constexpr int CG = 2, NACC = CG * 27, NGROUPS_C = 8 / CG;   // channels per workgroup, its sums, channel groups
  static constexpr int MAX_GROUPS = 512;               // workgroups per channel pair (x NGROUPS_C in the grid)
  static constexpr int DUMMY = BOX;                    // where the staging rounds past the last item write
  static constexpr int LDS_FLOATS = (BOX + 4 > 16 * NACC ? BOX + 4 : 16 * NACC);
};

// sum over a row of 16 lanes, left in every lane of the row (the same four DPP exchanges as split_f16.h's wave maximum)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

__global__ __launch_bounds__(PwCfg::THREADS, 4) void prob_wgrad_kernel(const float *__restrict__ in, const float *__restrict__ gout,
                                                                      float *__restrict__ partial, int B, int D, int H, int W, int tiles_x,
                                                                      int tiles_y, int tiles_z) {
  using Cfg = PwCfg;
  constexpr int NR = Cfg::NR, RS = Cfg::RS, PLANE = Cfg::PLANE;
  __shared__ __attribute__((aligned(16))) float box[Cfg::LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tz = wave, ty = (tid >> 3) & 7, xg = tid & 7;   // the thread's voxels: (tz, ty, 4 xg .. 4 xg + 3) of the tile
  const int total = tiles_x * tiles_y * tiles_z * B;
  const int HW = H * W, cs = D * HW;
  constexpr int CG = Cfg::CG;
  const int c0 = blockIdx.y * CG;
  float acc[CG][27];
#pragma unroll
  for (int c = 0; c < CG; ++c)
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[c][k] = 0.0f;
  // LDS position of the thread's staging items (the same for every tile)
  int loff[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = tid + r * Cfg::THREADS;
    const int iz = e / (Cfg::IY * Cfg::IX), rem = e - iz * (Cfg::IY * Cfg::IX), iy = rem / Cfg::IX, ix = rem - iy * Cfg::IX;
    loff[r] = e < Cfg::ITEMS ? iz * PLANE + iy * RS + ix : Cfg::DUMMY;
  }
  const int rbase = tz * PLANE + ty * RS + 4 * xg;   // window of (kz, ky): rbase + kz * PLANE + ky * RS, floats [0, 6)

  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    int it = xcd_major(item, total);
    const int z0 = (it % tiles_z) * Cfg::TZ;
    it /= tiles_z;
    const int x0 = (it % tiles_x) * Cfg::TX;
    it /= tiles_x;
    const int y0 = (it % tiles_y) * Cfg::TY, b = it / tiles_y;
    const rsrc_t src = make_rsrc(in + (size_t)b * 8 * cs, (size_t)8 * cs * 4);
    const rsrc_t gsrc = make_rsrc(gout + (size_t)b * cs, (size_t)cs * 4);
    int voff[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = tid + r * Cfg::THREADS;
      const int iz = e / (Cfg::IY * Cfg::IX), rem = e - iz * (Cfg::IY * Cfg::IX), iy = rem / Cfg::IX, ix = rem - iy * Cfg::IX;
      const int gz = z0 - 1 + iz, gy = y0 - 1 + iy, gx = x0 - 1 + ix;
      const bool ok = e < Cfg::ITEMS && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
      voff[r] = ok ? (gz * HW + gy * W + gx) * 4 : kOOB;
    }
    const int oz = z0 + tz, oy = y0 + ty, ox = x0 + 4 * xg;
    const f32x4v g = buf_load4(gsrc, (oz < D && oy < H && ox < W) ? (oz * HW + oy * W + ox) * 4 : kOOB, 0);   // W % 4 == 0: all four or none
    float R[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) R[r] = buf_load(src, voff[r], c0 * cs * 4);
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      __syncthreads();   // the previous channel's (or tile's) window reads are done
#pragma unroll
      for (int r = 0; r < NR; ++r) box[loff[r]] = R[r];
      if (c + 1 < CG) {
#pragma unroll
        for (int r = 0; r < NR; ++r) R[r] = buf_load(src, voff[r], (c0 + c + 1) * cs * 4);
      }
      __syncthreads();
#pragma unroll
      for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float *wrow = box + rbase + kz * PLANE + ky * RS;
          const f32x4v lo = *reinterpret_cast<const f32x4v *>(wrow);
          const f32x2 hi = *reinterpret_cast<const f32x2 *>(wrow + 4);
          const float w6[6] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1]};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            float s = acc[c][(kz * 3 + ky) * 3 + kx];
#pragma unroll
            for (int j = 0; j < 4; ++j) s = fmaf(g[j], w6[j + kx], s);
            acc[c][(kz * 3 + ky) * 3 + kx] = s;
          }
        }
    }
  }
  // ---- 54 sums: over the rows of 16 lanes (DPP adds: no LDS round trips), then the 16 rows of the workgroup in a fixed order ----
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CG; ++c)
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[c][k] = row16_sum(acc[c][k]);
  if ((lane & 15) == 0) {
    float *dstrow = box + (wave * 4 + (lane >> 4)) * Cfg::NACC;
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int k = 0; k < 27; ++k) dstrow[c * 27 + k] = acc[c][k];
  }
  __syncthreads();
  if (tid < Cfg::NACC) {
    float t[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) t[w] = (box[(4 * w) * Cfg::NACC + tid] + box[(4 * w + 1) * Cfg::NACC + tid]) + (box[(4 * w + 2) * Cfg::NACC + tid] + box[(4 * w + 3) * Cfg::NACC + tid]);
    partial[((size_t)blockIdx.x * Cfg::NGROUPS_C + blockIdx.y) * Cfg::NACC + tid] = (t[0] + t[1]) + (t[2] + t[3]);
  }
}

__global__ __launch_bounds__(256) void prob_wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ gw, int groups) {
  const int i = threadIdx.x;   // = channel * 27 + tap = (channel group) * NACC + index inside the group's row
  constexpr int ROW = PwCfg::NGROUPS_C * PwCfg::NACC;
  if (i >= ROW) return;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int w = 0;
  for (; w + 4 <= groups; w += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += partial[(size_t)(w + k) * ROW + i];
  }
  for (; w < groups; ++w) a[w & 3] += partial[(size_t)w * ROW + i];
  gw[i] = (a[0] + a[1]) + (a[2] + a[3]);
}

}  // namespace

extern "C" int casmvs_prob_wgrad_supported(int B, int D, int H, int W) {
  return B > 0 && D > 0 && H > 0 && W >= 4 && W % 4 == 0 && (size_t)8 * D * H * W * 4 < ((size_t)1 << 31);
}

extern "C" size_t casmvs_prob_wgrad_workspace_bytes(int B, int D, int H, int W) {
  return casmvs_prob_wgrad_supported(B, D, H, W) ? (size_t)PwCfg::MAX_GROUPS * PwCfg::NGROUPS_C * PwCfg::NACC * sizeof(float) : 0;
}

extern "C" int casmvs_prob_wgrad_f32(const float *in, const float *grad_out, float *grad_weight, void *workspace, int B, int D, int H, int W,
                                     void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(in && grad_out && grad_weight && workspace, "prob_wgrad: null pointer");
  CASMVS_REQUIRE(casmvs_prob_wgrad_supported(B, D, H, W), "prob_wgrad: B=%d D=%d H=%d W=%d (W %% 4 == 0, 8 D H W < 2^29)", B, D, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(grad_out)) & 15) == 0, "prob_wgrad: 16-byte aligned tensors");
  using Cfg = PwCfg;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY), tiles_z = casmvs::ceil_div(D, Cfg::TZ);
  const long total = (long)tiles_x * tiles_y * tiles_z * B;
  CASMVS_REQUIRE(total < (1L << 31), "prob_wgrad: too many tiles");
  int groups = casmvs::resident_blocks(reinterpret_cast<const void *>(prob_wgrad_kernel), Cfg::THREADS, 0);
  groups /= Cfg::NGROUPS_C;   // blockIdx.y doubles the grid
  if (groups > Cfg::MAX_GROUPS) groups = Cfg::MAX_GROUPS;
  if (groups > total) groups = (int)total;
  if (groups < 1) groups = 1;
  hipStream_t st = (hipStream_t)stream;
  float *partial = static_cast<float *>(workspace);
  hipLaunchKernelGGL(prob_wgrad_kernel, dim3((unsigned)groups, (unsigned)Cfg::NGROUPS_C), dim3(Cfg::THREADS), 0, st, in, grad_out, partial, B, D, H, W, tiles_x, tiles_y, tiles_z);
  if (int rc = casmvs::check_launch("prob_wgrad_kernel")) return rc;
  hipLaunchKernelGGL(prob_wgrad_reduce_kernel, dim3(1), dim3(256), 0, st, partial, grad_weight, groups);
  return casmvs::check_launch("prob_wgrad_reduce_kernel");
}
