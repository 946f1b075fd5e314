// Stand-alone reproducer of the co-residency fault of DESIGN.md section 3 (round 5): ONE instruction,
//     v_pk_fma_f32 vD, vA, vS, vB op_sel:[0,1,1]          (low = A.lo * S.hi + B.hi, high = A.hi * S.hi + B.hi)
// executed by a wave that shares its SIMD with f16 / bf16 matrix instructions of another kernel (another stream), returns low = B.hi (the product
// term is lost) in lanes 48-63.  This is the instruction the compiler picks for the Cout = 8 float32 layer kernel's epilogue (conv16db_kernel<PX>,
// csrc/conv3d_mfma.hip: channel 2 kq + 1 of column tile 0 - exactly the element that was wrong in every failing round).
//   hipcc -O2 --offload-arch=gfx950 tools/probes/pk_fma_opsel_repro.hip -o tools/probes/bin/pk_fma_opsel_repro
//   pk_fma_opsel_repro [rounds = 5 [more]]      (more: three further matrix-instruction neighbours)
// Per packed form (FORMS below) x neighbour (none, f32 MFMA, f16 MFMA, bf16 MFMA, VALU): wrong results by lane quarter and half; the last two lines add matrix
// instructions of the wave itself in front of every packed instruction (the fault does not need them).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// 1 f32 MFMA 16x16x4 (32-bit A / B operands), 2 f16 16x16x32 (128-bit), 3 bf16 16x16x32 (128-bit), 4 VALU; argv[2] = "more" adds 5 f16 16x16x16 (64-bit
// operands), 6 f16 32x32x16 (128-bit, 16 passes), 7 i8 16x16x64 (128-bit)
template <int KIND>
__global__ __launch_bounds__(256) void neighbour(float *sink, int iters) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (KIND >= 5) {
    f32x16 c16 = {};
    i32x4 ic = {0, 0, 0, 0};
    const f16x4 h4 = {(_Float16)1, (_Float16)1, (_Float16)1, (_Float16)1};
    const u32x4 ub = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    for (int it = 0; it < iters; ++it) {
      if (KIND == 5) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(h4, h4, acc, 0, 0, 0);
      else if (KIND == 6) c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ub), __builtin_bit_cast(f16x8, ub), c16, 0, 0, 0);
      else ic = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, ub), __builtin_bit_cast(i32x4, ub), ic, 0, 0, 0);
    }
    if (acc[0] + c16[0] + c16[15] + (float)ic[0] == 12345.678f) sink[0] = acc[0];
    return;
  }
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  const u32x4 ua = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  for (int it = 0; it < iters; ++it) {
    if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    else if (KIND == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, ua), acc, 0, 0, 0);
    else if (KIND == 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ua), acc, 0, 0, 0);
    else { acc[0] = fmaf(acc[0], a, b); acc[1] = fmaf(acc[1], a, b); }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// The packed float32 forms under test: OP 0 v_pk_fma_f32 (R = A * S + B), 1 v_pk_mul_f32 (A * S), 2 v_pk_add_f32 (A + S), 3 two v_fma_f32, 4 v_pk_fma_f32 with S in
// scalar registers; LO / HI = which 32-bit half of (A, S, B) the low / high result takes: bit 0 A, bit 1 S, bit 2 B (op_sel / op_sel_hi of the instruction).
// Every (op_sel, op_sel_hi) combination of the three instructions is run.
#define F3(a, s, b, ha, hs, hb) X(0, (a | s << 1 | b << 2), (ha | hs << 1 | hb << 2), "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[" #a "," #s "," #b "] op_sel_hi:[" #ha "," #hs "," #hb "]")
#define F2(op, name, a, s, ha, hs) X(op, (a | s << 1), (ha | hs << 1), name " %0, %1, %2 op_sel:[" #a "," #s "] op_sel_hi:[" #ha "," #hs "]")
#define HI3(a, s, b) F3(a, s, b, 0, 0, 0) F3(a, s, b, 1, 0, 0) F3(a, s, b, 0, 1, 0) F3(a, s, b, 1, 1, 0) F3(a, s, b, 0, 0, 1) F3(a, s, b, 1, 0, 1) F3(a, s, b, 0, 1, 1) F3(a, s, b, 1, 1, 1)
#define HI2(op, name, a, s) F2(op, name, a, s, 0, 0) F2(op, name, a, s, 1, 0) F2(op, name, a, s, 0, 1) F2(op, name, a, s, 1, 1)
#define ALL2(op, name) HI2(op, name, 0, 0) HI2(op, name, 1, 0) HI2(op, name, 0, 1) HI2(op, name, 1, 1)
#define FORMS(X)                                                                                                     \
  HI3(0, 0, 0) HI3(1, 0, 0) HI3(0, 1, 0) HI3(1, 1, 0) HI3(0, 0, 1) HI3(1, 0, 1) HI3(0, 1, 1) HI3(1, 1, 1)            \
  ALL2(1, "v_pk_mul_f32") ALL2(2, "v_pk_add_f32")                                                                    \
  X(4, 2, 7, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]")   /* S in scalar registers */                             \
  X(3, 6, 7, "two v_fma_f32")                                                                                        \
  /* the other instructions of the library that carry op_sel: OP 5 v_pk_mov_b32 (low = A[op_sel 0], high = S[op_sel 1]), 6 v_fma_mix_f32 (B = a float16 half) */ \
  X(5, 0, 0, "v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]") X(5, 1, 0, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]")              \
  X(5, 2, 0, "v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]") X(5, 3, 0, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]")              \
  X(6, 0, 0, "v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]") X(6, 4, 0, "v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]") \
  X(6, 2, 1, "v_fma_mix_f32 %0, %1, %3, %2 op_sel:[0,1,0] op_sel_hi:[0,1,0]")

// counts[quarter * 2 + half]: wrong results of lanes 16 quarter .. 16 quarter + 15 (half 0 = low, 1 = high); counts[8]: wrong results that equal the addend B
// (the product term lost); OWN: matrix instructions of the wave itself in front of every packed instruction
template <int OP, int LO, int HI, int OWN>
__global__ __launch_bounds__(256, 2) void victim(unsigned *counts, float *sink, int iters) {
  extern __shared__ float lds[];   // (48 KiB requested: two workgroups per CU, as the library kernel)
  const int lane = threadIdx.x & 63;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const float ma = 1.0f + lane * 1e-3f, mb = 0.25f;
  unsigned wrong_lo = 0, wrong_hi = 0, lost = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < OWN; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[k & 3], 0, 0, 0);
    // operands that differ per lane and per iteration; the halves an instruction does NOT select are decoys
    f32x2 A = {1.0f + 0.015625f * lane + it, 2.0f + 0.03125f * lane}, S = {-7.0f, 1.5f + (it & 3)}, B = {1000.0f, 0.06f + 0.001f * lane}, R;
    unsigned HB = 0x3c004200u + (lane << 16) + (it & 7);   // two float16 halves (v_fma_mix_f32)
    asm volatile("" : "+v"(A), "+v"(S), "+v"(B), "+v"(HB));
#define X(op, lo, hi, text)                                                                                                          \
  if constexpr (OP == op && LO == lo && HI == hi && op != 3) {                                                                       \
    if constexpr (op == 4) asm volatile(text : "=&v"(R) : "v"(A), "s"(f32x2{-7.0f, 2.5f}), "v"(B));                                   \
    else if constexpr (op == 6) asm volatile(text : "=&v"(R[0]) : "v"(A[0]), "v"(S[1]), "v"(HB));                                    \
    else asm volatile(text : "=&v"(R) : "v"(A), "v"(S), "v"(B));   /* (the two-operand forms do not name %3) */                      \
  }
    FORMS(X)
#undef X
    if constexpr (OP == 4) S = f32x2{-7.0f, 2.5f};
    if constexpr (OP == 3) asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %5" : "=&v"(R[0]), "=&v"(R[1]) : "v"(A[0]), "v"(A[1]), "v"(S[1]), "v"(B[1]));
    float e[2];   // expected: separately rounded scalar instructions on the selected halves
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int sel = h ? HI : LO;
      const float a = A[sel & 1], sv = S[(sel >> 1) & 1], b = B[(sel >> 2) & 1];
      if (OP == 5) e[h] = h ? S[(LO >> 1) & 1] : A[LO & 1];
      else if (OP == 6) {   // LO bit 2: which half of HB is the addend (negated), bit 1: the multiplier is the high half of HB (HI = 1) instead of S.hi
        const _Float16 hc = __builtin_bit_cast(_Float16, (unsigned short)(LO & 4 ? HB >> 16 : HB)), hs = __builtin_bit_cast(_Float16, (unsigned short)(HB >> 16));
        if (HI) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[h]) : "v"(A[0]), "v"((float)hs), "v"(S[1]));
        else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[h]) : "v"(A[0]), "v"(S[1]), "v"(-(float)hc));
        if (h) e[h] = R[1] = 0.f;
      } else if (OP == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e[h]) : "v"(a), "v"(sv));
      else if (OP == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(e[h]) : "v"(a), "v"(sv));
      else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[h]) : "v"(a), "v"(sv), "v"(b));
    }
    wrong_lo += __float_as_uint(R[0]) != __float_as_uint(e[0]);
    wrong_hi += __float_as_uint(R[1]) != __float_as_uint(e[1]);
    lost += (__float_as_uint(R[0]) != __float_as_uint(e[0]) && __float_as_uint(R[0]) == __float_as_uint(B[(LO >> 2) & 1])) +
            (__float_as_uint(R[1]) != __float_as_uint(e[1]) && __float_as_uint(R[1]) == __float_as_uint(B[(HI >> 2) & 1]));
  }
  if (wrong_lo) atomicAdd(&counts[(lane >> 4) * 2], wrong_lo);
  if (wrong_hi) atomicAdd(&counts[(lane >> 4) * 2 + 1], wrong_hi);
  if (lost) atomicAdd(&counts[8], lost);
  float s = 0;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == 12345.678f) sink[0] = s + lds[lane];
}

static int g_neighbours = 5;

template <int OP, int LO, int HI, int OWN>
static void run(const char *fname, int rounds, int cus, hipStream_t sa, hipStream_t sb, unsigned *counts, float *sink) {
  const char *nnames[] = {"none", "f32mfma", "f16mfma", "bf16mfma", "valu", "f16mfma_16x16x16(64-bit operands)", "f16mfma_32x32x16", "i8mfma_16x16x64"};
  CHECK(hipFuncSetAttribute((const void *)victim<OP, LO, HI, OWN>, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
  printf("%-58s own MFMA %2d:", fname, OWN);
  for (int k = 0; k < g_neighbours; ++k) {
    CHECK(hipMemset(counts, 0, 9 * sizeof(unsigned)));
    for (int r = 0; r < rounds; ++r) {
      for (int i = 0; i < 6 && k; ++i) {
        if (k == 1) neighbour<1><<<cus * 2, 256, 0, sa>>>(sink, 20000);
        if (k == 2) neighbour<2><<<cus * 2, 256, 0, sa>>>(sink, 20000);
        if (k == 3) neighbour<3><<<cus * 2, 256, 0, sa>>>(sink, 20000);
        if (k == 4) neighbour<4><<<cus * 2, 256, 0, sa>>>(sink, 20000);
        if (k == 5) neighbour<5><<<cus * 2, 256, 0, sa>>>(sink, 20000);
        if (k == 6) neighbour<6><<<cus * 2, 256, 0, sa>>>(sink, 10000);
        if (k == 7) neighbour<7><<<cus * 2, 256, 0, sa>>>(sink, 20000);
      }
      for (int i = 0; i < 4; ++i) victim<OP, LO, HI, OWN><<<cus, 256, 48 * 1024, sb>>>(counts, sink, 500);
      CHECK(hipDeviceSynchronize());
    }
    unsigned h[9];
    CHECK(hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost));
    unsigned total = 0;
    for (int i = 0; i < 8; ++i) total += h[i];
    printf("  %s %u", nnames[k], total);
    if (total) printf(" (low half by lane quarter %u %u %u %u, high %u %u %u %u; product lost %u)", h[0], h[2], h[4], h[6], h[1], h[3], h[5], h[7], h[8]);
  }
  printf("\n");
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  if (argc > 2 && !strcmp(argv[2], "more")) g_neighbours = 8;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipStream_t sa, sb;
  CHECK(hipStreamCreate(&sa));
  CHECK(hipStreamCreate(&sb));
  unsigned *counts;
  float *sink;
  CHECK(hipMalloc(&counts, 9 * sizeof(unsigned)));
  CHECK(hipMalloc(&sink, 1024));
  printf("%s, %d CUs; %d rounds of 6 neighbour launches + 4 victim launches (each %d workgroups x 4 waves x 500 packed instructions per lane); wrong results:\n",
         prop.gcnArchName, cus, rounds, cus);
#define X(op, lo, hi, text) run<op, lo, hi, 0>(op == 4 ? text " (S in SGPRs)" : text, rounds, cus, sa, sb, counts, sink);
  FORMS(X)
#undef X
  run<0, 6, 7, 4>("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]", rounds, cus, sa, sb, counts, sink);
  run<0, 6, 7, 16>("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]", rounds, cus, sa, sb, counts, sink);
  return 0;
}
