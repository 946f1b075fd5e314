// Issue cost of the vector instructions the split-f16 producers are made of (v_cvt_pk_f16_f32, v_fma_mix_f32, v_max3_f32, ...): cycles per wave-instruction
// at 1 / 2 waves per SIMD, eight independent register chains (inline asm, so the compiler cannot fold anything).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_mix_rate.hip -o tools/probes/bin/valu_mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define REPS 16
#define KERNEL(name, ASM)                                                                              \
  __global__ __launch_bounds__(256) void name(float *sink, int iters) {                                \
    float a[CHAINS], b[CHAINS];                                                                        \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 1.0f + i * 1e-3f; } \
    for (int it = 0; it < iters; ++it) {                                                               \
      _Pragma("unroll") for (int r = 0; r < REPS; ++r)                                                 \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % CHAINS])); \
    }                                                                                                  \
    float s = 0;                                                                                       \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += a[i];                                      \
    if (s == 12345.678f) sink[0] = s;                                                                  \
  }

KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_cvtpk, "v_cvt_pk_f16_f32 %0, %0, %1")
KERNEL(k_cvtpkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
KERNEL(k_fmamix, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]")
KERNEL(k_fmamixlo, "v_fma_mixlo_f16 %0, %0, %1, %2")
KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %0")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cndmask_nodep, "v_cndmask_b32 %0, %1, %2, vcc")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[2:3]")
KERNEL(k_cndmask_sgpr_nodep, "v_cndmask_b32_e64 %0, %1, %2, s[2:3]")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cmpx_sgpr_cnd, "v_cmp_lt_f32_e64 s[4:5], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[4:5]")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
KERNEL(k_min, "v_min_f32 %0, %0, %1")
KERNEL(k_maxi, "v_max_i32 %0, %0, %1")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_or, "v_or_b32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_addu, "v_add_u32 %0, %0, %1")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_maxdpp, "v_max_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1")
KERNEL(k_cvt_i32, "v_cvt_i32_f32 %0, %0")
KERNEL(k_pkmul_f16, "v_pk_mul_f16 %0, %0, %1")
KERNEL(k_pkfma_f16, "v_pk_fma_f16 %0, %0, %1, %2")

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERNEL2(name, ASM)                                                                             \
  __global__ __launch_bounds__(256) void name(float *sink, int iters) {                                \
    f32x2 a[CHAINS], b[CHAINS];                                                                        \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) { a[i] = f32x2{threadIdx.x * 1e-3f + i, 1.0f}; b[i] = f32x2{1.0f + i * 1e-3f, 1.0f}; } \
    for (int it = 0; it < iters; ++it) {                                                               \
      _Pragma("unroll") for (int r = 0; r < REPS; ++r)                                                 \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % CHAINS])); \
    }                                                                                                  \
    float s = 0;                                                                                       \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += a[i][0] + a[i][1];                         \
    if (s == 12345.678f) sink[0] = s;                                                                  \
  }
KERNEL2(k_pkmul, "v_pk_mul_f32 %0, %0, %1")
KERNEL2(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL2(k_pkadd, "v_pk_add_f32 %0, %0, %1")
KERNEL2(k_mov64, "v_mov_b64 %0, %1")
KERNEL2(k_fmaf64, "v_fma_f64 %0, %0, %1, %2")
KERNEL2(k_lshl64, "v_lshlrev_b64 %0, 3, %0")

struct K { const char *name; void (*fn)(float *, int); };

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float *sink;
  hipMalloc(&sink, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  K ks[] = {{"v_mul_f32", k_mul}, {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_max_f32", k_max}, {"v_max3_f32", k_max3}, {"v_cvt_pk_f16_f32", k_cvtpk},
            {"v_cvt_pkrtz_f16_f32", k_cvtpkrtz}, {"v_fma_mix_f32", k_fmamix}, {"v_fma_mixlo_f16", k_fmamixlo}, {"v_cvt_f32_f16", k_cvt_f32_f16},
            {"v_and_b32", k_and}, {"v_and_or_b32", k_andor}, {"v_perm_b32", k_perm}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask (no self dep)", k_cndmask_nodep}, {"v_cndmask_e64 sgpr", k_cndmask_sgpr},
            {"v_cndmask_e64 sgpr nodep", k_cndmask_sgpr_nodep}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp vcc + v_cndmask", k_cmp_cnd}, {"v_cmp sgpr + v_cndmask", k_cmpx_sgpr_cnd},
            {"v_med3_f32", k_med3}, {"v_bfi_b32", k_bfi}, {"v_min_f32", k_min}, {"v_max_i32", k_maxi}, {"v_fmac_f32", k_fmac}, {"v_sub_f32", k_sub},
            {"v_lshlrev_b32", k_lshl}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_mov_b32", k_mov},
            {"v_lshl_add_u32", k_lshladd}, {"v_add_u32", k_addu}, {"v_mov_b32_dpp", k_dpp}, {"v_max_f32_dpp", k_maxdpp}, {"v_rcp_f32", k_rcp},
            {"v_ldexp_f32", k_ldexp}, {"v_cvt_i32_f32", k_cvt_i32}, {"v_pk_mul_f16", k_pkmul_f16}, {"v_pk_fma_f16", k_pkfma_f16},
            {"v_pk_mul_f32", k_pkmul}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_add_f32", k_pkadd}, {"v_mov_b64", k_mov64},
            {"v_fma_f64", k_fmaf64}, {"v_lshlrev_b64", k_lshl64}};
  const int iters = 2000;
  printf("%s, %d CUs, clock %d kHz; %d independent chains x %d x %d iterations per wave; cycles per wave-instruction (per SIMD)\n", prop.gcnArchName, cus, prop.clockRate, CHAINS, REPS, iters);
  for (auto &k : ks) {
    printf("%-22s", k.name);
    for (int waves = 1; waves <= 2; ++waves) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k.fn, dim3(cus * waves), dim3(256), 0, 0, sink, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("  %d wave(s)/SIMD: %5.2f", waves, best * 1e-3 * prop.clockRate * 1e3 / ((double)CHAINS * REPS * iters * waves));
    }
    printf("\n");
  }
  return 0;
}
