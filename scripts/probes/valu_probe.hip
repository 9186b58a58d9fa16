// What one vector instruction costs on gfx950: every wave of a full chip (8 waves per SIMD) runs a loop of 16 independent
// copies of one instruction; cycles per wave-instruction and SIMD = time * clock / (waves per SIMD * instructions).
//   hipcc --offload-arch=gfx950 -O3 -o valu_probe scripts/probes/valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define PROBE(NAME, ASM)                                                                          \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                           \
    float a[16], b = out[threadIdx.x & 7] + 1.5f, c = out[(threadIdx.x & 7) + 8] + 0.25f;         \
    for (int k = 0; k < 16; ++k) a[k] = b + (float)k;                                             \
    for (int i = 0; i < iters; ++i) {                                                             \
      _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); \
    }                                                                                             \
    float s = 0; for (int k = 0; k < 16; ++k) s += a[k];                                          \
    if (s == 123.456f) out[0] = s;                                                                \
  }
// 64-bit operands: pairs of registers
#define PROBE2(NAME, ASM)                                                                         \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                           \
    double a[16], b = (double)out[threadIdx.x & 7] + 1.5, c = (double)out[(threadIdx.x & 7) + 8] + 0.25; \
    for (int k = 0; k < 16; ++k) a[k] = b + (double)k;                                            \
    for (int i = 0; i < iters; ++i) {                                                             \
      _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); \
    }                                                                                             \
    double s = 0; for (int k = 0; k < 16; ++k) s += a[k];                                         \
    if (s == 123.456) out[0] = (float)s;                                                          \
  }

// one dependent chain per wave instead of 16 independent ones: what a wave pays when the next instruction needs the result
#define CHAIN(NAME, ASM)                                                                          \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                           \
    float a = out[threadIdx.x & 7] + 1.5f, b = a + 1.0f, c = out[(threadIdx.x & 7) + 8] + 0.25f;  \
    for (int i = 0; i < iters; ++i) {                                                             \
      _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
    }                                                                                             \
    if (a == 123.456f) out[0] = a;                                                                \
  }
#define CHAIN2(NAME, ASM)                                                                         \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                           \
    double a = (double)out[threadIdx.x & 7] + 1.5, b = a + 1.0, c = (double)out[(threadIdx.x & 7) + 8] + 0.25; \
    for (int i = 0; i < iters; ++i) {                                                             \
      _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(ASM : "+v"(a) : "v"(b), "v"(c)); \
    }                                                                                             \
    if (a == 123.456) out[0] = (float)a;                                                          \
  }
CHAIN(c_fma, "v_fma_f32 %0, %1, %2, %0")
CHAIN(c_max, "v_max_f32 %0, %1, %0")
CHAIN(c_min3, "v_min3_f32 %0, %1, %2, %0")
CHAIN(c_cnd, "v_cndmask_b32 %0, %1, %0, vcc")
CHAIN(c_lshladd, "v_lshl_add_u32 %0, %0, 3, %1")
CHAIN(c_and, "v_and_b32 %0, %1, %0")
CHAIN2(c_pkfma, "v_pk_fma_f32 %0, %1, %2, %0")
// the same with the IEEE bit of the MODE register cleared (no quieting of signalling NaNs in min / max)
#define PROBE_NOIEEE(NAME, ASM)                                                                   \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                           \
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");                               \
    float a[16], b = out[threadIdx.x & 7] + 1.5f, c = out[(threadIdx.x & 7) + 8] + 0.25f;         \
    for (int k = 0; k < 16; ++k) a[k] = b + (float)k;                                             \
    for (int i = 0; i < iters; ++i) {                                                             \
      _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); \
    }                                                                                             \
    float s = 0; for (int k = 0; k < 16; ++k) s += a[k];                                          \
    if (s == 123.456f) out[0] = s;                                                                \
  }
PROBE_NOIEEE(n_max, "v_max_f32 %0, %1, %0")
PROBE_NOIEEE(n_min3, "v_min3_f32 %0, %1, %2, %0")
PROBE_NOIEEE(n_fma, "v_fma_f32 %0, %1, %2, %0")
PROBE(p_cnd64v, "v_cndmask_b32_e64 %0, %1, %0, vcc")
PROBE(p_cnd_sdst, "v_cmp_gt_f32_e64 s[10:11], %1, %0")
PROBE(p_addc, "v_addc_co_u32 %0, vcc, %1, %0, vcc")
PROBE(p_mov, "v_mov_b32 %0, %1")
PROBE(p_max3, "v_max3_f32 %0, %1, %2, %0")
PROBE(p_med3, "v_med3_f32 %0, %1, %2, %0")
PROBE(p_cnd64, "v_cndmask_b32_e64 %0, %1, %0, s[10:11]")
PROBE(p_cnd_b, "v_cndmask_b32 %0, %1, %2, vcc")
PROBE(p_max64, "v_max_f32_e64 %0, %1, %0")
PROBE(p_fmac, "v_fmac_f32 %0, %1, %2")
PROBE(p_fma, "v_fma_f32 %0, %1, %2, %0")
PROBE(p_mul, "v_mul_f32 %0, %1, %0")
PROBE(p_add, "v_add_f32 %0, %1, %0")
PROBE(p_max, "v_max_f32 %0, %1, %0")
PROBE(p_min3, "v_min3_f32 %0, %1, %2, %0")
PROBE(p_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
PROBE(p_and, "v_and_b32 %0, %1, %0")
PROBE(p_addu, "v_add_u32 %0, %1, %0")
PROBE(p_lshladd, "v_lshl_add_u32 %0, %0, 3, %1")
PROBE(p_mullo, "v_mul_lo_u32 %0, %1, %0")
PROBE(p_mulhi, "v_mul_hi_u32 %0, %1, %0")
PROBE(p_mad24, "v_mad_u32_u24 %0, %1, %2, %0")
PROBE(p_cvt, "v_cvt_f32_u32 %0, %0")
PROBE(p_rcp, "v_rcp_f32 %0, %0")
PROBE(p_sqrt, "v_sqrt_f32 %0, %0")
PROBE(p_rsq, "v_rsq_f32 %0, %0")
PROBE(p_cmp, "v_cmp_gt_f32 vcc, %1, %0")
PROBE(p_bfe, "v_bfe_u32 %0, %0, 3, 7")
PROBE(p_alignbit, "v_alignbit_b32 %0, %0, %1, 5")
PROBE(p_divscale, "v_div_scale_f32 %0, vcc, %1, %2, %0")
PROBE(p_divfmas, "v_div_fmas_f32 %0, %1, %2, %0")
PROBE(p_divfixup, "v_div_fixup_f32 %0, %1, %2, %0")
PROBE2(p_pkfma, "v_pk_fma_f32 %0, %1, %2, %0")
PROBE2(p_pkmul, "v_pk_mul_f32 %0, %1, %0")
PROBE2(p_pkadd, "v_pk_add_f32 %0, %1, %0")
PROBE2(p_fma64, "v_fma_f64 %0, %1, %2, %0")
PROBE2(p_mov64, "v_pk_mov_b32 %0, %1, %0")

int main() {
  float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, iters = 4000, blocks = cus * 8;       // 8 blocks of 4 waves per CU: 8 waves per SIMD
  const double clock = pr.clockRate * 1e3;
  printf("%s, %d CUs, %.0f MHz; %d blocks of 256 threads, %d iterations of 16 instructions\n", pr.name, cus, clock / 1e6, blocks, iters);
  struct P { const char* name; void (*k)(float*, int); };
  const P ps[] = {{"v_fma_f32", p_fma}, {"v_mul_f32", p_mul}, {"v_add_f32", p_add}, {"v_max_f32", p_max}, {"v_min3_f32", p_min3},
                  {"v_cndmask_b32", p_cndmask}, {"v_and_b32", p_and}, {"v_add_u32", p_addu}, {"v_lshl_add_u32", p_lshladd},
                  {"v_mul_lo_u32", p_mullo}, {"v_mul_hi_u32", p_mulhi}, {"v_mad_u32_u24", p_mad24}, {"v_cvt_f32_u32", p_cvt},
                  {"v_rcp_f32", p_rcp}, {"v_sqrt_f32", p_sqrt}, {"v_rsq_f32", p_rsq}, {"v_cmp_gt_f32", p_cmp}, {"v_bfe_u32", p_bfe},
                  {"v_alignbit_b32", p_alignbit}, {"v_div_scale_f32", p_divscale}, {"v_div_fmas_f32", p_divfmas},
                  {"v_div_fixup_f32", p_divfixup}, {"v_pk_fma_f32", p_pkfma}, {"v_pk_mul_f32", p_pkmul}, {"v_pk_add_f32", p_pkadd},
                  {"v_fma_f64", p_fma64}, {"v_pk_mov_b32", p_mov64},
                  {"v_max_f32, IEEE bit off", n_max}, {"v_min3_f32, IEEE bit off", n_min3}, {"v_fma_f32, IEEE bit off", n_fma},
                  {"v_cndmask_b32_e64 vcc", p_cnd64v}, {"v_cmp_gt_f32_e64 to sgprs", p_cnd_sdst}, {"v_addc_co_u32 vcc", p_addc},
                  {"v_cndmask_b32_e64 sgpr mask", p_cnd64}, {"v_cndmask (dst not a source)", p_cnd_b}, {"v_max_f32_e64", p_max64}, {"v_fmac_f32", p_fmac},
                  {"chain v_fma_f32", c_fma}, {"chain v_max_f32", c_max}, {"chain v_min3_f32", c_min3}, {"chain v_cndmask_b32", c_cnd},
                  {"chain v_lshl_add_u32", c_lshladd}, {"chain v_and_b32", c_and}, {"chain v_pk_fma_f32", c_pkfma}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const P& p : ps) {
    hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double perSimd = 8.0 * iters * 16.0;            // wave-instructions a SIMD issues
    printf("%-30s %8.3f ms  %6.2f cycles per wave-instruction\n", p.name, ms, ms * 1e-3 * clock / perSimd);
  }
  return 0;
}
