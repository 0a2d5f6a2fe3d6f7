// Micro-benchmark: issue cost of the VALU instructions K6 / K7 are made of, on gfx950.
// Every test runs 16 copies of one instruction (independent destination registers) per loop
// iteration, W waves per SIMD; the report is wall cycles per wave-instruction per SIMD at the
// NOMINAL 2.4 GHz (the clock under load is lower) and, more usefully, the cost relative to v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cost.hip -o tools/ubench/valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define DEF_KERNEL(NAME, ASMSTR, CONSTR_OUT, ...)                                        \
  __global__ __launch_bounds__(256) void k_##NAME(float* out, int iters, float seed) {   \
    float a[16];                                                                         \
    const float l = (float)(threadIdx.x & 63) * 1e-3f + seed;                            \
    float b = 1.0001f + l, c = 0.5f + l, d = 0.25f + l;                                  \
    for (int i = 0; i < 16; i++) a[i] = l + i * 0.01f;                                   \
    for (int it = 0; it < iters; it++) {                                                 \
      _Pragma("unroll") for (int i = 0; i < 16; i++)                                     \
        asm volatile(ASMSTR : CONSTR_OUT(a[i]) : __VA_ARGS__);                           \
    }                                                                                    \
    float s = 0; for (int i = 0; i < 16; i++) s += a[i];                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + b + c + d;                          \
  }
#define OUT_RW(x) "+v"(x)

DEF_KERNEL(fma3, "v_fma_f32 %0, %0, %1, %2", OUT_RW, "v"(b), "v"(c))
DEF_KERNEL(fma_samesrc, "v_fma_f32 %0, %0, %0, %1", OUT_RW, "v"(c))
DEF_KERNEL(fmac, "v_fmac_f32_e32 %0, %1, %2", OUT_RW, "v"(b), "v"(c))
DEF_KERNEL(fma_sgpr, "v_fma_f32 %0, %0, %1, %2", OUT_RW, "s"(seed), "v"(c))
DEF_KERNEL(fmamk_lit, "v_fmamk_f32 %0, %1, 0x3dcccccd, %0", OUT_RW, "v"(b))
DEF_KERNEL(fmaak_lit, "v_fmaak_f32 %0, %0, %1, 0x3dcccccd", OUT_RW, "v"(b))
DEF_KERNEL(fma_inline, "v_fma_f32 %0, %0, 0.5, %1", OUT_RW, "v"(b))
DEF_KERNEL(mul_sgpr, "v_mul_f32_e32 %0, %1, %0", OUT_RW, "s"(seed))
DEF_KERNEL(add_sgpr, "v_add_f32_e32 %0, %1, %0", OUT_RW, "s"(seed))
DEF_KERNEL(mul_lit, "v_mul_f32_e32 %0, 0x3dcccccd, %0", OUT_RW, "v"(b))
DEF_KERNEL(cmp_sgprsrc, "v_cmp_lt_f32_e64 s[20:21], %1, %0", OUT_RW, "s"(seed) : "s20", "s21")
DEF_KERNEL(cmp_vcc_lit, "v_cmp_lt_f32_e32 vcc, 0x38d1b717, %0", OUT_RW, "v"(b) : "vcc")
DEF_KERNEL(fma_sgpr2, "v_fma_f32 %0, %1, %2, %0", OUT_RW, "s"(seed), "v"(c))
DEF_KERNEL(mul, "v_mul_f32_e32 %0, %0, %1", OUT_RW, "v"(b))
DEF_KERNEL(add, "v_add_f32_e32 %0, %0, %1", OUT_RW, "v"(b))
DEF_KERNEL(min, "v_min_f32_e32 %0, %0, %1", OUT_RW, "v"(b))
DEF_KERNEL(mov, "v_mov_b32_e32 %0, %1", OUT_RW, "v"(b))
DEF_KERNEL(exp, "v_exp_f32_e32 %0, %0", OUT_RW, "v"(b))
DEF_KERNEL(exp_neg, "v_exp_f32_e64 %0, -%0", OUT_RW, "v"(b))
DEF_KERNEL(rcp, "v_rcp_f32_e32 %0, %0", OUT_RW, "v"(b))
DEF_KERNEL(log, "v_log_f32_e32 %0, %0", OUT_RW, "v"(b))
DEF_KERNEL(sqrt, "v_sqrt_f32_e32 %0, %0", OUT_RW, "v"(b))
DEF_KERNEL(cmp_vcc, "v_cmp_le_f32_e32 vcc, %0, %1", OUT_RW, "v"(b) : "vcc")
DEF_KERNEL(cmp_sgpr, "v_cmp_lt_f32_e64 s[20:21], %0, %1", OUT_RW, "v"(b) : "s20", "s21")
DEF_KERNEL(cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", OUT_RW, "v"(b) : "s20", "s21")
DEF_KERNEL(pl16, "v_permlane16_swap_b32_e32 %0, %1", OUT_RW, "v"(b))
DEF_KERNEL(pl32, "v_permlane32_swap_b32_e32 %0, %1", OUT_RW, "v"(b))
DEF_KERNEL(dpp_add, "v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", OUT_RW, "v"(b))
DEF_KERNEL(fma_exp_mix, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32_e32 %0, %0", OUT_RW, "v"(b), "v"(c))

// dependent chain: every instruction consumes the previous result (latency with ONE wave per SIMD)
__global__ __launch_bounds__(64) void k_dep_fma(float* out, int iters, float seed) {
  float a = seed, b = 1.0001f, c = 0.5f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  }
  out[blockIdx.x * 64 + threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void k_dep_exp(float* out, int iters, float seed) {
  float a = seed;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a));
  }
  out[blockIdx.x * 64 + threadIdx.x] = a;
}

typedef void (*kern_t)(float*, int, float);
struct Test { const char* name; kern_t k; int per_iter; };

static double run(kern_t k, float* out, int blocks, int threads, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5f);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); exit(1); }
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {     // best of three: the first launches run while the clock still ramps
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}

int main(int argc, char** argv) {
  float* out; (void)hipMalloc(&out, 1 << 26);
  const int iters = 60000;
#define T(NAME, N) {#NAME, k_##NAME, N}
  Test tests[] = {T(fma3, 16), T(fma_samesrc, 16), T(fmac, 16), T(fma_sgpr, 16), T(fmamk_lit, 16), T(fmaak_lit, 16), T(fma_inline, 16), T(mul_sgpr, 16), T(add_sgpr, 16), T(mul_lit, 16), T(cmp_sgprsrc, 16), T(cmp_vcc_lit, 16), T(fma_sgpr2, 16), T(mul, 16), T(add, 16), T(min, 16), T(mov, 16),
                  T(exp, 16), T(exp_neg, 16), T(rcp, 16), T(log, 16), T(sqrt, 16), T(cmp_vcc, 16), T(cmp_sgpr, 16),
                  T(cnd_sgpr, 16), T(pl16, 16), T(pl32, 16),
                  T(dpp_add, 16), T(fma_exp_mix, 64),};
  // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD; B blocks per CU resident -> B waves per SIMD
  for (int wps : {4}) {
    printf("--- %d wave(s) per SIMD (256 CUs x %d blocks of 256 threads, one round) ---\n", wps, wps);
    double base = 0;
    for (auto& t : tests) {
      const int blocks = 256 * wps;   // one resident round: every CU gets `wps` blocks
      const double ms = run(t.k, out, blocks, 256, iters);
      const double inst_per_simd = (double)wps * iters * t.per_iter;   // wave-instructions each SIMD executes
      const double cyc = ms * 1e-3 * 2.4e9 / inst_per_simd;
      if (!strcmp(t.name, "fma3")) base = cyc;
      fflush(stdout);
      printf("%-14s %8.3f ms  %6.2f nominal cyc / wave-instr / SIMD   x%.2f of v_fma\n", t.name, ms, cyc, cyc / base);
    }
  }
  {
    const double ms = run(k_dep_fma, out, 1024, 64, iters);
    printf("dependent v_fma chain, 1 wave/SIMD: %.2f nominal cyc per instr\n", ms * 1e-3 * 2.4e9 / (iters * 16.0));
    const double ms2 = run(k_dep_exp, out, 1024, 64, iters);
    printf("dependent v_exp chain, 1 wave/SIMD: %.2f nominal cyc per instr\n", ms2 * 1e-3 * 2.4e9 / (iters * 16.0));
  }
  return 0;
}
