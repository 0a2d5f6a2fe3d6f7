// What is the shader clock while every SIMD issues VALU instructions back to back?  Each wave reads s_memtime
// (core-clock counter) and s_memrealtime (constant 100 MHz) around a long stream of independent v_fma_f32; the
// ratio is the clock the stream ran at, and cycles / instruction follows without assuming the nominal 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clock_probe.hip -o tools/ubench/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(unsigned long long* out, int iters, float seed) {
  float a[16];
  const float l = (float)(threadIdx.x & 63) * 1e-3f + seed;
  float b = 1.0001f + l, c = 0.5f + l;
  for (int i = 0; i < 16; i++) a[i] = l + i * 0.01f;
  const unsigned long long t0 = __builtin_readcyclecounter();        // s_memtime
  const unsigned long long r0 = wall_clock64();                      // s_memrealtime
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (MODE == 1) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a[i]));
      if (MODE == 2) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float s = 0; for (int i = 0; i < 16; i++) s += a[i];
  if ((threadIdx.x & 63) == 0) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    out[3 * w] = t1 - t0; out[3 * w + 1] = r1 - r0; out[3 * w + 2] = (unsigned long long)(s != 12345.f);
  }
}

template <int MODE>
static void run(const char* name, int wps, int iters, unsigned long long* dout, unsigned long long* hout) {
  const int blocks = 256 * wps, nw = blocks * 4;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters, 0.5f);   // warm-up
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(hout, dout, sizeof(unsigned long long) * 3 * nw, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int w = 0; w < nw; w++) { cyc += (double)hout[3 * w]; rt += (double)hout[3 * w + 1]; }
  cyc /= nw; rt /= nw;
  const double per_wave_instr = (double)iters * 16;
  printf("%-8s %d waves/SIMD: wall %.3f ms | per wave: s_memtime %.0f ticks, s_memrealtime %.0f ticks (= %.3f ms at 100 MHz)"
         " -> s_memtime runs at %.1f MHz | %.2f s_memtime ticks per instruction of ONE wave, %.2f per SIMD-instruction\n",
         name, wps, ms, cyc, rt, rt / 1e5, cyc / (rt / 100.0), cyc / per_wave_instr, cyc / per_wave_instr / wps);
}

int main() {
  int wall_khz = 0, clk_khz = 0;
  (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("hipDeviceAttributeWallClockRate %d kHz, hipDeviceAttributeClockRate %d kHz\n", wall_khz, clk_khz);
  unsigned long long* dout; (void)hipMalloc(&dout, sizeof(unsigned long long) * 3 * 256 * 8 * 4);
  unsigned long long* hout = (unsigned long long*)malloc(sizeof(unsigned long long) * 3 * 256 * 8 * 4);
  run<0>("fma", 4, 400000, dout, hout);      // ~40 ms: the clock governor has settled, launch overheads are negligible
  run<0>("fma", 4, 400000, dout, hout);
  for (int wps : {1, 2, 4, 8}) run<0>("fma", wps, 200000, dout, hout);
  for (int wps : {4}) run<1>("exp", wps, 200000, dout, hout);
  for (int wps : {4}) run<2>("mov", wps, 200000, dout, hout);
  return 0;
}
