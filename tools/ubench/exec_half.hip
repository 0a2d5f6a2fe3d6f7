// Micro-benchmark: does a wave64 VALU instruction cost less when one 32-lane half has EXEC = 0?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out, int active_lanes, int iters) {
  const int lane = threadIdx.x & 63;
  float a = (float)lane, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f, f = 2.f, g = 3.f, h = 4.f;
  if (lane < active_lanes) {
    for (int i = 0; i < iters; i++) {
      a = fmaf(a, b, c); d = fmaf(d, b, c); e = fmaf(e, b, c); f = fmaf(f, b, c);
      g = fmaf(g, b, c); h = fmaf(h, b, c); a = fmaf(a, b, d); e = fmaf(e, b, f);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 8 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int act : {64, 48, 32, 16, 1}) {
    hipLaunchKernelGGL(k, dim3(1024 * 8), dim3(256), 0, 0, out, act, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1024 * 8), dim3(256), 0, 0, out, act, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)1024 * 8 * 4 * iters * 8;   // wave-instructions
    printf("active lanes %2d: %.3f ms  -> %.2f cycles/wave-instr/SIMD at 2.4 GHz\n", act, ms,
           ms * 1e-3 * 2.4e9 * 1024 / inst);
  }
  return 0;
}
