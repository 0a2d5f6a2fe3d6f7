// Micro-benchmark: issue cost of v_pk_fma_f32 vs v_fma_f32 on gfx950 (8 independent chains, 8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ void k(float* out, int iters) {
  const float l = (float)(threadIdx.x & 63);
  if (PK) {
    v2f a[8], b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int i = 0; i < 8; i++) a[i] = (v2f){l + i, l - i};
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = __builtin_elementwise_fma(a[i], b, c);
    float s = 0; for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    float a[8], b = 1.0001f, c = 0.5f;
    for (int i = 0; i < 8; i++) a[i] = l + i;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = fmaf(a[i], b, c);
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}
template <bool PK> void run(float* out, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<PK>, dim3(8192), dim3(256), 0, 0, out, iters); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<PK>, dim3(8192), dim3(256), 0, 0, out, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  double inst = (double)8192 * 4 * iters * 8;
  printf("%s: %.3f ms, %.2f nominal cycles (2.4 GHz) per wave-instruction per SIMD, %.1f TFLOP/s\n", name, ms,
         ms * 1e-3 * 2.4e9 * 1024 / inst, inst * 64 * (PK ? 4 : 2) / (ms * 1e-3) / 1e12);
}
int main() { float* out; (void)hipMalloc(&out, 8192 * 256 * sizeof(float)); run<false>(out, "v_fma_f32   "); run<true>(out, "v_pk_fma_f32"); return 0; }
