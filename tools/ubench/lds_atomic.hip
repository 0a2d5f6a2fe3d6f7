// Micro-benchmark: cost of LDS atomics on gfx950 (wave64, distinct address per lane, no conflicts).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic.hip -o tools/ubench/lds_atomic && tools/ubench/lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, int* out, float* fout) {
  __shared__ int hi[1024];
  __shared__ float hf[1024];
  const int t = threadIdx.x;
  hi[t] = 0; hf[t] = 0.f; hi[t + 256] = 0; hf[t + 256] = 0.f;
  __syncthreads();
  int acc = 0;
  for (int i = 0; i < iters; i++) {
    const int a = (t + 17 * i) & 1023;
    if (MODE == 0) acc += atomicAdd(&hi[a], 1);                      // returning int
    if (MODE == 1) atomicAdd(&hi[a], 1);                              // non-returning int
    if (MODE == 2) __hip_atomic_fetch_add(&hf[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // float
    if (MODE == 3) { hi[a] = i; acc += hi[(a + 64) & 1023]; }        // plain write + read for reference
  }
  __syncthreads();
  out[blockIdx.x * 256 + t] = acc + hi[t];
  fout[blockIdx.x * 256 + t] = hf[t];
}
template <int MODE>
void run(const char* name) {
  int* o; float* f;
  hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&f, 1024 * 256 * 4);
  const int iters = 4096, blocks = 1024;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, 16, o, f);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, iters, o, f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // wave-instructions: blocks * 4 waves * iters; 256 CUs
  const double winstr = (double)blocks * 4 * iters;
  printf("%-28s %8.3f ms  %6.1f cycles per wave-instruction per CU (2.4 GHz, 256 CUs)\n", name, ms,
         ms * 1e-3 * 2.4e9 * 256 / winstr);
}
int main() {
  run<0>("ds_add_rtn_u32");
  run<1>("ds_add_u32 (no return)");
  run<2>("ds_add_f32 (no return)");
  run<3>("ds_write_b32 + ds_read_b32");
  return 0;
}
