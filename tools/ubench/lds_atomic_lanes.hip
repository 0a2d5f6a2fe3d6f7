// Micro-benchmark: cost of the LDS float atomic ds_add_f32 on gfx950 as a function of the
// number of ACTIVE lanes (1 / 4 / 16 / 64) and of the address pattern (every active lane its own address, or all
// active lanes the same address), next to ds_add_u32 and a plain ds_write_b32.  The block-form backward would issue
// its per-Gaussian accumulation with 4 active lanes (one per DPP row), so the 64-lane figure of lds_atomic.hip
// (194 cycles per wave instruction) is not the one that decides it.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic_lanes.hip -o tools/ubench/lds_atomic_lanes
//   disassembly (committed: profiles/r4_lds_atomic_lanes_disasm.txt):
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/ubench/lds_atomic_lanes.hip -o - | grep -E "ds_|s_cbranch"
#include <hip/hip_runtime.h>
#include <stdio.h>

// OP: 0 = ds_add_f32, 1 = ds_add_u32, 2 = ds_write_b32;   lanes with (lane & MASKSEL) == 0 are active:
// STRIDE = 1 (64 lanes), 4 (16 lanes: 0,4,8..), 16 (4 lanes: 0,16,32,48), 64 (1 lane);  SAME: all active lanes hit one address
template <int OP, int STRIDE, int SAME>
__global__ __launch_bounds__(256) void k(int iters, float* fout) {
  __shared__ float hf[2048];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int i = t; i < 2048; i += 256) hf[i] = 0.f;
  __syncthreads();
  const bool act = (lane & (STRIDE - 1)) == 0;
  // each wave works on its own 512-float region; distinct addresses are spread over all 64 banks
  const int base = w * 512 + (SAME ? 0 : lane);
  if (act) {
#pragma unroll 8
    for (int i = 0; i < iters; i++) {
      const int a = base + ((i & 7) << 6);
      if (OP == 0) __hip_atomic_fetch_add(&hf[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 1) __hip_atomic_fetch_add((int*)&hf[a], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == 2) ((volatile float*)hf)[a] = (float)i;
    }
  }
  __syncthreads();
  fout[blockIdx.x * 256 + t] = hf[t] + hf[t + 256];
}

template <int OP, int STRIDE, int SAME>
void run(const char* name) {
  float* f;
  hipMalloc(&f, 1024 * 256 * 4);
  const int iters = 4096, blocks = 1024;
  hipLaunchKernelGGL((k<OP, STRIDE, SAME>), dim3(blocks), dim3(256), 0, 0, 16, f);
  (void)hipDeviceSynchronize();
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<OP, STRIDE, SAME>), dim3(blocks), dim3(256), 0, 0, iters, f);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const double winstr = (double)blocks * 4 * iters;
  printf("%-14s active lanes %2d  %-9s %8.3f ms  %7.1f cycles per wave-instruction per CU (2.4 GHz nominal, 256 CUs, 4 waves/CU x 4 blocks)\n",
         name, 64 / STRIDE, SAME ? "same addr" : "distinct", ms, ms * 1e-3 * 2.4e9 * 256 / winstr);
  (void)hipFree(f);
}

int main() {
  run<0, 1, 0>("ds_add_f32");  run<0, 4, 0>("ds_add_f32");  run<0, 16, 0>("ds_add_f32");  run<0, 64, 0>("ds_add_f32");
  run<0, 1, 1>("ds_add_f32");  run<0, 4, 1>("ds_add_f32");  run<0, 16, 1>("ds_add_f32");
  run<1, 1, 0>("ds_add_u32");  run<1, 16, 0>("ds_add_u32"); run<1, 1, 1>("ds_add_u32");  run<1, 16, 1>("ds_add_u32");
  run<2, 1, 0>("ds_write_b32"); run<2, 16, 0>("ds_write_b32");
  return 0;
}
