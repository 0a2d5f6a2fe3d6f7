// Developer probe: the DPP prefix scans of k_raster_bwd_scan on known data.  hipcc --offload-arch=gfx950 -O2 tools/probe/dpp_scan_test.hip -o /tmp/dpp_scan && /tmp/dpp_scan
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float scan_incl_add(float v) {
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
  return v;
}
__device__ __forceinline__ float lane_shr1_zero(float v) {
  float t;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(t) : "v"(v));
  return t;
}
__global__ void k(float* o) {
  const int l = threadIdx.x;
  float v = 1.0f;
  const float s = scan_incl_add(v);
  o[l] = s;
  o[64 + l] = lane_shr1_zero(s);
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) { if (h[l] != l + 1) bad++; if (h[64 + l] != l) bad++; }
  for (int l = 0; l < 64; l++) printf("%g ", h[l]); printf("\n");
  for (int l = 0; l < 64; l++) printf("%g ", h[64 + l]); printf("\nbad %d\n", bad);
  return bad != 0;
}
