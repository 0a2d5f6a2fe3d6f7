// contrib_probe.hip -- measurement aid (not part of libtgs_hip.so): how much of the work the
// compositing kernels evaluate actually contributes.
// One workgroup per tile, wave k = 8x8 quadrant k with the lane -> pixel mapping of K6/K7.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k_probe(int W, int H, int TW, float pix_center,
                                               const float* __restrict__ splats,
                                               const int32_t* __restrict__ sorted_gid,
                                               const int32_t* __restrict__ tile_start,
                                               unsigned long long* __restrict__ ctr) {
  const int tile = blockIdx.x;
  const int ty = tile / TW, tx = tile - ty * TW;
  const int k = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int px = tx * 16 + 8 * (k & 1) + (l & 7), py = ty * 16 + 8 * (k >> 1) + (l >> 3);
  const bool inside = px < W && py < H;
  const float fx = px + pix_center, fy = py + pix_center;
  float T = 1.f;
  bool alive = inside;
  const int s = tile_start[tile], e = tile_start[tile + 1];
  unsigned long long c_quad = 0, c_contrib = 0, c_dead = 0, c_blk = 0, c_row = 0, c_quad_any = 0, c_reach = 0;
  __shared__ int anyflag[2][4];
  __shared__ unsigned long long tot[8];
  if (threadIdx.x < 8) tot[threadIdx.x] = 0;
  unsigned long long c_pair_any = 0, c_after = 0, c_pair_reach = 0;
  for (int i = s; i < e; i++) {
    const float* r = splats + (size_t)sorted_gid[i] * 12;
    // record slots 0, 1 are relative to the origin of the Gaussian's tile rect (include/tgs.h)
    const unsigned rect = __float_as_uint(r[10]);
    const float dx = r[0] + 16.f * (float)(rect & 255u) - fx, dy = r[1] + 16.f * (float)((rect >> 8) & 255u) - fy;
    const float sig = 0.5f * (r[4] * dx * dx + r[6] * dy * dy) + r[5] * dx * dy;
    const float al = fminf(0.999f, r[3] * __expf(-sig));
    const bool reach = inside && sig >= 0.f && al >= 1.f / 255.f;
    bool go = false;
    if (reach && alive) {
      const float Tn = T * (1.f - al);
      if (Tn <= 1e-4f) alive = false; else { T = Tn; go = true; }
    }
    const unsigned long long b_reach = __ballot(reach), b_go = __ballot(go), b_alive = __ballot(alive || go);
    if (l == 0) {
      const bool quad_live = b_alive != 0ull;
      if (b_reach != 0ull && quad_live) c_quad++;
      if (b_go != 0ull) c_quad_any++;
      c_contrib += __popcll(b_go);
      c_reach += __popcll(b_reach);
      if (quad_live) c_dead += __popcll(b_reach & ~b_go);
      // 4x4 blocks of the quadrant: lanes with (l&7)>>2 == bx, (l>>5) == by
      for (int by = 0; by < 2; by++)
        for (int bx = 0; bx < 2; bx++) {
          unsigned long long m = 0;
          for (int yy = 0; yy < 4; yy++) m |= (0xFull << (bx * 4)) << ((by * 4 + yy) * 8);
          if (b_go & m) c_blk++;
        }
      for (int yy = 0; yy < 8; yy++) if (b_go & (0xFFull << (yy * 8))) c_row++;
      anyflag[i & 1][k] = (b_go != 0ull) | ((b_alive != 0ull) << 1) | ((b_reach != 0ull) << 2);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int a = 0;
      for (int q = 0; q < 4; q++) a |= anyflag[i & 1][q];
      if (a & 1) c_pair_any++;
      if (!(a & 2)) c_after++;
      if (a & 4) c_pair_reach++;   // alpha >= 1/255 at some pixel centre of the tile (alive or not)
    }
  }
  if (l == 0) {
    atomicAdd(&tot[1], c_quad); atomicAdd(&tot[2], c_contrib); atomicAdd(&tot[3], c_dead);
    atomicAdd(&tot[4], c_blk); atomicAdd(&tot[7], c_row);
  }
  __syncthreads();
  if (l == 0) { atomicAdd(&ctr[8], c_quad_any); atomicAdd(&ctr[9], c_reach); }
  if (threadIdx.x == 0) {
    atomicAdd(&ctr[0], (unsigned long long)(e - s));
    for (int j = 1; j < 8; j++) if (j != 5 && j != 6) atomicAdd(&ctr[j], tot[j]);
    atomicAdd(&ctr[5], c_pair_any); atomicAdd(&ctr[6], c_after); atomicAdd(&ctr[10], c_pair_reach);
  }
}

extern "C" int probe_contrib(int W, int H, float pix_center, const float* splats, const int32_t* sorted_gid,
                             const int32_t* tile_start, unsigned long long* ctr, void* stream) {
  const int TW = (W + 15) / 16, TH = (H + 15) / 16;
  hipLaunchKernelGGL(k_probe, dim3(TW * TH), dim3(256), 0, (hipStream_t)stream, W, H, TW, pix_center, splats,
                     sorted_gid, tile_start, ctr);
  return (int)hipGetLastError();
}
