// imgloss.hip -- K10 SSIM forward + gradient image (11x11 Gaussian window, sigma 1.5, zero
// padding), the (1-SSIM) term of the Splatfacto-style loss (1-l)*L1 + l*(1-SSIM) that the
// reference's `depth-gaussian-splatting` method trains with.  gfx950.
//
// Two separable-convolution kernels, row streaming:
//   k_ssim_fwd : 5 windowed moments -> SSIM map; writes the three adjoint maps
//                A = dm/dmu1 (total), B = dm/dE[x^2], C = dm/dE[xy] and a per-workgroup sum of the map
//   k_ssim_bwd : v_img = weight * (G*A + 2*img*(G*B) + gt*(G*C))
// A workgroup owns a strip of 64 columns x SEG rows, thread = (column, channel).  Input rows are
// staged 11 at a time in LDS exactly as they lie in memory (interleaved channels: coalesced loads,
// conflict-free ds_read with lane stride 1 / 3); every thread filters its row horizontally from LDS
// (11 taps) and keeps the last 11 horizontally-filtered rows in REGISTERS, so the vertical pass
// costs no LDS traffic and no barrier -- the first version staged 26x26 halo tiles per 16x16
// outputs (2.6x read amplification, 48 KB of LDS per workgroup, 3 workgroups per CU, 39-44 % LDS
// bank-conflict cycles; 77 + 80 us at 1080p).  Read amplification here: 74/64 horizontally,
// (SEG+10)/SEG vertically.  Roofline: HBM (9 adjoint planes written, then read).
#include <math.h>
#include <stdlib.h>
#include "tgs_common.h"

namespace {

constexpr int WIN = 11, HALO = 5;
struct Win { float g[WIN]; };
// The window weights as compile-time constants (11-tap Gaussian, sigma 1.5, normalised in double and rounded
// to fp32 -- the values the host used to pass as a kernel argument).  As kernel arguments they sat in SGPRs,
// and on gfx950 a VALU instruction with an SGPR source is the most expensive form there is (v_mul / v_add
// 5.45 cycles, v_fma 4.5 - 5.5, against 2.8 - 3.0 with a literal: tools/ubench/valu_cost.hip) -- every FMA
// of the two filters is of that kind.  TGS_SSIM_SGPR_WINDOW restores the argument form for A/B runs.
#ifdef TGS_SSIM_SGPR_WINDOW
#define WK(k) win.g[k]
#else
__device__ constexpr float WGT[WIN] = {0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c4p-3f,
                                       0x1.10656p-2f, 0x1.b43c4p-3f, 0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f,
                                       0x1.0d956cp-10f};
#define WK(k) WGT[k]
#endif

// TGS_SSIM_DMA (bit 0: k_ssim_fwd, bit 1: k_ssim_bwd): the staged rows are brought in by LDS-DMA (`buffer_load_dword ... lds`: no
// VGPRs, zero fill outside the image by the buffer's own range check) into one of TWO LDS images, and the next block's
// rows are requested before the current block is filtered -- a request stays in flight across the barrier, so the
// load -> barrier -> filter chain of a workgroup becomes max(load, filter) per block.
#ifndef TGS_SSIM_DMA
#define TGS_SSIM_DMA 0
#endif
constexpr int SW = 64;              // strip width (output columns per workgroup)
constexpr int NTH = 3 * SW;         // threads: (column, channel)
constexpr int RB = WIN;             // input rows per staged block = window height (phase p = row in block)
// blocks per segment (nblk, a launch argument): SEG = 11 nblk - 10 output rows per workgroup.  4 (34 rows)
// at 1080p and above; small images take shorter segments -- the kernels are latency chains over the
// blocks of a segment, and an 800x800 image has only 312 segments of 34 rows for 256 CUs.
__host__ __device__ constexpr int seg_rows(int nblk) { return RB * nblk - 2 * HALO; }

// Stages input rows [r0, r0 + RB) x columns [x0 - 5, x0 + SW + 5) of an interleaved [H, W, CH]
// image into LDS (zero outside the image), in batches of 8 loads in flight per thread.
template <int CH>
struct RowBlock {
  static constexpr int ROWF = (SW + 2 * HALO) * CH;             // floats per staged row
  static constexpr int NIT = (RB * ROWF + NTH - 1) / NTH;
  template <int B = 8>
  static __device__ __forceinline__ void stage(const float* __restrict__ src, float* __restrict__ lds,
                                               int W, int H, int x0, int r0, int tid) {
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += B) {
      float v[B];
#pragma unroll
      for (int u = 0; u < B; u++) {
        const int i = tid + NTH * (it0 + u);
        const int j = i / ROWF, e = i - j * ROWF;
        const int gy = r0 + j, gxf = (x0 - HALO) * CH + e;
        v[u] = 0.f;
        if (i < RB * ROWF && gy >= 0 && gy < H && gxf >= 0 && gxf < W * CH) v[u] = src[(size_t)gy * W * CH + gxf];
      }
#pragma unroll
      for (int u = 0; u < B; u++) {
        const int i = tid + NTH * (it0 + u);
        if (i < RB * ROWF) lds[i] = v[u];
      }
    }
  }
  // LDS-DMA staging: wave w of the workgroup fills floats [64 w + NTH u, + 64) of the image in iteration u (one
  // `buffer_load_dword ... lds` per wave and iteration: 64 lanes x 4 B land contiguously at M0 + 4 lane); an element
  // outside the image gets an offset beyond the buffer and reads as zero.  PADF: the image rounded up to whole waves.
  static constexpr int PADF = (RB * ROWF + TGS_WAVE - 1) / TGS_WAVE * TGS_WAVE;
  static constexpr int NDMA = PADF / TGS_WAVE;                       // wave-chunks per staged block
  static __device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t rsrc, float* __restrict__ lds, int W, int H, int x0, int r0, int tid) {
    typedef __attribute__((address_space(3))) float lds_float;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int NW = NTH / TGS_WAVE;
    const int gx0 = (x0 - HALO) * CH, WC = W * CH;
#pragma unroll
    for (int u = 0; u * NW < NDMA; u++) {
      const int chunk = u * NW + wave;                               // wave-uniform
      if (chunk < NDMA) {
        const int i = chunk * TGS_WAVE + lane;
        const int j = i / ROWF, e = i - j * ROWF;                    // (constant divisor: a multiply)
        const int gy = r0 + j, gxf = gx0 + e;
        const bool in = i < RB * ROWF && (unsigned)gy < (unsigned)H && (unsigned)gxf < (unsigned)WC;
        const unsigned off = in ? (unsigned)(gy * WC + gxf) * 4u : 0xffffffffu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_float*)(lds + chunk * TGS_WAVE), 4, off, 0, 0, 0);
      }
    }
  }
  // Two images of the same geometry in one pass: the loads of both are in flight together, which halves
  // the number of load -> wait round trips per block (the forward kernel is a chain of them).
  static __device__ __forceinline__ void stage2(const float* __restrict__ src_a, const float* __restrict__ src_b,
                                                float* __restrict__ lds_a, float* __restrict__ lds_b,
                                                int W, int H, int x0, int r0, int tid) {
#ifndef TGS_SSIM_FWD_B
#define TGS_SSIM_FWD_B 13   // all 13 + 13 loads of a block in flight at once (round 5: 128 VGPRs, still 4 waves per SIMD); 7 = two rounds
#endif
    constexpr int B = TGS_SSIM_FWD_B;
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += B) {
      float va[B], vb[B];
#pragma unroll
      for (int u = 0; u < B; u++) {
        const int i = tid + NTH * (it0 + u);
        const int j = i / ROWF, e = i - j * ROWF;
        const int gy = r0 + j, gxf = (x0 - HALO) * CH + e;
        va[u] = 0.f; vb[u] = 0.f;
        if (it0 + u < NIT && i < RB * ROWF && gy >= 0 && gy < H && gxf >= 0 && gxf < W * CH) {
          const size_t o = (size_t)gy * W * CH + gxf;
          va[u] = src_a[o]; vb[u] = src_b[o];
        }
      }
#pragma unroll
      for (int u = 0; u < B; u++) {
        const int i = tid + NTH * (it0 + u);
        if (it0 + u < NIT && i < RB * ROWF) { lds_a[i] = va[u]; lds_b[i] = vb[u]; }
      }
    }
  }
};

__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_ssim_fwd(int W, int H, Win win,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ gt,
                                                  float* __restrict__ adj /*[H,W,3,3] or null*/,
                                                  float* __restrict__ block_partials, int n_partials, int NBLK,
                                                  int y_lo, int y_hi, int c_lo, int c_hi) {
  // rows [y_lo, y_hi) of the map / adjoint maps are produced (the whole image, or one band of it plus the
  // 5-row halo its backward pass needs); rows [c_lo, c_hi) count towards the sum
  const int SEG = seg_rows(NBLK);
  constexpr int ROWF = RowBlock<3>::ROWF;
#if TGS_SSIM_DMA & 1
  __shared__ float sa2[2][RowBlock<3>::PADF], sb2[2][RowBlock<3>::PADF];
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, (int)((size_t)W * H * 12), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)gt, 0, (int)((size_t)W * H * 12), 0x00020000);
#else
  __shared__ float sa[RB * ROWF], sb[RB * ROWF];
#endif
  __shared__ float red[NTH / TGS_WAVE];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * SW, ys = y_lo + blockIdx.y * SEG;
  const int gx = x0 + tid / 3, c = tid - (tid / 3) * 3;
  const int ye = min(ys + SEG, y_hi);
  float w[RB][5];
#pragma unroll
  for (int p = 0; p < RB; p++)
#pragma unroll
    for (int q = 0; q < 5; q++) w[p][q] = 0.f;
  float msum = 0.f;
#if TGS_SSIM_DMA & 1
  RowBlock<3>::dma(rs_a, sa2[0], W, H, x0, ys - HALO, tid);
  RowBlock<3>::dma(rs_b, sb2[0], W, H, x0, ys - HALO, tid);
#endif
  for (int blk = 0; blk < NBLK; blk++) {
    const int r0 = ys - HALO + blk * RB;          // first input row of this block
    if (r0 - HALO >= ye) break;                   // no output row left (uniform)
#if TGS_SSIM_DMA & 1
    // this block's rows have landed (own requests: vmcnt; the other waves': the barrier) -- and every wave is done with
    // the OTHER image, which the next block's requests overwrite while this block is filtered
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (blk + 1 < NBLK && r0 + RB - HALO < ye) {
      RowBlock<3>::dma(rs_a, sa2[(blk + 1) & 1], W, H, x0, r0 + RB, tid);
      RowBlock<3>::dma(rs_b, sb2[(blk + 1) & 1], W, H, x0, r0 + RB, tid);
    }
    const float* sa = sa2[blk & 1];
    const float* sb = sb2[blk & 1];
#else
    __syncthreads();                              // previous block fully consumed
    {
      // no register prefetch of the next block: the window already holds 55 VGPRs per thread and
      // the other resident workgroups of the CU cover the load latency
      // both images in one pass, NIT = 13: two rounds of 7 + 7 loads in flight (staging them one after
      // the other in rounds of 8: forward + backward 118 -> 111 us at 1080p, same box)
      RowBlock<3>::stage2(img, gt, sa, sb, W, H, x0, r0, tid);
    }
    __syncthreads();
#endif
#pragma unroll
    for (int p = 0; p < RB; p++) {
      // horizontal pass of input row r0 + p for (column, channel) = this thread
      const float* ra = sa + p * ROWF + tid;
      const float* rb = sb + p * ROWF + tid;
      float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; k++) {
        const float a = ra[3 * k], b = rb[3 * k];
        const float ga = WK(k) * a, gb = WK(k) * b;
        m1 += ga; m2 += gb;
        e11 = fmaf(ga, a, e11); e22 = fmaf(gb, b, e22); e12 = fmaf(ga, b, e12);
      }
      w[p][0] = m1; w[p][1] = m2; w[p][2] = e11; w[p][3] = e22; w[p][4] = e12;
      // vertical pass: output row = r0 + p - 5; window slots (p+1 .. p+11) mod 11 = oldest .. newest
      const int gy = r0 + p - HALO;
      if (gy >= ys && gy < ye) {                  // uniform
        float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < WIN; k++) {
#pragma unroll
          for (int q = 0; q < 5; q++) o[q] = fmaf(WK(k), w[(p + 1 + k) % RB][q], o[q]);
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1 = o[0], mu2 = o[1];
        const float s11 = o[2] - mu1 * mu1, s22 = o[3] - mu2 * mu2, s12 = o[4] - mu1 * mu2;
        const float n1 = 2.f * mu1 * mu2 + C1, n2 = 2.f * s12 + C2;
        const float d1 = mu1 * mu1 + mu2 * mu2 + C1, d2 = s11 + s22 + C2;
        // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: d1, d2 >= C1, C2 > 0
        const float id1 = __builtin_amdgcn_rcpf(d1), id2 = __builtin_amdgcn_rcpf(d2);
        const float m = n1 * n2 * id1 * id2;
        if (gx < W) {
          if (gy >= c_lo && gy < c_hi) msum += m;
          if (adj) {
            const float dm_ds12 = 2.f * n1 * id1 * id2;
            const float dm_ds11 = -m * id2;
            const float dm_dmu1 = 2.f * mu2 * n2 * id1 * id2 - m * 2.f * mu1 * id1;
            float* o3 = adj + (((size_t)gy * W + gx) * 3 + c) * 3;
            o3[0] = dm_dmu1 - 2.f * mu1 * dm_ds11 - mu2 * dm_ds12;
            o3[1] = dm_ds11;
            o3[2] = dm_ds12;
          }
        }
      }
    }
  }
  const float tot = wave_sum(msum);
  if ((tid & 63) == 0) red[tid >> 6] = tot;
  __syncthreads();
  const int wg = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
  if (tid == 0) block_partials[wg] = red[0] + red[1] + red[2];
  // the caller's buffer has one entry per 16x16 tile: the entries beyond the workgroup count are zero
  for (int i = nwg + wg * NTH + tid; i < n_partials; i += nwg * NTH) block_partials[i] = 0.f;
}

__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_ssim_bwd(int W, int H, Win win, float weight,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ gt,
                                                  const float* __restrict__ adj,
                                                  float* __restrict__ v_img, int NBLK, int y_lo, int y_hi) {
  const int SEG = seg_rows(NBLK);
  constexpr int ROWF = RowBlock<9>::ROWF;
#if TGS_SSIM_DMA & 2
  __shared__ float sadj2[2][RowBlock<9>::PADF];
  const __amdgpu_buffer_rsrc_t rs_j = __builtin_amdgcn_make_buffer_rsrc((void*)adj, 0, (int)((size_t)W * H * 36), 0x00020000);
#else
  __shared__ float sadj[RB * ROWF];
#endif
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * SW, ys = y_lo + blockIdx.y * SEG;
  const int gx = x0 + tid / 3;
  const int ye = min(ys + SEG, y_hi);
  float w[RB][3];
#pragma unroll
  for (int p = 0; p < RB; p++) { w[p][0] = 0.f; w[p][1] = 0.f; w[p][2] = 0.f; }
#if TGS_SSIM_DMA & 2
  RowBlock<9>::dma(rs_j, sadj2[0], W, H, x0, ys - HALO, tid);
#endif
  for (int blk = 0; blk < NBLK; blk++) {
    const int r0 = ys - HALO + blk * RB;
    if (r0 - HALO >= ye) break;
#if TGS_SSIM_DMA & 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (blk + 1 < NBLK && r0 + RB - HALO < ye) RowBlock<9>::dma(rs_j, sadj2[(blk + 1) & 1], W, H, x0, r0 + RB, tid);
    const float* sadj = sadj2[blk & 1];
#else
    __syncthreads();
    {
      // 38 loads per thread and block: three rounds of 13 in flight (the kernel has registers to spare)
#ifndef TGS_SSIM_BWD_B
#define TGS_SSIM_BWD_B 20   // two rounds of 20 instead of three of 13 (66 VGPRs); all 39 at once spills.  Both: 99.4 -> 97.7 us (r5_ab_runs.txt)
#endif
      RowBlock<9>::stage<TGS_SSIM_BWD_B>(adj, sadj, W, H, x0, r0, tid);
    }
    __syncthreads();
#endif
#pragma unroll
    for (int p = 0; p < RB; p++) {
      const float* r = sadj + p * ROWF + 3 * tid;     // (pixel, channel) -> its three adjoint maps
      float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; k++) {
        h0 = fmaf(WK(k), r[9 * k], h0);
        h1 = fmaf(WK(k), r[9 * k + 1], h1);
        h2 = fmaf(WK(k), r[9 * k + 2], h2);
      }
      w[p][0] = h0; w[p][1] = h1; w[p][2] = h2;
      const int gy = r0 + p - HALO;
      if (gy >= ys && gy < ye) {
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; k++) {
          o0 = fmaf(WK(k), w[(p + 1 + k) % RB][0], o0);
          o1 = fmaf(WK(k), w[(p + 1 + k) % RB][1], o1);
          o2 = fmaf(WK(k), w[(p + 1 + k) % RB][2], o2);
        }
        if (gx < W) {
          const size_t pidx = (size_t)gy * W * 3 + (size_t)x0 * 3 + tid;
          v_img[pidx] = weight * (o0 + 2.f * img[pidx] * o1 + gt[pidx] * o2);
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// One pass (round 5): forward + backward of a strip without the adjoint planes in HBM
// ---------------------------------------------------------------------------------------------
// k_ssim_fwd writes nine adjoint values per pixel (75 MB at 1080p) that k_ssim_bwd reads back with its own halo
// (x 1.5): 351 MB of HBM traffic for 75 MB of compulsory bytes (img, gt in; v_img out), 10 % of the train step.
// Here a workgroup owns SWO = 54 output columns x SEG rows and keeps everything between the two filters on chip:
// its 192 threads = (64 adjoint columns [xo0 - 5, xo0 + 59)) x 3 channels.  Per block of 11 staged input rows
//   phase 1  = k_ssim_fwd's loop verbatim on the strip shifted by the halo: horizontal moments from LDS, the last 11 rows
//              in registers, SSIM map + the three adjoint values of (row r - 5, own column) -> LDS (zero outside the
//              image: the backward filter's zero padding), 11 rows x 64 x 9 floats;
//   phase 2  = k_ssim_bwd's loop verbatim, reading those rows from LDS instead of staging them from HBM: horizontal
//              filter of the three adjoint maps (threads of the 54 inner columns), the last 11 rows in registers,
//              v_img of row r - 10.
// Same arithmetic in the same order per value: v_img is bit-identical to the two-kernel path (test_ssim_one_pass_equals_two_kernels).
// NOT the default -- it is slower (see ssim_impl).  Costs: the forward part
// runs on (64 / 54) x ((SEG + 10) / SEG) more pixels, img / gt are read with a 10-pixel halo ((74 / 54) x ((SEG + 20) / SEG));
// LDS 45 KB per workgroup (3 per CU).  The map sum counts a pixel in the workgroup that owns its OUTPUT.
constexpr int SWO = SW - 2 * HALO;
__host__ __device__ constexpr int fused_seg_rows(int nblk) { return RB * nblk - 4 * HALO; }

__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_ssim_fused(
    int W, int H, float weight, const float* __restrict__ img, const float* __restrict__ gt,
    float* __restrict__ v_img, float* __restrict__ block_partials, int n_partials, int NBLK,
    int y_lo, int y_hi, int c_lo, int c_hi) {
  const int SEG = fused_seg_rows(NBLK);
  constexpr int ROWF = RowBlock<3>::ROWF;
  constexpr int ADJF = SW * 9;                       // floats per adjoint row: [column][channel][3 maps]
  __shared__ float sa[RB * ROWF], sb[RB * ROWF];
  __shared__ float sadj[RB * ADJF];
  __shared__ float red[NTH / TGS_WAVE];
  const int tid = threadIdx.x;
  const int col = tid / 3, c = tid - col * 3;
  const int xa0 = blockIdx.x * SWO - HALO;           // first adjoint column of the strip (thread column 0)
  const int gx = xa0 + col;
  const int ys = y_lo + blockIdx.y * SEG, ye = min(ys + SEG, y_hi);
  const bool own_col = col >= HALO && col < SW - HALO;
  // phase 2 reads the 11 adjoint columns col - 5 .. col + 5; the threads of the halo columns produce no output and
  // read a clamped (in-row) window instead
  const int cb = min(max(col - HALO, 0), SWO - 1);
  float w[RB][5], w2[RB][3];
#pragma unroll
  for (int p = 0; p < RB; p++) {
#pragma unroll
    for (int q = 0; q < 5; q++) w[p][q] = 0.f;
    w2[p][0] = 0.f; w2[p][1] = 0.f; w2[p][2] = 0.f;
  }
  float msum = 0.f;
  for (int blk = 0; blk < NBLK; blk++) {
    const int r0 = ys - 2 * HALO + blk * RB;         // first input row of this block; its output rows are r0 - 10 ..
    if (r0 - 2 * HALO >= ye) break;                  // no output row left (uniform)
    __syncthreads();                                 // previous block fully consumed (sa / sb and sadj)
    RowBlock<3>::stage2(img, gt, sa, sb, W, H, xa0, r0, tid);
    __syncthreads();
    // ---- phase 1: moments -> SSIM map -> adjoint values of row r0 + p - 5 at column gx
#pragma unroll
    for (int p = 0; p < RB; p++) {
      const float* ra = sa + p * ROWF + tid;
      const float* rb = sb + p * ROWF + tid;
      float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; k++) {
        const float a = ra[3 * k], b = rb[3 * k];
        const float ga = WK(k) * a, gb = WK(k) * b;
        m1 += ga; m2 += gb;
        e11 = fmaf(ga, a, e11); e22 = fmaf(gb, b, e22); e12 = fmaf(ga, b, e12);
      }
      w[p][0] = m1; w[p][1] = m2; w[p][2] = e11; w[p][3] = e22; w[p][4] = e12;
      const int gy = r0 + p - HALO;                  // map / adjoint row
      float A = 0.f, B = 0.f, C = 0.f;
      if (gy >= ys - HALO && gy < ye + HALO && gy >= 0 && gy < H) {      // uniform
        float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < WIN; k++) {
#pragma unroll
          for (int q = 0; q < 5; q++) o[q] = fmaf(WK(k), w[(p + 1 + k) % RB][q], o[q]);
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1 = o[0], mu2 = o[1];
        const float s11 = o[2] - mu1 * mu1, s22 = o[3] - mu2 * mu2, s12 = o[4] - mu1 * mu2;
        const float n1 = 2.f * mu1 * mu2 + C1, n2 = 2.f * s12 + C2;
        const float d1 = mu1 * mu1 + mu2 * mu2 + C1, d2 = s11 + s22 + C2;
        const float id1 = __builtin_amdgcn_rcpf(d1), id2 = __builtin_amdgcn_rcpf(d2);
        const float m = n1 * n2 * id1 * id2;
        if (gx >= 0 && gx < W) {
          if (own_col && gy >= ys && gy < ye && gy >= c_lo && gy < c_hi) msum += m;
          const float dm_ds12 = 2.f * n1 * id1 * id2;
          const float dm_ds11 = -m * id2;
          const float dm_dmu1 = 2.f * mu2 * n2 * id1 * id2 - m * 2.f * mu1 * id1;
          A = dm_dmu1 - 2.f * mu1 * dm_ds11 - mu2 * dm_ds12;
          B = dm_ds11;
          C = dm_ds12;
        }
      }
      float* o3 = sadj + p * ADJF + tid * 3;
      o3[0] = A; o3[1] = B; o3[2] = C;
    }
    __syncthreads();
    // ---- phase 2: filtered adjoint maps -> v_img of row r0 + p - 10
#pragma unroll
    for (int p = 0; p < RB; p++) {
      const float* r = sadj + p * ADJF + (cb * 3 + c) * 3;
      float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; k++) {
        h0 = fmaf(WK(k), r[9 * k], h0);
        h1 = fmaf(WK(k), r[9 * k + 1], h1);
        h2 = fmaf(WK(k), r[9 * k + 2], h2);
      }
      w2[p][0] = h0; w2[p][1] = h1; w2[p][2] = h2;
      const int gy = r0 + p - 2 * HALO;              // output row
      if (gy >= ys && gy < ye) {                     // uniform
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; k++) {
          o0 = fmaf(WK(k), w2[(p + 1 + k) % RB][0], o0);
          o1 = fmaf(WK(k), w2[(p + 1 + k) % RB][1], o1);
          o2 = fmaf(WK(k), w2[(p + 1 + k) % RB][2], o2);
        }
        if (own_col && gx < W) {
          const size_t pidx = ((size_t)gy * W + gx) * 3 + c;
          v_img[pidx] = weight * (o0 + 2.f * img[pidx] * o1 + gt[pidx] * o2);
        }
      }
    }
  }
  const float tot = wave_sum(msum);
  if ((tid & 63) == 0) red[tid >> 6] = tot;
  __syncthreads();
  const int wg = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
  if (tid == 0) block_partials[wg] = red[0] + red[1] + red[2];
  for (int i = nwg + wg * NTH + tid; i < n_partials; i += nwg * NTH) block_partials[i] = 0.f;
}

}  // namespace

// the fused kernel counts a map pixel in the workgroup that owns its output row: the counted rows must be output rows
static inline bool count_inside(int y0, int y1, int c0, int c1) { return c0 >= c1 || (c0 >= y0 && c1 <= y1); }

static int ssim_impl(int W, int H, const float* img, const float* gt, float weight,
                     float* block_partials, int n_partials, float* v_img, float* scratch,
                     int y0, int y1, int c0, int c1, void* stream) {
  TGS_CHECK_ARG(W > 0 && H > 0, "bad image size");
  TGS_CHECK_ARG(img && gt && block_partials, "null pointer");
  TGS_CHECK_ARG(!v_img || scratch, "gradient needs scratch");
  TGS_CHECK_ARG(0 <= y0 && y0 <= y1 && y1 <= H && y0 <= c0 && c0 <= c1 && c1 <= y1, "bad row range");
  Win win;
  double g[WIN], sum = 0.0;
  for (int i = 0; i < WIN; i++) { g[i] = exp(-(double)((i - HALO) * (i - HALO)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
  for (int i = 0; i < WIN; i++) win.g[i] = (float)(g[i] / sum);
  static const int env_nblk = [] { const char* e = getenv("TGS_SSIM_NBLK"); return e ? max(2, min(16, atoi(e))) : 0; }();
  // TGS_SSIM_FUSED=1 selects the one-pass kernel.  Default 0: measured SLOWER (1080p 119 against 108 us alone, 104 against 100
  // inside the step, profiles/r5_ssim_one_pass.txt) although it moves half the bytes (171 against 343 MB, PMC) -- the two kernels are not
  // HBM-bound (3.3 TB/s), and the one-pass form pays 1.5x the forward arithmetic (halo) at 3 workgroups per CU (45 KB of LDS)
  static const int env_fused = [] { const char* e = getenv("TGS_SSIM_FUSED"); return e ? atoi(e) : 0; }();
  if (v_img && env_fused && y1 > y0 && count_inside(y0, y1, c0, c1)) {
    // one pass, adjoint planes on chip (k_ssim_fused); `scratch` is not touched.  Strip = 54 columns x (11 nblk - 20) rows:
    // long segments keep the vertical halo small ((SEG + 20) / SEG input rows, (SEG + 10) / SEG of the forward arithmetic);
    // the kernel holds 3 workgroups per CU (LDS), so the longest segment that still gives every slot a workgroup is taken
    const int fx = (W + SWO - 1) / SWO;
    auto wgs = [&](int nb) { return (long long)fx * ((y1 - y0 + fused_seg_rows(nb) - 1) / fused_seg_rows(nb)); };
    int nblk = 7;
    while (nblk > 3 && wgs(nblk) < 600) nblk--;
    if (env_nblk) nblk = max(3, env_nblk);
    while (nblk < 16 && wgs(nblk) > n_partials) nblk++;
    TGS_CHECK_ARG(wgs(nblk) <= n_partials, "block_partials too small");
    const int SEGF = fused_seg_rows(nblk);
    hipLaunchKernelGGL(k_ssim_fused, dim3(fx, (y1 - y0 + SEGF - 1) / SEGF, 1), dim3(NTH), 0, (hipStream_t)stream, W, H, weight,
                       img, gt, v_img, block_partials, n_partials, nblk, y0, y1, c0, c1);
    TGS_CHECK_LAUNCH();
    return TGS_OK;
  }
  // forward rows: the gradient rows plus the 5-row halo the backward filter reads (adjoint maps)
  const int f0 = v_img ? max(0, y0 - HALO) : y0, f1 = v_img ? min(H, y1 + HALO) : y1;
  const int rows = f1 - f0;
  if (rows <= 0) return TGS_OK;
  const int sx = (W + SW - 1) / SW;
  int nblk = 4;
  while (nblk > 2 && (long long)sx * ((rows + seg_rows(nblk) - 1) / seg_rows(nblk)) < 3 * 256) nblk--;
  if (env_nblk) nblk = env_nblk;   // tuning override (read once, above)
  // one workgroup sum goes into each of the first entries of block_partials, the rest is zeroed
  while (nblk < 16 && (long long)sx * ((rows + seg_rows(nblk) - 1) / seg_rows(nblk)) > n_partials) nblk++;
  TGS_CHECK_ARG((long long)sx * ((rows + seg_rows(nblk) - 1) / seg_rows(nblk)) <= n_partials, "block_partials too small");
  const int SEG = seg_rows(nblk);
  const dim3 block(NTH);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ssim_fwd, dim3(sx, (rows + SEG - 1) / SEG, 1), block, 0, s, W, H, win, img, gt,
                     v_img ? scratch : nullptr, block_partials, n_partials, nblk, f0, f1, c0, c1);
  TGS_CHECK_LAUNCH();
  if (v_img && y1 > y0) {
    hipLaunchKernelGGL(k_ssim_bwd, dim3(sx, (y1 - y0 + SEG - 1) / SEG, 1), block, 0, s, W, H, win, weight, img, gt,
                       scratch, v_img, nblk, y0, y1);
    TGS_CHECK_LAUNCH();
  }
  return TGS_OK;
}

extern "C" int tgs_ssim_fwd_bwd(int W, int H, const float* img, const float* gt, float weight,
                                float* block_partials, float* v_img, float* scratch,
                                void* stream) {
  // block_partials has one entry per 16x16 tile (the buffer contract of include/tgs.h)
  return ssim_impl(W, H, img, gt, weight, block_partials, ((W + 15) / 16) * ((H + 15) / 16), v_img, scratch,
                   0, H, 0, H, stream);
}

extern "C" int tgs_ssim_fwd_bwd_rows(int W, int H, const float* img, const float* gt, float weight,
                                     float* block_partials, int n_partials, float* v_img,
                                     float* scratch, int y0, int y1, int count_y0, int count_y1,
                                     void* stream) {
  return ssim_impl(W, H, img, gt, weight, block_partials, n_partials, v_img, scratch, y0, y1, count_y0,
                   count_y1, stream);
}
