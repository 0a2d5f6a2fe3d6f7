// imgloss.hip -- K10 SSIM forward + gradient image (11x11 Gaussian window, sigma 1.5, zero
// padding), the (1-SSIM) term of the Splatfacto-style loss (1-l)*L1 + l*(1-SSIM) that the
// reference's `depth-gaussian-splatting` method trains with.  gfx950.
//
// Two separable-convolution kernels over 16x16 output tiles with a 5-pixel halo staged in LDS:
//   k_ssim_fwd : 5 windowed moments -> SSIM map; writes the three adjoint maps
//                A = dm/dmu1 (total), B = dm/dE[x^2], C = dm/dE[xy] and a per-block sum of the map
//   k_ssim_bwd : v_img = weight * (G*A + 2*img*(G*B) + gt*(G*C))
// Roofline: HBM (each image is read ~1.3x incl. halo; 9 adjoint planes written then read).
#include <math.h>
#include "tgs_common.h"

namespace {

constexpr int WIN = 11, HALO = 5, TS = 16, EXT = TS + 2 * HALO;  // 26
struct Win { float g[WIN]; };

// Register-blocked separable convolution: every thread produces OPT consecutive outputs along the
// filter direction from a sliding window of OPT+10 inputs held in registers.  One workgroup owns a
// 16x16 tile for ALL THREE channels, so that the interleaved [H,W,3] images are read as contiguous
// 78-float row segments (a per-channel block would use 4 of every 12 bytes it fetches), and the
// adjoints are stored interleaved per pixel ([H,W,3 ch,3 maps]) so the second kernel reads
// contiguous 234-float row segments.
constexpr int OPT = 4;
constexpr int CG = TS / OPT;  // column / row groups per tile edge
constexpr int HS = 20;        // row stride of the horizontally-filtered maps: 4*HS = 16 (mod 32), so the
                              // four row groups of a 32-lane LDS access group hit disjoint banks

__global__ __launch_bounds__(256) void k_ssim_fwd(int W, int H, Win win,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ gt,
                                                  float* __restrict__ adj /*[H,W,3,3] or null*/,
                                                  float* __restrict__ block_partials) {
  __shared__ float sa[3][EXT][EXT + 1], sb[3][EXT][EXT + 1];
  __shared__ float h[3][5][EXT][HS];
  __shared__ float red[4];
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const int tid = threadIdx.x;
  {
    // all global loads of the halo tile are issued before the first LDS store (the loop is
    // otherwise a chain of dependent load->store latencies)
    constexpr int NIT = (EXT * EXT * 3 + 255) / 256;
    float va[NIT], vb[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = tid + 256 * it;
      const int ly = i / (EXT * 3), e = i - ly * (EXT * 3);
      const int lx = e / 3, c = e - lx * 3;
      const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
      va[it] = 0.f; vb[it] = 0.f;
      if (i < EXT * EXT * 3 && gx >= 0 && gx < W && gy >= 0 && gy < H) {
        const size_t p = ((size_t)gy * W + gx) * 3 + c;
        va[it] = img[p]; vb[it] = gt[p];
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = tid + 256 * it;
      if (i < EXT * EXT * 3) {
        const int ly = i / (EXT * 3), e = i - ly * (EXT * 3);
        const int lx = e / 3, c = e - lx * 3;
        sa[c][ly][lx] = va[it]; sb[c][ly][lx] = vb[it];
      }
    }
  }
  __syncthreads();
  // horizontal pass; consecutive lanes take consecutive ROWS (row stride 27 is odd -> no conflicts)
  for (int i = tid; i < 3 * EXT * CG; i += 256) {
    const int c = i / (EXT * CG), r = i - c * (EXT * CG);
    const int cg = r / EXT, ly = r - cg * EXT, lx0 = cg * OPT;
    float a[OPT + WIN - 1], b[OPT + WIN - 1];
#pragma unroll
    for (int k = 0; k < OPT + WIN - 1; k++) { a[k] = sa[c][ly][lx0 + k]; b[k] = sb[c][ly][lx0 + k]; }
#pragma unroll
    for (int o = 0; o < OPT; o++) {
      float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; k++) {
        const float g = win.g[k], ga = g * a[o + k], gb = g * b[o + k];
        m1 += ga; m2 += gb;
        e11 = fmaf(ga, a[o + k], e11); e22 = fmaf(gb, b[o + k], e22); e12 = fmaf(ga, b[o + k], e12);
      }
      h[c][0][ly][lx0 + o] = m1; h[c][1][ly][lx0 + o] = m2; h[c][2][ly][lx0 + o] = e11;
      h[c][3][ly][lx0 + o] = e22; h[c][4][ly][lx0 + o] = e12;
    }
  }
  __syncthreads();
  float msum = 0.f;
  if (tid < 3 * TS * CG) {
    const int c = tid / (TS * CG), r = tid - c * (TS * CG);
    const int lx = r & (TS - 1), ly0 = (r / TS) * OPT;
    float out[5][OPT];
#pragma unroll
    for (int q = 0; q < 5; q++) {
      float col[OPT + WIN - 1];
#pragma unroll
      for (int k = 0; k < OPT + WIN - 1; k++) col[k] = h[c][q][ly0 + k][lx];
#pragma unroll
      for (int o = 0; o < OPT; o++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; k++) acc = fmaf(win.g[k], col[o + k], acc);
        out[q][o] = acc;
      }
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
#pragma unroll
    for (int o = 0; o < OPT; o++) {
      const float mu1 = out[0][o], mu2 = out[1][o];
      const float s11 = out[2][o] - mu1 * mu1, s22 = out[3][o] - mu2 * mu2, s12 = out[4][o] - mu1 * mu2;
      const float n1 = 2.f * mu1 * mu2 + C1, n2 = 2.f * s12 + C2;
      const float d1 = mu1 * mu1 + mu2 * mu2 + C1, d2 = s11 + s22 + C2;
      const float id1 = 1.f / d1, id2 = 1.f / d2;
      const float m = n1 * n2 * id1 * id2;
      const int gx = x0 + lx, gy = y0 + ly0 + o;
      if (gx < W && gy < H) {
        msum += m;
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
  const float tot = wave_sum(msum);
  if ((tid & 63) == 0) red[tid >> 6] = tot;
  __syncthreads();
  if (tid == 0) block_partials[blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_ssim_bwd(int W, int H, Win win, float weight,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ gt,
                                                  const float* __restrict__ adj,
                                                  float* __restrict__ v_img) {
  __shared__ float s[9][EXT][EXT + 1];   // [c*3 + map]
  __shared__ float h[9][EXT][HS];
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const int tid = threadIdx.x;
  {
    constexpr int NIT = (EXT * EXT * 9 + 255) / 256;
    float vv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = tid + 256 * it;
      const int ly = i / (EXT * 9), e = i - ly * (EXT * 9);
      const int lx = e / 9, q = e - lx * 9;
      const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
      vv[it] = 0.f;
      if (i < EXT * EXT * 9 && gx >= 0 && gx < W && gy >= 0 && gy < H) vv[it] = adj[((size_t)gy * W + gx) * 9 + q];
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = tid + 256 * it;
      if (i < EXT * EXT * 9) {
        const int ly = i / (EXT * 9), e = i - ly * (EXT * 9);
        const int lx = e / 9, q = e - lx * 9;
        s[q][ly][lx] = vv[it];
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 9 * EXT * CG; i += 256) {
    const int q = i / (EXT * CG), r = i - q * (EXT * CG);
    const int cg = r / EXT, ly = r - cg * EXT, lx0 = cg * OPT;
    float a[OPT + WIN - 1];
#pragma unroll
    for (int k = 0; k < OPT + WIN - 1; k++) a[k] = s[q][ly][lx0 + k];
#pragma unroll
    for (int o = 0; o < OPT; o++) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; k++) acc = fmaf(win.g[k], a[o + k], acc);
      h[q][ly][lx0 + o] = acc;
    }
  }
  __syncthreads();
  if (tid >= 3 * TS * CG) return;
  const int c = tid / (TS * CG), r = tid - c * (TS * CG);
  const int lx = r & (TS - 1), ly0 = (r / TS) * OPT;
  float out[3][OPT];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    float col[OPT + WIN - 1];
#pragma unroll
    for (int k = 0; k < OPT + WIN - 1; k++) col[k] = h[c * 3 + q][ly0 + k][lx];
#pragma unroll
    for (int o = 0; o < OPT; o++) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; k++) acc = fmaf(win.g[k], col[o + k], acc);
      out[q][o] = acc;
    }
  }
#pragma unroll
  for (int o = 0; o < OPT; o++) {
    const int gx = x0 + lx, gy = y0 + ly0 + o;
    if (gx < W && gy < H) {
      const size_t p = ((size_t)gy * W + gx) * 3 + c;
      v_img[p] = weight * (out[0][o] + 2.f * img[p] * out[1][o] + gt[p] * out[2][o]);
    }
  }
}

}  // namespace

extern "C" int tgs_ssim_fwd_bwd(int W, int H, const float* img, const float* gt, float weight,
                                float* block_partials, float* v_img, float* scratch,
                                void* stream) {
  TGS_CHECK_ARG(W > 0 && H > 0, "bad image size");
  TGS_CHECK_ARG(img && gt && block_partials, "null pointer");
  TGS_CHECK_ARG(!v_img || scratch, "gradient needs scratch");
  Win win;
  double g[WIN], sum = 0.0;
  for (int i = 0; i < WIN; i++) { g[i] = exp(-(double)((i - HALO) * (i - HALO)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
  for (int i = 0; i < WIN; i++) win.g[i] = (float)(g[i] / sum);
  const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, 1), block(256);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ssim_fwd, grid, block, 0, s, W, H, win, img, gt, v_img ? scratch : nullptr,
                     block_partials);
  TGS_CHECK_LAUNCH();
  if (v_img) {
    hipLaunchKernelGGL(k_ssim_bwd, grid, block, 0, s, W, H, win, weight, img, gt, scratch, v_img);
    TGS_CHECK_LAUNCH();
  }
  return TGS_OK;
}
