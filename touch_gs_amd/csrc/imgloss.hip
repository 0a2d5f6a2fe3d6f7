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

__global__ __launch_bounds__(256) void k_ssim_fwd(int W, int H, Win win,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ gt,
                                                  float* __restrict__ adj /*[3 maps][3 ch][H*W] or null*/,
                                                  float* __restrict__ block_partials) {
  __shared__ float sa[EXT][EXT + 1], sb[EXT][EXT + 1];
  __shared__ float h[5][EXT][TS + 1];
  __shared__ float red[4];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const int tid = threadIdx.x;
  for (int i = tid; i < EXT * EXT; i += 256) {
    const int ly = i / EXT, lx = i - ly * EXT;
    const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
    float a = 0.f, b = 0.f;
    if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
      const size_t p = ((size_t)gy * W + gx) * 3 + c;
      a = img[p]; b = gt[p];
    }
    sa[ly][lx] = a; sb[ly][lx] = b;
  }
  __syncthreads();
  for (int i = tid; i < EXT * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < WIN; k++) {
      const float a = sa[ly][lx + k], b = sb[ly][lx + k], g = win.g[k];
      m1 = fmaf(g, a, m1); m2 = fmaf(g, b, m2);
      e11 = fmaf(g, a * a, e11); e22 = fmaf(g, b * b, e22); e12 = fmaf(g, a * b, e12);
    }
    h[0][ly][lx] = m1; h[1][ly][lx] = m2; h[2][ly][lx] = e11; h[3][ly][lx] = e22; h[4][ly][lx] = e12;
  }
  __syncthreads();
  const int lx = tid & 15, ly = tid >> 4;
  float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
  for (int k = 0; k < WIN; k++) {
    const float g = win.g[k];
    mu1 = fmaf(g, h[0][ly + k][lx], mu1); mu2 = fmaf(g, h[1][ly + k][lx], mu2);
    e11 = fmaf(g, h[2][ly + k][lx], e11); e22 = fmaf(g, h[3][ly + k][lx], e22);
    e12 = fmaf(g, h[4][ly + k][lx], e12);
  }
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
  const float n1 = 2.f * mu1 * mu2 + C1, n2 = 2.f * s12 + C2;
  const float d1 = mu1 * mu1 + mu2 * mu2 + C1, d2 = s11 + s22 + C2;
  const float id1 = 1.f / d1, id2 = 1.f / d2;
  const float m = n1 * n2 * id1 * id2;
  const int gx = x0 + lx, gy = y0 + ly;
  const bool in = gx < W && gy < H;
  if (adj && in) {
    const float dm_ds12 = 2.f * n1 * id1 * id2;
    const float dm_ds11 = -m * id2;
    const float dm_dmu1 = 2.f * mu2 * n2 * id1 * id2 - m * 2.f * mu1 * id1;
    const size_t HW = (size_t)W * H, p = (size_t)gy * W + gx;
    adj[(0 * 3 + c) * HW + p] = dm_dmu1 - 2.f * mu1 * dm_ds11 - mu2 * dm_ds12;
    adj[(1 * 3 + c) * HW + p] = dm_ds11;
    adj[(2 * 3 + c) * HW + p] = dm_ds12;
  }
  const float tot = wave_sum(in ? m : 0.f);
  if ((tid & 63) == 0) red[tid >> 6] = tot;
  __syncthreads();
  if (tid == 0)
    block_partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] =
        red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_ssim_bwd(int W, int H, Win win, float weight,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ gt,
                                                  const float* __restrict__ adj,
                                                  float* __restrict__ v_img) {
  __shared__ float s[3][EXT][EXT + 1];
  __shared__ float h[3][EXT][TS + 1];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const int tid = threadIdx.x;
  const size_t HW = (size_t)W * H;
  for (int i = tid; i < EXT * EXT; i += 256) {
    const int ly = i / EXT, lx = i - ly * EXT;
    const int gx = x0 + lx - HALO, gy = y0 + ly - HALO;
    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
    const size_t p = (size_t)gy * W + gx;
#pragma unroll
    for (int k = 0; k < 3; k++) s[k][ly][lx] = in ? adj[(k * 3 + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < EXT * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
    for (int k = 0; k < WIN; k++) {
      const float g = win.g[k];
      a = fmaf(g, s[0][ly][lx + k], a); b = fmaf(g, s[1][ly][lx + k], b); cc = fmaf(g, s[2][ly][lx + k], cc);
    }
    h[0][ly][lx] = a; h[1][ly][lx] = b; h[2][ly][lx] = cc;
  }
  __syncthreads();
  const int lx = tid & 15, ly = tid >> 4;
  const int gx = x0 + lx, gy = y0 + ly;
  if (gx >= W || gy >= H) return;
  float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
  for (int k = 0; k < WIN; k++) {
    const float g = win.g[k];
    a = fmaf(g, h[0][ly + k][lx], a); b = fmaf(g, h[1][ly + k][lx], b); cc = fmaf(g, h[2][ly + k][lx], cc);
  }
  const size_t p = ((size_t)gy * W + gx) * 3 + c;
  v_img[p] = weight * (a + 2.f * img[p] * b + gt[p] * cc);
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
  const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, 3), block(256);
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
