// raster.hip -- K6 per-tile front-to-back RGB+depth compositing (forward) and K7 its backward
// with the tactile depth/uncertainty loss fused in.   gfx950, wave64.
//
// Spec: SURVEY.md App. B.6 (forward) / B.7 (backward).  Mirrors the op the reference's training
// loop reaches through `ns-train depth-gaussian-splatting` (scripts/train_bunny_real.sh:52):
// gsplat `rasterize_gaussians` (App. A.2) -- but RGB and depth are composited in ONE pass instead
// of the two full rasterizations Splatfacto issues.
//
// MI355X mapping (not a warp-shaped CUDA tiling):
//  * one wave64 owns one 16x16 tile; lane l owns the 4 pixels (x = l&15, y = (l>>4) + 4k).
//    No workgroup barriers, no cross-wave reduction; the 4 pixels of a lane share the dx terms
//    and give 4-way ILP on the exp chain.
//  * the tile's Gaussian list is staged 64 records at a time through LDS (each lane gathers one
//    48-B record with three 16-B loads) and read back as wave-uniform broadcast ds_read_b128.
//  * backward: each lane first sums its 4 pixels in registers, then a 6-step DPP wave reduction
//    yields the tile's total for that Gaussian; lane j of the batch keeps Gaussian j's 10 sums and
//    the batch is written as 64 contiguous 48-B partial records.  There are NO float atomics:
//    cross-tile accumulation is a segmented sum in K8 (deterministic, and it avoids cross-XCD
//    memory-side atomics, which are the expensive primitive on an 8-chiplet part).
//  * blockIdx -> tile mapping gives each XCD a contiguous band of tiles so the gathered splat
//    records of neighbouring tiles stay in that XCD's 4 MB L2.
#include "tgs_common.h"

namespace {

constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.999f;
constexpr float T_STOP = 1e-4f;

__device__ __forceinline__ int xcd_tile(int b, int T) {
  const int q = (T + 7) >> 3;          // tiles per XCD band
  return (b & 7) * q + (b >> 3);       // may be >= T (padded grid)
}

struct LossK {
  const float* gt_rgb;
  const float* gt_depth;
  const float* unc;
  float l1w, dw, uw, eps;
  int on;
};

// ---------------------------------------------------------------------------------------------
// K6 forward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_raster_fwd(
    CamK cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ sorted_gid,
    const int32_t* __restrict__ tile_start, float* __restrict__ out_rgb,
    float* __restrict__ out_depth, float* __restrict__ final_T, int32_t* __restrict__ final_idx) {
  const int tile = xcd_tile(blockIdx.x, T_total);
  if (tile >= T_total) return;
  const int lane = threadIdx.x;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  const int px = tx * TGS_BLOCK + (lane & 15);
  const int py0 = ty * TGS_BLOCK + (lane >> 4);
  const float pxf = (float)px + cam.pix_center;
  float pyf[4];
  bool live[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    pyf[k] = (float)(py0 + 4 * k) + cam.pix_center;
    live[k] = (px < cam.W) && (py0 + 4 * k < cam.H);
  }
  float T[4] = {1.f, 1.f, 1.f, 1.f};
  float Cr[4] = {0.f, 0.f, 0.f, 0.f}, Cg[4] = {0.f, 0.f, 0.f, 0.f}, Cb[4] = {0.f, 0.f, 0.f, 0.f};
  float D[4] = {0.f, 0.f, 0.f, 0.f};
  int last[4] = {-1, -1, -1, -1};

  __shared__ float4 recs[64 * 3];
  const int start = tile_start[tile], end = tile_start[tile + 1];

  // software pipeline: the next batch's records are gathered while the current one is blended
  float4 n0, n1, n2;
  n0 = n1 = n2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (start + lane < end) {
    const float* r = splats + (size_t)sorted_gid[start + lane] * TGS_SPLAT_FLOATS;
    n0 = ld4(r); n1 = ld4(r + 4); n2 = ld4(r + 8);
  }
  for (int base = start; base < end; base += 64) {
    if (__ballot(live[0] | live[1] | live[2] | live[3]) == 0ull) break;
    __syncthreads();
    recs[lane * 3] = n0; recs[lane * 3 + 1] = n1; recs[lane * 3 + 2] = n2;
    __syncthreads();
    if (base + 64 + lane < end) {
      const float* r = splats + (size_t)sorted_gid[base + 64 + lane] * TGS_SPLAT_FLOATS;
      n0 = ld4(r); n1 = ld4(r + 4); n2 = ld4(r + 8);
    }
    const int cnt = min(64, end - base);
    for (int j = 0; j < cnt; j++) {
      const float4 r0 = recs[j * 3], r1 = recs[j * 3 + 1], r2 = recs[j * 3 + 2];
      // r0 = {x, y, depth, opac}  r1 = {a, b, c, r}  r2 = {g, b, -, -}
      const float dx = r0.x - pxf;
      const float hadx2 = 0.5f * r1.x * dx * dx;
      const float bdx = r1.y * dx;
      const float hc = 0.5f * r1.z;
      const int pos = base - start + j;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (live[k]) {
          const float dy = r0.y - pyf[k];
          const float sigma = fmaf(dy, fmaf(hc, dy, bdx), hadx2);
          const float al = fminf(ALPHA_MAX, r0.w * __expf(-sigma));
          if (sigma >= 0.f && al >= ALPHA_MIN) {
            const float Tn = T[k] * (1.f - al);
            if (Tn <= T_STOP) {
              live[k] = false;
            } else {
              const float w = al * T[k];
              Cr[k] = fmaf(w, r1.w, Cr[k]); Cg[k] = fmaf(w, r2.x, Cg[k]);
              Cb[k] = fmaf(w, r2.y, Cb[k]); D[k] = fmaf(w, r0.z, D[k]);
              T[k] = Tn; last[k] = pos;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int py = py0 + 4 * k;
    if (px < cam.W && py < cam.H) {
      const size_t p = (size_t)py * cam.W + px;
      out_rgb[3 * p] = Cr[k] + T[k] * cam.bg[0];
      out_rgb[3 * p + 1] = Cg[k] + T[k] * cam.bg[1];
      out_rgb[3 * p + 2] = Cb[k] + T[k] * cam.bg[2];
      out_depth[p] = D[k];
      final_T[p] = T[k];
      final_idx[p] = last[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K7 backward
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t pair_index(const CamK& cam, const int32_t* __restrict__ group_base,
                                             int gid, float4 r0, float4 r2, int tx, int ty) {
  int x0, y0, x1, y1;
  tile_rect(r0.x, r0.y, __float_as_int(r2.z), cam.TW, cam.TH, x0, y0, x1, y1);
  return (size_t)group_base[gid / TGS_GROUP] + (size_t)__float_as_int(r2.w) +
         (size_t)((ty - y0) * (x1 - x0) + (tx - x0));
}

__global__ __launch_bounds__(64) void k_raster_bwd(
    CamK cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const int32_t* __restrict__ sorted_gid, const int32_t* __restrict__ tile_start,
    const float* __restrict__ out_rgb, const float* __restrict__ out_depth,
    const float* __restrict__ final_T, const int32_t* __restrict__ final_idx,
    const float* __restrict__ v_rgb, const float* __restrict__ v_depth,
    const float* __restrict__ v_alpha, LossK loss, float* __restrict__ partials,
    float* __restrict__ tile_loss) {
  const int tile = xcd_tile(blockIdx.x, T_total);
  if (tile >= T_total) return;
  const int lane = threadIdx.x;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  const int px = tx * TGS_BLOCK + (lane & 15);
  const int py0 = ty * TGS_BLOCK + (lane >> 4);
  const float pxf = (float)px + cam.pix_center;
  const int start = tile_start[tile], end = tile_start[tile + 1];
  const int n = end - start;

  float pyf[4], T[4], vCr[4], vCg[4], vCb[4], vD[4], vAT[4];
  float Sr[4] = {0.f, 0.f, 0.f, 0.f}, Sg[4] = {0.f, 0.f, 0.f, 0.f}, Sb[4] = {0.f, 0.f, 0.f, 0.f};
  float SD[4] = {0.f, 0.f, 0.f, 0.f};
  int last[4];
  float l_l1 = 0.f, l_dep = 0.f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int py = py0 + 4 * k;
    pyf[k] = (float)py + cam.pix_center;
    T[k] = 1.f; vCr[k] = vCg[k] = vCb[k] = vD[k] = vAT[k] = 0.f; last[k] = -1;
    if (px < cam.W && py < cam.H) {
      const size_t p = (size_t)py * cam.W + px;
      const float Tf = final_T[p];
      T[k] = Tf;
      last[k] = final_idx[p];
      float vA = v_alpha ? v_alpha[p] : 0.f;
      if (v_rgb) { vCr[k] = v_rgb[3 * p]; vCg[k] = v_rgb[3 * p + 1]; vCb[k] = v_rgb[3 * p + 2]; }
      if (v_depth) vD[k] = v_depth[p];
      if (loss.on) {
        if (loss.gt_rgb) {
          const float d0 = out_rgb[3 * p] - loss.gt_rgb[3 * p];
          const float d1 = out_rgb[3 * p + 1] - loss.gt_rgb[3 * p + 1];
          const float d2 = out_rgb[3 * p + 2] - loss.gt_rgb[3 * p + 2];
          vCr[k] += loss.l1w * ((d0 > 0.f) - (d0 < 0.f));
          vCg[k] += loss.l1w * ((d1 > 0.f) - (d1 < 0.f));
          vCb[k] += loss.l1w * ((d2 > 0.f) - (d2 < 0.f));
          l_l1 += loss.l1w * (fabsf(d0) + fabsf(d1) + fabsf(d2));
        }
        if (loss.gt_depth) {
          const float gd = loss.gt_depth[p];
          if (gd > 0.f) {
            const float alpha = fmaxf(1.f - Tf, 1e-10f);
            const float ia = 1.0f / alpha;
            const float dhat = out_depth[p] * ia;
            const float r = dhat - gd;
            float wgt = loss.dw;
            if (loss.unc) wgt = wgt / (loss.uw * loss.unc[p] + loss.eps);
            l_dep += wgt * r * r;
            const float gdh = 2.f * wgt * r;
            vD[k] += gdh * ia;
            // alpha is clamped from below: the clamp gates its gradient
            if (1.f - Tf > 1e-10f) vA += -gdh * dhat * ia;
          }
        }
      }
      const float bgdot = cam.bg[0] * vCr[k] + cam.bg[1] * vCg[k] + cam.bg[2] * vCb[k];
      vAT[k] = Tf * (vA - bgdot);  // the T_final * (v_A - b.v_C) factor of B.7
    }
  }
  if (tile_loss) {
    const float a = wave_sum(l_l1), b = wave_sum(l_dep);
    if (lane == 0) { tile_loss[2 * tile] = a; tile_loss[2 * tile + 1] = b; }
  }
  if (n == 0) return;

  const int maxlast = wave_max_i(max(max(last[0], last[1]), max(last[2], last[3])));

  // list positions past the last contributor of every pixel of the tile: zero partials
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int pos = maxlast + 1 + lane; pos < n; pos += 64) {
    const int gid = sorted_gid[start + pos];
    const float* r = splats + (size_t)gid * TGS_SPLAT_FLOATS;
    const size_t P = pair_index(cam, group_base, gid, ld4(r), ld4(r + 8), tx, ty);
    float* o = partials + P * TGS_PARTIAL_FLOATS;
    st4(o, z4); st4(o + 4, z4); st4(o + 8, z4);
  }

  __shared__ float4 recs[64 * 3];
  for (int base = (maxlast >= 0 ? (maxlast >> 6) << 6 : -1); base >= 0; base -= 64) {
    const int cnt = min(64, maxlast + 1 - base);
    size_t P = 0;
    __syncthreads();
    if (lane < cnt) {
      const int gid = sorted_gid[start + base + lane];
      const float* r = splats + (size_t)gid * TGS_SPLAT_FLOATS;
      const float4 a0 = ld4(r), a1 = ld4(r + 4), a2 = ld4(r + 8);
      recs[lane * 3] = a0; recs[lane * 3 + 1] = a1; recs[lane * 3 + 2] = a2;
      P = pair_index(cam, group_base, gid, a0, a2, tx, ty);
    }
    __syncthreads();
    float hold[10];
#pragma unroll
    for (int c = 0; c < 10; c++) hold[c] = 0.f;

    for (int j = cnt - 1; j >= 0; j--) {
      const float4 r0 = recs[j * 3], r1 = recs[j * 3 + 1], r2 = recs[j * 3 + 2];
      const int pos = base + j;
      const float dx = r0.x - pxf;
      const float hadx2 = 0.5f * r1.x * dx * dx;
      const float bdx = r1.y * dx;
      const float hc = 0.5f * r1.z;
      float acc[10];
#pragma unroll
      for (int c = 0; c < 10; c++) acc[c] = 0.f;
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (pos <= last[k]) {
          const float dy = r0.y - pyf[k];
          const float sigma = fmaf(dy, fmaf(hc, dy, bdx), hadx2);
          const float ex = __expf(-sigma);
          const float al = fminf(ALPHA_MAX, r0.w * ex);
          if (sigma >= 0.f && al >= ALPHA_MIN) {
            any = true;
            const float ra = 1.0f / (1.f - al);
            const float Tb = T[k] * ra;  // transmittance in front of this Gaussian
            T[k] = Tb;
            const float w = al * Tb;
            acc[7] = fmaf(w, vCr[k], acc[7]);
            acc[8] = fmaf(w, vCg[k], acc[8]);
            acc[9] = fmaf(w, vCb[k], acc[9]);
            acc[2] = fmaf(w, vD[k], acc[2]);
            float va = (r1.w * Tb - Sr[k] * ra) * vCr[k];
            va = fmaf(r2.x * Tb - Sg[k] * ra, vCg[k], va);
            va = fmaf(r2.y * Tb - Sb[k] * ra, vCb[k], va);
            va = fmaf(r0.z * Tb - SD[k] * ra, vD[k], va);
            va = fmaf(vAT[k], ra, va);
            Sr[k] = fmaf(w, r1.w, Sr[k]); Sg[k] = fmaf(w, r2.x, Sg[k]);
            Sb[k] = fmaf(w, r2.y, Sb[k]); SD[k] = fmaf(w, r0.z, SD[k]);
            acc[3] = fmaf(ex, va, acc[3]);             // d/d opacity
            const float vs = -r0.w * ex * va;          // d/d sigma
            const float vsdx = vs * dx, vsdy = vs * dy;
            acc[4] = fmaf(0.5f * vsdx, dx, acc[4]);    // conic a
            acc[5] = fmaf(vsdx, dy, acc[5]);           // conic b
            acc[6] = fmaf(0.5f * vsdy, dy, acc[6]);    // conic c
            acc[0] = fmaf(vs, fmaf(r1.x, dx, r1.y * dy), acc[0]);  // mean2d x
            acc[1] = fmaf(vs, fmaf(r1.y, dx, r1.z * dy), acc[1]);  // mean2d y
          }
        }
      }
      if (__ballot(any) != 0ull) {
#pragma unroll
        for (int c = 0; c < 10; c++) {
          const float tot = wave_sum(acc[c]);
          hold[c] = (lane == j) ? tot : hold[c];
        }
      }
    }
    if (lane < cnt) {
      float* o = partials + P * TGS_PARTIAL_FLOATS;
      st4(o, make_float4(hold[0], hold[1], hold[2], hold[3]));
      st4(o + 4, make_float4(hold[4], hold[5], hold[6], hold[7]));
      st4(o + 8, make_float4(hold[8], hold[9], 0.f, 0.f));
    }
  }
}

}  // namespace

extern "C" int tgs_rasterize_fwd(const TgsCamera* cam, const float* splats,
                                 const int32_t* sorted_gid, const int32_t* tile_start,
                                 float* out_rgb, float* out_depth, float* final_T,
                                 int32_t* final_idx, void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  TGS_CHECK_ARG(splats && sorted_gid && tile_start && out_rgb && out_depth && final_T && final_idx,
                "null pointer");
  const CamK k = make_camk(cam);
  const int T = k.TW * k.TH;
  const int grid = ((T + 7) / 8) * 8;
  hipLaunchKernelGGL(k_raster_fwd, dim3(grid), dim3(64), 0, (hipStream_t)stream, k, T, splats,
                     sorted_gid, tile_start, out_rgb, out_depth, final_T, final_idx);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_rasterize_bwd(const TgsCamera* cam, const float* splats,
                                 const int32_t* group_base, const int32_t* sorted_gid,
                                 const int32_t* tile_start, const float* out_rgb,
                                 const float* out_depth, const float* final_T,
                                 const int32_t* final_idx, const float* v_rgb,
                                 const float* v_depth, const float* v_alpha,
                                 const TgsLossSpec* loss, float* partials, float* tile_loss,
                                 void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  TGS_CHECK_ARG(splats && group_base && sorted_gid && tile_start && final_T && final_idx && partials,
                "null pointer");
  LossK lk;
  lk.on = 0; lk.gt_rgb = nullptr; lk.gt_depth = nullptr; lk.unc = nullptr;
  lk.l1w = lk.dw = lk.uw = 0.f; lk.eps = 1e-6f;
  if (loss) {
    TGS_CHECK_ARG(out_rgb && out_depth, "fused loss needs out_rgb and out_depth");
    lk.on = 1;
    lk.gt_rgb = (loss->l1_weight != 0.f) ? loss->gt_rgb : nullptr;
    lk.gt_depth = (loss->depth_weight != 0.f) ? loss->gt_depth : nullptr;
    lk.unc = loss->uncertainty;
    lk.l1w = loss->l1_weight; lk.dw = loss->depth_weight;
    lk.uw = loss->uncertainty_weight; lk.eps = loss->eps;
  }
  const CamK k = make_camk(cam);
  const int T = k.TW * k.TH;
  const int grid = ((T + 7) / 8) * 8;
  hipLaunchKernelGGL(k_raster_bwd, dim3(grid), dim3(64), 0, (hipStream_t)stream, k, T, splats,
                     group_base, sorted_gid, tile_start, out_rgb, out_depth, final_T, final_idx,
                     v_rgb, v_depth, v_alpha, lk, partials, tile_loss);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}
