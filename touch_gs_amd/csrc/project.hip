// project.hip -- K1 projection + EWA covariance + SH colour (forward), K8 its backward,
// K8a segmented reduction of per-(tile,Gaussian) partial gradients.   gfx950, wave64.
//
// Algorithm: SURVEY.md Appendix B.1-B.5 (forward) and B.8 (backward).  The reference tree has no
// source for this path (scripts/train_bunny_real.sh:52 shells out to an absent submodule); the
// op surface mirrored is gsplat-0.1 `project_gaussians` / `spherical_harmonics` (App. A.2).
//
// Roofline: pure streaming, HBM bound.  Forward reads 44+12K B and writes 48 B per Gaussian;
// backward reads 44+12K+48 (+48*tiles_hit of partials) and writes 44+12K (+8) B per Gaussian.
#include "tgs_adam.h"
#include "tgs_binning.h"

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f,
                SH_C2_2 = 0.31539156525252005f, SH_C2_3 = -1.0925484305920792f,
                SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f,
                SH_C3_2 = -0.4570457994644658f, SH_C3_3 = 0.3731763325901154f,
                SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;
constexpr float BLUR = 0.3f;

// The basis and the colour sum are evaluated by K1 AND by the colour prefetch of the fused optimizer
// kernel, which must agree bit for bit.  Under -ffp-contract=fast the backend fuses any multiply that
// feeds an add/subtract, and whether it does can depend on the surrounding code -- so every such
// pair is written as an explicit fmaf here and no fusable pattern is left to the compiler.
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* Y) {
  Y[0] = SH_C0;
  if constexpr (DEG >= 1) { Y[1] = -SH_C1 * y; Y[2] = SH_C1 * z; Y[3] = -SH_C1 * x; }
  if constexpr (DEG >= 2) {
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xx_yy = fmaf(x, x, -yy);                      // xx - yy
    Y[4] = SH_C2_0 * x * y; Y[5] = SH_C2_1 * y * z;
    Y[6] = SH_C2_2 * fmaf(-y, y, fmaf(2.f, zz, -xx));         // 2zz - xx - yy
    Y[7] = SH_C2_3 * x * z; Y[8] = SH_C2_4 * xx_yy;
    if constexpr (DEG >= 3) {
      const float b = fmaf(-y, y, fmaf(4.f, zz, -xx));        // 4zz - xx - yy
      Y[9] = SH_C3_0 * y * fmaf(3.f, xx, -yy); Y[10] = SH_C3_1 * x * y * z;
      Y[11] = SH_C3_2 * y * b;
      Y[12] = SH_C3_3 * z * fmaf(-3.f, yy, fmaf(-3.f, xx, 2.f * zz));
      Y[13] = SH_C3_4 * x * b;
      Y[14] = SH_C3_5 * z * xx_yy; Y[15] = SH_C3_6 * x * fmaf(-3.f, yy, xx);
    }
  }
}

// dY_k/d(x,y,z), k >= 1 (the DC term has zero derivative)
template <int DEG>
__device__ __forceinline__ void sh_basis_grad(float x, float y, float z, float (*dY)[3]) {
  constexpr int K = (DEG + 1) * (DEG + 1);
#pragma unroll
  for (int k = 0; k < K; k++) { dY[k][0] = 0.f; dY[k][1] = 0.f; dY[k][2] = 0.f; }
  if constexpr (DEG >= 1) { dY[1][1] = -SH_C1; dY[2][2] = SH_C1; dY[3][0] = -SH_C1; }
  if constexpr (DEG >= 2) {
    dY[4][0] = SH_C2_0 * y; dY[4][1] = SH_C2_0 * x;
    dY[5][1] = SH_C2_1 * z; dY[5][2] = SH_C2_1 * y;
    dY[6][0] = -2.f * SH_C2_2 * x; dY[6][1] = -2.f * SH_C2_2 * y; dY[6][2] = 4.f * SH_C2_2 * z;
    dY[7][0] = SH_C2_3 * z; dY[7][2] = SH_C2_3 * x;
    dY[8][0] = 2.f * SH_C2_4 * x; dY[8][1] = -2.f * SH_C2_4 * y;
  }
  if constexpr (DEG >= 3) {
    const float xx = x * x, yy = y * y, zz = z * z;
    dY[9][0] = 6.f * SH_C3_0 * x * y; dY[9][1] = 3.f * SH_C3_0 * (xx - yy);
    dY[10][0] = SH_C3_1 * y * z; dY[10][1] = SH_C3_1 * x * z; dY[10][2] = SH_C3_1 * x * y;
    dY[11][0] = -2.f * SH_C3_2 * x * y; dY[11][1] = SH_C3_2 * (4.f * zz - xx - 3.f * yy);
    dY[11][2] = 8.f * SH_C3_2 * y * z;
    dY[12][0] = -6.f * SH_C3_3 * x * z; dY[12][1] = -6.f * SH_C3_3 * y * z;
    dY[12][2] = SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
    dY[13][0] = SH_C3_4 * (4.f * zz - 3.f * xx - yy); dY[13][1] = -2.f * SH_C3_4 * x * y;
    dY[13][2] = 8.f * SH_C3_4 * x * z;
    dY[14][0] = 2.f * SH_C3_5 * x * z; dY[14][1] = -2.f * SH_C3_5 * y * z;
    dY[14][2] = SH_C3_5 * (xx - yy);
    dY[15][0] = 3.f * SH_C3_6 * (xx - yy); dY[15][1] = -6.f * SH_C3_6 * x * y;
  }
}

// View-dependent colour of one Gaussian (B.5): rgb = max(sum_k Y_k(dir) c_k + 0.5, 0), dir from the
// camera position to the mean.  `c` = the Gaussian's coefficient row (3 floats per basis) in global
// memory, LDS or registers -- K1 and the colour prefetch of the fused optimizer kernel share this
// function so that both produce the same bits.
template <int DEG, typename Row>
__device__ __forceinline__ void sh_color(const float* m, const float* campos, Row c, float* rgb) {
  constexpr int K = (DEG + 1) * (DEG + 1);
  float dx = m[0] - campos[0], dy = m[1] - campos[1], dz = m[2] - campos[2];
  const float inv = 1.0f / sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  float Y[16];
  sh_basis<DEG>(dx * inv, dy * inv, dz * inv, Y);
  rgb[0] = 0.f; rgb[1] = 0.f; rgb[2] = 0.f;
#pragma unroll
  for (int k = 0; k < K; k++) {
    rgb[0] = fmaf(Y[k], c[3 * k], rgb[0]); rgb[1] = fmaf(Y[k], c[3 * k + 1], rgb[1]);
    rgb[2] = fmaf(Y[k], c[3 * k + 2], rgb[2]);
  }
#pragma unroll
  for (int ch = 0; ch < 3; ch++) rgb[ch] = fmaxf(rgb[ch] + 0.5f, 0.f);
}

// Colour prefetch request of the fused K8+Adam kernel: the NEXT view's camera position, the [N,3]
// colour buffer and the tag word that marks it valid (colors == NULL: off).
// Front prefetch: the fused K8 + Adam kernel also runs the NEXT view's K1 (projection, record, tile counting of its
// 256-Gaussian group) on the parameters it has just updated -- they are in registers, the workgroup is the binning
// group, and the kernel is HBM-bound with the VALU idle.  splats == nullptr: not requested.
struct NextFront {
  CamK cam;
  float* splats;
  int32_t* radii;
  int32_t* group_base;
  int32_t* tile_count;
  int32_t* rank;
  int32_t* status;
  long long capacity;
  int32_t* sticky;
};

struct NextView {
  float campos[3];
  float* colors;
  int32_t* tag;
  int32_t tag_value;
  NextFront fr;
};

// ---------------------------------------------------------------------------------------------
// Screen position to ~1e-5 px: double-float (hi + lo) evaluation of  u = fx X/Z + cx.
// In plain fp32 the view transform, the quotient and the absolute pixel coordinate each carry an error
// of about one ulp of a number of size |u| -- 1.2e-4 px at 1080p, 2.4e-4 px at 4K -- which reaches the
// composited image at the 1e-4 level (alpha of a sub-pixel Gaussian is that sensitive to its centre).
// K1 is HBM-bound, so ~150 extra VALU per Gaussian are free: products and sums are carried with
// their rounding errors (FMA-based two-product, two-sum), and the record stores the position
// RELATIVE to the origin of the Gaussian's own tile rectangle (16 x0, 16 y0), a number of a few tens
// of pixels whose fp32 ulp is ~1e-6 px.  The compositing kernels subtract the (exactly
// representable) tile centre relative to the same origin.
// The `asm` statements pin each product / sum as computed: -ffp-contract=fast would otherwise fuse a
// product into a later add and break the error-free transformations.
// ---------------------------------------------------------------------------------------------
struct DF { float hi, lo; };
__device__ __forceinline__ float df_pin(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ DF two_prod(float a, float b) {
  const float p = df_pin(a * b);
  return {p, fmaf(a, b, -p)};
}
__device__ __forceinline__ DF two_sum(float a, float b) {
  const float s = df_pin(a + b);
  const float bb = df_pin(s - a);
  const float ea = df_pin(a - df_pin(s - bb)), eb = df_pin(b - bb);
  return {s, df_pin(ea + eb)};
}
// r . m + t
__device__ __forceinline__ DF df_affine(const float* r, const float* m, float t) {
  const DF p1 = two_prod(r[0], m[0]), p2 = two_prod(r[1], m[1]), p3 = two_prod(r[2], m[2]);
  const DF s1 = two_sum(p1.hi, p2.hi), s2 = two_sum(s1.hi, p3.hi), s3 = two_sum(s2.hi, t);
  return {s3.hi, ((p1.lo + p2.lo) + p3.lo) + ((s1.lo + s2.lo) + s3.lo)};
}
// f * (X / Z) + c  ->  (hi, lo) with hi + lo accurate to ~2^-45 relative
__device__ __forceinline__ DF df_project(float f, DF X, DF Z, float c) {
  const float q = df_pin(X.hi / Z.hi);                            // correctly rounded quotient
  const float r = fmaf(-q, Z.hi, X.hi);                          // exact remainder
  const float ql = (r + fmaf(-q, Z.lo, X.lo)) * __builtin_amdgcn_rcpf(Z.hi);
  const DF p = two_prod(f, q);
  const float pl = fmaf(f, ql, p.lo);
  const DF d = two_sum(p.hi, c);
  return {d.hi, d.lo + pl};
}
// screen position of mean m relative to the tile-rect origin (16 x0, 16 y0)
__device__ __forceinline__ void screen_pos_rel(const CamK& cam, const float* m, unsigned rect, float& xr, float& yr) {
  const DF X = df_affine(cam.R, m, cam.t[0]), Y = df_affine(cam.R + 3, m, cam.t[1]), Z = df_affine(cam.R + 6, m, cam.t[2]);
  const DF u = df_project(cam.fx, X, Z, cam.cx), v = df_project(cam.fy, Y, Z, cam.cy);
  int x0, y0, w, h;
  unpack_rect(rect, x0, y0, w, h);
  xr = df_pin(u.hi - (float)(x0 * TGS_BLOCK)) + u.lo;
  yr = df_pin(v.hi - (float)(y0 * TGS_BLOCK)) + v.lo;
}

struct Geom {  // everything the forward and backward share for one Gaussian
  float tx, ty, tz;
  float Rq[9];      // rotation of the normalised quaternion
  float qn[4];      // normalised quaternion
  float qnorm;
  float s[3];       // exp(log_scale) * glob_scale
  float Sig[6];     // 3-D covariance, upper triangle: 00 01 02 11 12 22
  float ucx, ucy;   // clamped tx/tz, ty/tz
  bool inx, iny;
  float Tm[6];      // J * R_world (2x3)
  float c00, c01, c11, det;
};

__device__ __forceinline__ void geom_eval(const CamK& cam, const float* m, const float* ls,
                                          const float* q, Geom& G) {
  // This code is inlined into several kernels -- the stand-alone K1 and the fused optimizer kernel that runs the NEXT
  // view's K1 (front prefetch).  Under -ffp-contract=fast the backend chose different products to fuse in the two
  // (1-ulp differences in ~9 % of the conics), so project.hip is compiled with -ffp-contract=on (build.sh): fusion is
  // then decided per source expression by the front end, identically in every instantiation.
  const float* R = cam.R;
  G.tx = R[0] * m[0] + R[1] * m[1] + R[2] * m[2] + cam.t[0];
  G.ty = R[3] * m[0] + R[4] * m[1] + R[5] * m[2] + cam.t[1];
  G.tz = R[6] * m[0] + R[7] * m[1] + R[8] * m[2] + cam.t[2];
  const float qq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  G.qnorm = sqrtf(qq);
  const float inv = 1.0f / G.qnorm;
  const float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  G.qn[0] = w; G.qn[1] = x; G.qn[2] = y; G.qn[3] = z;
  G.Rq[0] = 1.f - 2.f * (y * y + z * z); G.Rq[1] = 2.f * (x * y - w * z); G.Rq[2] = 2.f * (x * z + w * y);
  G.Rq[3] = 2.f * (x * y + w * z); G.Rq[4] = 1.f - 2.f * (x * x + z * z); G.Rq[5] = 2.f * (y * z - w * x);
  G.Rq[6] = 2.f * (x * z - w * y); G.Rq[7] = 2.f * (y * z + w * x); G.Rq[8] = 1.f - 2.f * (x * x + y * y);
#pragma unroll
  for (int j = 0; j < 3; j++) G.s[j] = expf(ls[j]) * cam.glob_scale;
  float M[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) M[3 * r + c] = G.Rq[3 * r + c] * G.s[c];
  G.Sig[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
  G.Sig[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
  G.Sig[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
  G.Sig[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
  G.Sig[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
  G.Sig[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
  const float rz = 1.0f / G.tz;
  const float ux = G.tx * rz, uy = G.ty * rz;
  G.inx = (ux >= -cam.limx) && (ux <= cam.limx);
  G.iny = (uy >= -cam.limy) && (uy <= cam.limy);
  G.ucx = fminf(fmaxf(ux, -cam.limx), cam.limx);
  G.ucy = fminf(fmaxf(uy, -cam.limy), cam.limy);
  const float j00 = cam.fx * rz, j02 = -cam.fx * G.ucx * rz;
  const float j11 = cam.fy * rz, j12 = -cam.fy * G.ucy * rz;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    G.Tm[c] = j00 * R[c] + j02 * R[6 + c];
    G.Tm[3 + c] = j11 * R[3 + c] + j12 * R[6 + c];
  }
  // TS = Tm * Sigma (2x3), cov = TS * Tm^T
  float TS[6];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const float a = G.Tm[3 * r], b = G.Tm[3 * r + 1], c = G.Tm[3 * r + 2];
    TS[3 * r + 0] = a * G.Sig[0] + b * G.Sig[1] + c * G.Sig[2];
    TS[3 * r + 1] = a * G.Sig[1] + b * G.Sig[3] + c * G.Sig[4];
    TS[3 * r + 2] = a * G.Sig[2] + b * G.Sig[4] + c * G.Sig[5];
  }
  G.c00 = TS[0] * G.Tm[0] + TS[1] * G.Tm[1] + TS[2] * G.Tm[2] + BLUR;
  G.c01 = TS[0] * G.Tm[3] + TS[1] * G.Tm[4] + TS[2] * G.Tm[5];
  G.c11 = TS[3] * G.Tm[3] + TS[4] * G.Tm[4] + TS[5] * G.Tm[5] + BLUR;
  G.det = G.c00 * G.c11 - G.c01 * G.c01;
}

// ---------------------------------------------------------------------------------------------
// K1 forward
// ---------------------------------------------------------------------------------------------
// FUSED: the workgroup (= one 256-Gaussian binning group) also builds the group scan, allocates the
// group's pair range and counts its (tile, Gaussian) intersections (K3a), so the records are not
// re-read and the pair offset is stored with the record instead of patched into it afterwards.
// PRE: the colours come from colors_in although DEG >= 0 (colour prefetch, k_project_fwd_colors)
// d sigmoid / d logit without the cancellation of o (1 - o): at opacity 0.99999 (logit 12, a converged opaque surface)
// 1 - o has only 7 bits left in fp32.  sigmoid'(x) = t / (1 + t)^2 with t = exp(-|x|) (even in x, no overflow).
__device__ __forceinline__ float sigmoid_deriv(float x) {
  const float t = expf(-fabsf(x));
  const float s = 1.0f / (1.0f + t);
  return t * s * s;
}

// project_fwd_core: everything after the parameters and the colour are in registers (m, ls, q, opac_logit value,
// rgb; ignored for g >= N) -- shared by the stand-alone K1 below and by the fused K8 + Adam kernel, which runs
// the NEXT view's K1 on the parameters it has just updated (front prefetch).
template <bool FUSED>
__device__ __forceinline__ void project_fwd_core(
    const CamK& cam, int N, int g, const float* m, const float* ls, const float* q, float ol, const float* rgb,
    float* __restrict__ splats, int32_t* __restrict__ radii, int32_t* __restrict__ group_base,
    int32_t* __restrict__ tile_count, int32_t* __restrict__ rank, int32_t* __restrict__ status,
    long long capacity, int32_t* __restrict__ sticky, GroupScan* Sp) {
  unsigned rect = 0u;
  float tz = 0.f, c1 = 0.f, c2 = 0.f;
  if (g < N) {
  const float opac = 1.0f / (1.0f + expf(-ol));

  Geom G;
  geom_eval(cam, m, ls, q, G);
  float x2 = 0.f, y2 = 0.f, ca = 0.f, cb = 0.f, cc = 0.f;
  int radius = 0;
  if (G.tz > cam.near_plane) {
    const float rz = 1.0f / G.tz;
    x2 = cam.fx * G.tx * rz + cam.cx;
    y2 = cam.fy * G.ty * rz + cam.cy;
    if (G.det > 0.f) {
      const float id = 1.0f / G.det;
      ca = G.c11 * id; cb = -G.c01 * id; cc = G.c00 * id;
      const float mid = 0.5f * (G.c00 + G.c11);
      const float lam1 = mid + sqrtf(fmaxf(0.1f, mid * mid - G.det));
      const int r = (int)ceilf(3.0f * sqrtf(lam1));
      int x0, y0, x1, y1;
      tile_rect(x2, y2, r, cam.TW, cam.TH, x0, y0, x1, y1);
      if ((x1 - x0) * (y1 - y0) > 0) {
        radius = r;
        // Output-preserving tightening: alpha = o*exp(-sigma) >= 1/255 only inside the ellipse
        // d^T Cov^-1 d <= 2 ln(255 o), whose axis-aligned half extents are sqrt(tau2 * Cov_ii).
        const float tau2 = 2.0f * logf(255.0f * opac);
        if (tau2 > 0.f) {
          const float ex = sqrtf(tau2 * G.c00) + 0.02f, ey = sqrtf(tau2 * G.c11) + 0.02f;
          const int jx0 = (int)ceilf(x2 - ex - cam.pix_center), jx1 = (int)floorf(x2 + ex - cam.pix_center);
          const int jy0 = (int)ceilf(y2 - ey - cam.pix_center), jy1 = (int)floorf(y2 + ey - cam.pix_center);
          const int tx0 = max(x0, max(jx0, 0) >> 4), tx1 = min(x1 - 1, jx1 >> 4);
          const int ty0 = max(y0, max(jy0, 0) >> 4), ty1 = min(y1 - 1, jy1 >> 4);
          if (jx1 >= 0 && jy1 >= 0 && tx1 >= tx0 && ty1 >= ty0)
            rect = pack_rect(tx0, ty0, tx1 - tx0 + 1, ty1 - ty0 + 1);
        }
      }
    }
  }
  if (radius == 0) { ca = 0.f; cb = 0.f; cc = 0.f; }  // conic != 0  <=>  passed the App. B culls
  tz = G.tz;
  // record slots 0, 1: the screen position relative to the origin of the Gaussian's tile rect, from the
  // compensated evaluation (the tile decisions above keep using the plain fp32 x2, y2: they carry a
  // 0.02 px margin).  Gaussians behind the near plane keep (0, 0): they are in no list.
  float xs = x2, ys = y2;
  if (G.tz > cam.near_plane) screen_pos_rel(cam, m, rect, xs, ys);   // free: K1+count 185 us with and without (same box)
  float* o = splats + (size_t)g * TGS_SPLAT_FLOATS;
  st4(o, make_float4(xs, ys, G.tz, opac));
  st4(o + 4, make_float4(ca, cb, cc, rgb[0]));
  if constexpr (!FUSED) st4(o + 8, make_float4(rgb[1], rgb[2], __uint_as_float(rect), 0.f));
  else { c1 = rgb[1]; c2 = rgb[2]; }   // third 16-B store follows once the group scan is known
  if (radii) radii[g] = radius;
  }  // g < N
  if constexpr (FUSED) {
    GroupScan& S = *Sp;
    int x0, y0, w, h, my_off;
    unpack_rect(rect, x0, y0, w, h);
    const int total = group_scan_store(S, w * h, x0, y0, w, __float_as_uint(tz), my_off, cam.long_run);
    if (g < N) {
      float* o = splats + (size_t)g * TGS_SPLAT_FLOATS;
      st4(o + 8, make_float4(c1, c2, __uint_as_float(rect), __int_as_float(my_off)));
    }
    group_count_tiles(S, cam.TW, cam.TW * cam.TH, total, group_base, tile_count, rank, status, capacity, sticky, cam.long_run);
  }
}

template <int DEG, bool FUSED, bool PRE>  // DEG = -1: no SH (colours from colors_in or zero)
__device__ __forceinline__ void project_fwd_body(
    const CamK& cam, int N, const float* __restrict__ means, const float* __restrict__ log_scales,
    const float* __restrict__ quats, const float* __restrict__ opac_logit,
    const float* __restrict__ sh, int sh_stride, const float* __restrict__ colors_in,
    float* __restrict__ splats, int32_t* __restrict__ radii, int32_t* __restrict__ group_base,
    int32_t* __restrict__ tile_count, int32_t* __restrict__ rank, int32_t* __restrict__ status,
    long long capacity, int32_t* __restrict__ sticky, GroupScan* Sp) {
  const int g = tgs_group_id() * 256 + threadIdx.x;
  if constexpr (!FUSED) { if (g >= N) return; }
  float m[3] = {0.f, 0.f, 0.f}, ls[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f};
  float ol = 0.f;
  if (g < N) {
    m[0] = means[3 * g]; m[1] = means[3 * g + 1]; m[2] = means[3 * g + 2];
    ls[0] = log_scales[3 * g]; ls[1] = log_scales[3 * g + 1]; ls[2] = log_scales[3 * g + 2];
    const float4 q4 = ld4(quats + 4 * (size_t)g);
    q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
    ol = opac_logit[g];
    if constexpr (DEG >= 0 && !PRE) {
      sh_color<DEG>(m, cam.campos, sh + (size_t)g * sh_stride * 3, rgb);
    } else if (colors_in) {
      rgb[0] = colors_in[3 * g]; rgb[1] = colors_in[3 * g + 1]; rgb[2] = colors_in[3 * g + 2];
    }
  }
  project_fwd_core<FUSED>(cam, N, g, m, ls, q, ol, rgb, splats, radii, group_base, tile_count, rank, status,
                          capacity, sticky, Sp);
}

template <int DEG, bool FUSED>
__global__ __launch_bounds__(256) void k_project_fwd(
    CamK cam, int N, const float* __restrict__ means, const float* __restrict__ log_scales,
    const float* __restrict__ quats, const float* __restrict__ opac_logit,
    const float* __restrict__ sh, int sh_stride, const float* __restrict__ colors_in,
    float* __restrict__ splats, int32_t* __restrict__ radii, int32_t* __restrict__ group_base,
    int32_t* __restrict__ tile_count, int32_t* __restrict__ rank, int32_t* __restrict__ status,
    long long capacity, int32_t* __restrict__ sticky) {
  __shared__ GroupScan S;
  project_fwd_body<DEG, FUSED, false>(cam, N, means, log_scales, quats, opac_logit, sh, sh_stride, colors_in,
                                      splats, radii, group_base, tile_count, rank, status, capacity, sticky,
                                      FUSED ? &S : nullptr);
}

// K1 + count with the colours evaluated ahead of time by the previous step's optimizer kernel
// (tgs_project_bwd_adam_next): valid iff that kernel ran to its end, which it records in the tag word.
// The choice is made ONCE, around the whole body -- a branch around the SH evaluation alone keeps
// the compiler from issuing the coefficient loads up front with the other parameter loads
// (52 instead of 78 VGPRs, K1 71 -> 83 us on the fallback path).
template <int DEG>
__global__ __launch_bounds__(256) void k_project_fwd_colors(
    CamK cam, int N, const float* __restrict__ means, const float* __restrict__ log_scales,
    const float* __restrict__ quats, const float* __restrict__ opac_logit,
    const float* __restrict__ sh, int sh_stride, const float* __restrict__ colors_in,
    float* __restrict__ splats, int32_t* __restrict__ radii, int32_t* __restrict__ group_base,
    int32_t* __restrict__ tile_count, int32_t* __restrict__ rank, int32_t* __restrict__ status,
    long long capacity, int32_t* __restrict__ sticky, const int32_t* __restrict__ color_tag,
    int32_t tag_expect) {
  __shared__ GroupScan S;
  if (*color_tag == tag_expect)
    project_fwd_body<DEG, true, true>(cam, N, means, log_scales, quats, opac_logit, sh, sh_stride, colors_in,
                                      splats, radii, group_base, tile_count, rank, status, capacity, sticky, &S);
  else
    project_fwd_body<DEG, true, false>(cam, N, means, log_scales, quats, opac_logit, sh, sh_stride, colors_in,
                                       splats, radii, group_base, tile_count, rank, status, capacity, sticky, &S);
}

// Data-parallel step: Adam on the 11 geometry parameters of every Gaussian from the all-reduced gradients AND the
// next view's K1 on the result (front prefetch of the data-parallel form: there the optimizer is not fused with K8,
// and the SH rows were stepped just before by the gathered-SH kernel, so the colour comes from the SH rows as in the
// stand-alone K1).  Replaces k_adam on the geometry segments + the next step's k_project_fwd; same adam1, same
// project_fwd_core on the same values: bit-identical to the two-kernel sequence.
template <int DEG>
__global__ __launch_bounds__(256) void k_adam_geom_project_next(
    CamK cam, int N, int sh_stride, float* __restrict__ means, float* __restrict__ log_scales,
    float* __restrict__ quats, float* __restrict__ opac_logit, const float* __restrict__ sh,
    const float* __restrict__ grads, AdamK ad_in, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
    float* __restrict__ splats, int32_t* __restrict__ radii, int32_t* __restrict__ group_base,
    int32_t* __restrict__ tile_count, int32_t* __restrict__ rank, int32_t* __restrict__ status,
    long long capacity, int32_t* __restrict__ sticky, int32_t* __restrict__ tag, int32_t tag_value) {
  __shared__ GroupScan S;
  if (ad_in.guard && ad_in.guard[1]) return;   // a rank's frame overflowed: no update, the tag keeps its old value
  const AdamK ad = adam_resolve(ad_in);
  const int g = tgs_group_id() * 256 + threadIdx.x;
  float m[3] = {0.f, 0.f, 0.f}, ls[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f};
  float ol = 0.f;
  if (g < N) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const long long e = 3ll * g + j;
      float P = means[3 * g + j], M = exp_avg[e], V = exp_avg_sq[e];
      adam1(ad, ad.lr_means, P, grads[e], M, V);
      means[3 * g + j] = P; exp_avg[e] = M; exp_avg_sq[e] = V;
      m[j] = P;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const long long e = ad.e_means + 3ll * g + j;
      float P = log_scales[3 * g + j], M = exp_avg[e], V = exp_avg_sq[e];
      adam1(ad, ad.lr_scales, P, grads[e], M, V);
      log_scales[3 * g + j] = P; exp_avg[e] = M; exp_avg_sq[e] = V;
      ls[j] = P;
    }
    {
      const long long e = ad.e_scales + 4ll * g;
      float4 Q = ld4(quats + 4 * (size_t)g), G = ld4_nt(grads + e), M = ld4_nt(exp_avg + e), V = ld4_nt(exp_avg_sq + e);
      adam1(ad, ad.lr_quats, Q.x, G.x, M.x, V.x); adam1(ad, ad.lr_quats, Q.y, G.y, M.y, V.y);
      adam1(ad, ad.lr_quats, Q.z, G.z, M.z, V.z); adam1(ad, ad.lr_quats, Q.w, G.w, M.w, V.w);
      st4(quats + 4 * (size_t)g, Q); st4_nt(exp_avg + e, M); st4_nt(exp_avg_sq + e, V);
      q[0] = Q.x; q[1] = Q.y; q[2] = Q.z; q[3] = Q.w;
    }
    {
      const long long e = ad.e_quats + g;
      float P = opac_logit[g], M = exp_avg[e], V = exp_avg_sq[e];
      adam1(ad, ad.lr_opac, P, grads[e], M, V);
      opac_logit[g] = P; exp_avg[e] = M; exp_avg_sq[e] = V;
      ol = P;
    }
    sh_color<DEG>(m, cam.campos, sh + (size_t)g * sh_stride * 3, rgb);
  }
  project_fwd_core<true>(cam, N, g, m, ls, q, ol, rgb, splats, radii, group_base, tile_count, rank, status,
                         capacity, sticky, &S);
  if (g == 0) *tag = tag_value;
}

// ---------------------------------------------------------------------------------------------
// K8a: partials -> one gradient record per Gaussian
// ---------------------------------------------------------------------------------------------
// One Gaussian's run of `hits` partial records, summed by its owner thread in the order k = 0, 1, 2, ...
__device__ __forceinline__ void sum_run_serial(const float* __restrict__ p, int hits, float* v) {
  // 4 records (12 independent 16-B loads) in flight per round: the loop is latency-bound
  // otherwise.  Accumulation order is still k = 0, 1, 2, ... (bit-identical to a serial loop).
  for (int k0 = 0; k0 < hits; k0 += 4, p += 4 * TGS_PARTIAL_FLOATS) {
    float4 a[4], b[4];
    float2 c[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (k0 + u < hits) {
        a[u] = ld4_nt(p + u * TGS_PARTIAL_FLOATS);
        b[u] = ld4_nt(p + u * TGS_PARTIAL_FLOATS + 4);
        c[u] = *reinterpret_cast<const float2*>(p + u * TGS_PARTIAL_FLOATS + 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (k0 + u < hits) {
        v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w;
        v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
        v[8] += c[u].x; v[9] += c[u].y;
      }
    }
  }
}

// Segmented sum of the partial records of a workgroup's 256 Gaussians (round 6).
//
// A Gaussian's records are contiguous (one per tile of its rect) and until round 5 its owner thread added them
// one after the other.  That is a latency chain of hits / 4 dependent round trips: fine for the ~4 records of a
// cfg3 Gaussian, but a converged object-centric model (the reference's kind of scene: an object on a table at
// 1280 x 720) has table / background Gaussians that cover 1000 - 3600 tiles, and ONE thread's chain of 900 round
// trips then IS the kernel: K8 150 - 400 us for 79 k Gaussians against 120 us for cfg3's 1 M
// (profiles/r5_train_quality.json, r6_before_kernel_stats_*_720p.csv).  Now runs longer than TGS_LONG_RUN are cut
// into segments of TGS_RUN_SEG records and the workgroup's sixteen 16-lane teams take 16 segments per round:
// lane q of a team adds records q, q + 16, q + 32, q + 48 of its segment (all twelve loads in flight), a DPP tree
// over the team's 16 lanes forms the segment's sum, and the owner thread adds its segments' sums in order.  The
// shape of every sum depends on `hits` alone -- not on which Gaussians share the group, nor on scheduling -- so the
// result is bit-reproducible and independent of the row order, as before.  Groups without a long run (all of a
// uniform scene's) leave through one __syncthreads_or.
#define TGS_RUN_SEG 64     // (TGS_LONG_RUN: tgs_binning.h)
struct RunScan {
  int seg_off[TGS_GROUP + 1];   // exclusive scan of the Gaussians' segment counts (0 for short runs)
  int base[TGS_GROUP];          // first pair index of the Gaussian's run (capacity < 2^31)
  int hits[TGS_GROUP];
  int wave_tot[TGS_GROUP / TGS_WAVE];
  float seg_sum[2][16][12];     // the teams' results of a round, double buffered (10 of 12 used)
};

template <int CTRL>
__device__ __forceinline__ float row_dpp_add(float v) {
  const int s = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, s);
}
// sum over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ float row_sum16(float v) {
  v = row_dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = row_dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = row_dpp_add<0x141>(v);   // row_half_mirror
  v = row_dpp_add<0x140>(v);   // row_mirror
  return v;
}

// Must be called by all TGS_GROUP threads of the workgroup (g >= N: no Gaussian); S may alias LDS that is reused
// afterwards (the function ends with a barrier whenever it has touched S).
__device__ __forceinline__ void group_sum_partials(const float* __restrict__ splats,
                                                   const int32_t* __restrict__ group_base,
                                                   const float* __restrict__ partials, int g, int N, float* v,
                                                   RunScan& S, int long_run) {
#pragma unroll
  for (int i = 0; i < 10; i++) v[i] = 0.f;
  const int tid = threadIdx.x;
  int hits = 0, base = 0;
  if (g < N) {
    const float4 r2 = ld4(splats + (size_t)g * TGS_SPLAT_FLOATS + 8);
    int x0, y0, w, h;
    unpack_rect(__float_as_uint(r2.z), x0, y0, w, h);
    hits = w * h;
    if (hits) base = group_base[g / TGS_GROUP] + __float_as_int(r2.w);
  }
  const bool big = hits > long_run;
  if (!__syncthreads_or(big)) {        // nothing long in this group
    sum_run_serial(partials + (size_t)base * TGS_PARTIAL_FLOATS, hits, v);
    return;
  }
  const int nseg = big ? (hits + TGS_RUN_SEG - 1) / TGS_RUN_SEG : 0;
  int incl = nseg;
#pragma unroll
  for (int o = 1; o < TGS_WAVE; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if ((tid & (TGS_WAVE - 1)) >= o) incl += t;
  }
  if ((tid & (TGS_WAVE - 1)) == TGS_WAVE - 1) S.wave_tot[tid / TGS_WAVE] = incl;
  S.base[tid] = base;
  S.hits[tid] = hits;
  __syncthreads();
  int my_off = incl - nseg, total = 0;
#pragma unroll
  for (int i = 0; i < TGS_GROUP / TGS_WAVE; i++) {
    if (i < tid / TGS_WAVE) my_off += S.wave_tot[i];
    total += S.wave_tot[i];
  }
  S.seg_off[tid] = my_off;
  if (tid == 0) S.seg_off[TGS_GROUP] = total;
  if (!big) sum_run_serial(partials + (size_t)base * TGS_PARTIAL_FLOATS, hits, v);   // the short runs, as ever
  __syncthreads();
  const int team = tid >> 4, q = tid & 15;
  for (int s0 = 0, buf = 0; s0 < total; s0 += 16, buf ^= 1) {
    const int s = s0 + team;
    if (s < total) {                   // uniform over the team's DPP row
      // owner of segment s: the last j with seg_off[j] <= s (entries without segments repeat their successor's offset)
      int lo = 0, hi = TGS_GROUP;
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int mid = (lo + hi) >> 1;
        if (S.seg_off[mid] <= s) lo = mid; else hi = mid;
      }
      const int k0 = (s - S.seg_off[lo]) * TGS_RUN_SEG;
      const int cnt = min(TGS_RUN_SEG, S.hits[lo] - k0);
      const float* p = partials + ((size_t)S.base[lo] + (size_t)(k0 + q)) * TGS_PARTIAL_FLOATS;
      float4 a[4], b[4];
      float2 c[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (q + 16 * u < cnt) {
          a[u] = ld4_nt(p + 16 * u * TGS_PARTIAL_FLOATS);
          b[u] = ld4_nt(p + 16 * u * TGS_PARTIAL_FLOATS + 4);
          c[u] = *reinterpret_cast<const float2*>(p + 16 * u * TGS_PARTIAL_FLOATS + 8);
        }
      }
      float t[10];
#pragma unroll
      for (int i = 0; i < 10; i++) t[i] = 0.f;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (q + 16 * u < cnt) {
          t[0] += a[u].x; t[1] += a[u].y; t[2] += a[u].z; t[3] += a[u].w;
          t[4] += b[u].x; t[5] += b[u].y; t[6] += b[u].z; t[7] += b[u].w;
          t[8] += c[u].x; t[9] += c[u].y;
        }
      }
#pragma unroll
      for (int i = 0; i < 10; i++) t[i] = row_sum16(t[i]);
      if (q == 0) {
        float* o = S.seg_sum[buf][team];
        st4(o, make_float4(t[0], t[1], t[2], t[3]));
        st4(o + 4, make_float4(t[4], t[5], t[6], t[7]));
        *reinterpret_cast<float2*>(o + 8) = make_float2(t[8], t[9]);
      }
    }
    __syncthreads();
    if (big) {                          // my segments of this round, in order
      const int e = min(my_off + nseg, s0 + 16);
      for (int sg = max(my_off, s0); sg < e; sg++) {
        const float* o = S.seg_sum[buf][sg - s0];
        const float4 x = ld4(o), y = ld4(o + 4);
        const float2 z = *reinterpret_cast<const float2*>(o + 8);
        v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
        v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
        v[8] += z.x; v[9] += z.y;
      }
    }
  }
  __syncthreads();   // S may be reused by the caller
}

__global__ __launch_bounds__(256) void k_reduce_partials(
    CamK cam, int N, const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const float* __restrict__ partials, float* __restrict__ v_splats) {
  __shared__ RunScan S;
  const int g = blockIdx.x * 256 + threadIdx.x;
  float v[10];
  group_sum_partials(splats, group_base, partials, g, N, v, S, cam.long_run);
  if (g >= N) return;
  float* o = v_splats + (size_t)g * TGS_SPLAT_FLOATS;
  st4(o, make_float4(v[0], v[1], v[2], v[3]));
  st4(o + 4, make_float4(v[4], v[5], v[6], v[7]));
  st4(o + 8, make_float4(v[8], v[9], 0.f, 0.f));
}

// B.8 geometry backward for one visible Gaussian: v = {v_x, v_y, v_depth, -, v_a, v_b, v_c, ...}
// -> accumulates into vm (means), writes vls (log-scales) and vq (quaternion).
__device__ __forceinline__ void geom_bwd(const CamK& cam, const float* m, const float* ls, const float* q,
                                         const float* v, float* vm, float* vls, float* vq) {
    Geom G;
    geom_eval(cam, m, ls, q, G);
    const float id = 1.0f / G.det;
    const float a = G.c11 * id, b = -G.c01 * id, c = G.c00 * id;
    const float va = v[4], vb = v[5], vc = v[6];
    // gradient w.r.t. the 2x2 covariance as a full symmetric matrix [[G00,G01],[G01,G11]]
    const float G00 = -(a * a * va + a * b * vb + b * b * vc);
    const float G11 = -(b * b * va + b * c * vb + c * c * vc);
    const float G01 = -0.5f * (2.f * a * b * va + (a * c + b * b) * vb + 2.f * b * c * vc);
    // GT = Gm * Tm (2x3)
    float GT[6];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      GT[cc] = G00 * G.Tm[cc] + G01 * G.Tm[3 + cc];
      GT[3 + cc] = G01 * G.Tm[cc] + G11 * G.Tm[3 + cc];
    }
    // v_Sigma = Tm^T * GT (3x3 symmetric)
    float vS[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) vS[3 * r + cc] = G.Tm[r] * GT[cc] + G.Tm[3 + r] * GT[3 + cc];
    // v_Tm = 2 * GT * Sigma
    const float S9[9] = {G.Sig[0], G.Sig[1], G.Sig[2], G.Sig[1], G.Sig[3], G.Sig[4],
                         G.Sig[2], G.Sig[4], G.Sig[5]};
    float vTm[6];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++)
        vTm[3 * r + cc] = 2.f * (GT[3 * r] * S9[cc] + GT[3 * r + 1] * S9[3 + cc] + GT[3 * r + 2] * S9[6 + cc]);
    // v_J = v_Tm * R_world^T; only J00, J02, J11, J12 are non-constant
    const float* R = cam.R;
    const float vJ00 = vTm[0] * R[0] + vTm[1] * R[1] + vTm[2] * R[2];
    const float vJ02 = vTm[0] * R[6] + vTm[1] * R[7] + vTm[2] * R[8];
    const float vJ11 = vTm[3] * R[3] + vTm[4] * R[4] + vTm[5] * R[5];
    const float vJ12 = vTm[3] * R[6] + vTm[4] * R[7] + vTm[5] * R[8];
    const float rz = 1.0f / G.tz, rz2 = rz * rz;
    float vt[3] = {0.f, 0.f, 0.f};
    vt[2] += -(vJ00 * cam.fx + vJ11 * cam.fy) * rz2;
    if (G.inx) { vt[0] += -vJ02 * cam.fx * rz2; vt[2] += 2.f * vJ02 * cam.fx * G.tx * rz2 * rz; }
    else vt[2] += vJ02 * cam.fx * G.ucx * rz2;
    if (G.iny) { vt[1] += -vJ12 * cam.fy * rz2; vt[2] += 2.f * vJ12 * cam.fy * G.ty * rz2 * rz; }
    else vt[2] += vJ12 * cam.fy * G.ucy * rz2;
    // mean2d and depth
    vt[0] += v[0] * cam.fx * rz; vt[2] += -v[0] * cam.fx * G.tx * rz2;
    vt[1] += v[1] * cam.fy * rz; vt[2] += -v[1] * cam.fy * G.ty * rz2;
    vt[2] += v[2];
#pragma unroll
    for (int j = 0; j < 3; j++) vm[j] += R[j] * vt[0] + R[3 + j] * vt[1] + R[6 + j] * vt[2];
    // Sigma = M M^T  ->  v_M = 2 v_Sigma M,  M = Rq diag(s)
    float vR[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float vMj[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
        vMj[r] = 2.f * (vS[3 * r] * G.Rq[j] + vS[3 * r + 1] * G.Rq[3 + j] + vS[3 * r + 2] * G.Rq[6 + j]) * G.s[j];
      const float vs = G.Rq[j] * vMj[0] + G.Rq[3 + j] * vMj[1] + G.Rq[6 + j] * vMj[2];
      vls[j] = vs * G.s[j];
      vR[j] = vMj[0] * G.s[j]; vR[3 + j] = vMj[1] * G.s[j]; vR[6 + j] = vMj[2] * G.s[j];
    }
    const float w = G.qn[0], x = G.qn[1], y = G.qn[2], z = G.qn[3];
    float vqn[4];
    vqn[0] = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
    vqn[1] = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
    vqn[2] = 2.f * (x * (vR[1] + vR[3]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
    vqn[3] = 2.f * (x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
    const float dot = w * vqn[0] + x * vqn[1] + y * vqn[2] + z * vqn[3];
    const float iq = 1.0f / G.qnorm;
#pragma unroll
    for (int j = 0; j < 4; j++) vq[j] = (vqn[j] - G.qn[j] * dot) * iq;
}

// ---------------------------------------------------------------------------------------------
// K8 backward
// ---------------------------------------------------------------------------------------------
template <int DEG>
__global__ __launch_bounds__(256) void k_project_bwd(
    CamK cam, int N, const float* __restrict__ means, const float* __restrict__ log_scales,
    const float* __restrict__ quats, const float* __restrict__ opac_logit,
    const float* __restrict__ sh, int sh_stride, const float* __restrict__ splats,
    const int32_t* __restrict__ group_base, const float* __restrict__ partials,
    const float* __restrict__ v_splats, float* __restrict__ v_means,
    float* __restrict__ v_log_scales, float* __restrict__ v_quats,
    float* __restrict__ v_opac_logit, float* __restrict__ v_sh, float* __restrict__ v_color,
    float* __restrict__ v_xy, const int32_t* __restrict__ guard) {
  __shared__ RunScan S;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (guard && guard[1]) {  // overflowed frame: the pair index space is not backed by memory
    if (v_color && g == 0) v_color[3 * (size_t)N + 3] = 1.f;   // tell the other ranks (tgs_dp_agree_overflow)
    return;
  }
  if (v_color && g == 0) {  // trailer of the colour-gradient block: this view's camera position
    v_color[3 * (size_t)N] = cam.campos[0]; v_color[3 * (size_t)N + 1] = cam.campos[1];
    v_color[3 * (size_t)N + 2] = cam.campos[2]; v_color[3 * (size_t)N + 3] = 0.f;
  }
  // per-Gaussian upstream gradient: {v_x, v_y, v_depth, v_opac, v_a, v_b, v_c, v_r, v_g, v_b}
  float v[10];
  if (partials) group_sum_partials(splats, group_base, partials, g, N, v, S, cam.long_run);   // (workgroup-uniform: every thread calls)
  if (g >= N) return;
  if (!partials) {
    const float* p = v_splats + (size_t)g * TGS_SPLAT_FLOATS;
    const float4 a = ld4(p), b = ld4(p + 4);
    const float2 c = *reinterpret_cast<const float2*>(p + 8);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z;
    v[7] = b.w; v[8] = c.x; v[9] = c.y;
  }
  const float m[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
  float vm[3] = {0.f, 0.f, 0.f};

  v_opac_logit[g] = v[3] * sigmoid_deriv(opac_logit[g]);

  // ---- SH colour backward (B.5) ----
  if constexpr (DEG >= 0) {
    constexpr int K = (DEG + 1) * (DEG + 1);
    float dx = m[0] - cam.campos[0], dy = m[1] - cam.campos[1], dz = m[2] - cam.campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float d[3] = {dx * inv, dy * inv, dz * inv};
    float Y[16];
    sh_basis<DEG>(d[0], d[1], d[2], Y);
    const float* c = sh + (size_t)g * sh_stride * 3;
    float col[3] = {0.5f, 0.5f, 0.5f};
    float ck[K][3];
#pragma unroll
    for (int k = 0; k < K; k++) {
#pragma unroll
      for (int ch = 0; ch < 3; ch++) { ck[k][ch] = c[3 * k + ch]; col[ch] += Y[k] * ck[k][ch]; }
    }
    float vr[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) vr[ch] = (col[ch] > 0.f) ? v[7 + ch] : 0.f;  // clamp gate
    if (v_color) { v_color[3 * g] = vr[0]; v_color[3 * g + 1] = vr[1]; v_color[3 * g + 2] = vr[2]; }
    if (v_sh) {
      float* o_sh = v_sh + (size_t)g * sh_stride * 3;
#pragma unroll
      for (int k = 0; k < K; k++) {
        o_sh[3 * k] = Y[k] * vr[0]; o_sh[3 * k + 1] = Y[k] * vr[1]; o_sh[3 * k + 2] = Y[k] * vr[2];
      }
      for (int k = K; k < sh_stride; k++) { o_sh[3 * k] = 0.f; o_sh[3 * k + 1] = 0.f; o_sh[3 * k + 2] = 0.f; }
    }
    if constexpr (DEG >= 1) {
      float dY[K][3];
      sh_basis_grad<DEG>(d[0], d[1], d[2], dY);
      float vd[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 1; k < K; k++) {
        const float s = ck[k][0] * vr[0] + ck[k][1] * vr[1] + ck[k][2] * vr[2];
        vd[0] += dY[k][0] * s; vd[1] += dY[k][1] * s; vd[2] += dY[k][2] * s;
      }
      const float dot = d[0] * vd[0] + d[1] * vd[1] + d[2] * vd[2];
#pragma unroll
      for (int j = 0; j < 3; j++) vm[j] += (vd[j] - d[j] * dot) * inv;
    }
  } else {
    if (v_sh) {
      float* o_sh = v_sh + (size_t)g * sh_stride * 3;
      for (int k = 0; k < sh_stride * 3; k++) o_sh[k] = 0.f;
    }
    if (v_color) { v_color[3 * g] = 0.f; v_color[3 * g + 1] = 0.f; v_color[3 * g + 2] = 0.f; }
  }

  // ---- geometry backward (B.8) ----
  float vls[3] = {0.f, 0.f, 0.f};
  float vq[4] = {0.f, 0.f, 0.f, 0.f};
  // a Gaussian passed the App. B culls iff its conic was written (a > 0)
  const bool visible = splats[(size_t)g * TGS_SPLAT_FLOATS + 4] > 0.f;
  if (visible) {
    const float ls[3] = {log_scales[3 * g], log_scales[3 * g + 1], log_scales[3 * g + 2]};
    const float4 q4 = ld4(quats + 4 * (size_t)g);
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    geom_bwd(cam, m, ls, q, v, vm, vls, vq);
  }
  v_means[3 * g] = vm[0]; v_means[3 * g + 1] = vm[1]; v_means[3 * g + 2] = vm[2];
  v_log_scales[3 * g] = vls[0]; v_log_scales[3 * g + 1] = vls[1]; v_log_scales[3 * g + 2] = vls[2];
  st4(v_quats + 4 * (size_t)g, make_float4(vq[0], vq[1], vq[2], vq[3]));
  if (v_xy) {
    v_xy[2 * g] = v[0]; v_xy[2 * g + 1] = v[1];
  }
}

// ---------------------------------------------------------------------------------------------
// K8 with LDS-staged SH rows (and optionally Adam fused in)
// ---------------------------------------------------------------------------------------------
// A workgroup owns 256 consecutive Gaussians whose SH rows are one contiguous 256*3K-float block.
// It is moved with fully coalesced 16-B accesses into an LDS image [256][3K+4] (the +4 pad makes
// the per-thread ds_read_b128 of a row conflict-free), each thread reads its own row from LDS,
// writes its SH gradient row back into the same LDS image, and the block then streams the image
// out coalesced -- either as v_sh, or (FUSE_ADAM) straight through the Adam update of sh/m/v, so
// that the 12K-float SH gradient never touches HBM.  The 11 non-SH parameters per Gaussian are
// updated by their owner thread.
// COLOR_ONLY (data-parallel training): `v_sh` receives the clamp-gated colour gradient [N,3] instead
// of the SH gradient image; nothing is streamed out at the end.
// DEG is the ACTIVE SH degree, KS the number of bases each Gaussian STORES (KS >= (DEG+1)^2, 3*KS a
// multiple of 4): while the trainer ramps the degree up, the rows above the active degree receive a
// zero gradient (and, fused, a plain Adam step on it).
template <int DEG, int KS, bool FUSE_ADAM, bool COLOR_ONLY = false>
__global__ __launch_bounds__(256) void k_project_bwd_lds(
    CamK cam, int N, float* __restrict__ means, float* __restrict__ log_scales,
    float* __restrict__ quats, float* __restrict__ opac_logit, float* __restrict__ sh,
    const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const float* __restrict__ partials, float* __restrict__ v_means,
    float* __restrict__ v_log_scales, float* __restrict__ v_quats,
    float* __restrict__ v_opac_logit, float* __restrict__ v_sh, float* __restrict__ v_xy,
    AdamK ad_in, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq, NextView nx) {
  // overflowed frame: the pair index space is not backed by memory and the step must not touch the model
  // (nor is the next view's colour prefetch produced: its tag word keeps its old value)
  if (ad_in.guard && ad_in.guard[1]) {
    // colour mode (v_sh is the rank's colour-gradient block): the pad slot carries the overflow flag
    if (COLOR_ONLY && blockIdx.x == 0 && threadIdx.x == 0) v_sh[3 * (size_t)N + 3] = 1.f;
    return;
  }
  const AdamK ad = FUSE_ADAM ? adam_resolve(ad_in) : ad_in;
  constexpr int K = (DEG + 1) * (DEG + 1);   // active bases; sh_stride == KS here
  static_assert(KS >= K && (3 * KS) % 4 == 0, "storage must hold the active degree in whole float4s");
  static_assert(256 * (3 * KS + 4) * sizeof(float) >= sizeof(GroupScan), "the group scan of the front prefetch reuses the SH image");
  constexpr int ROW = 3 * KS, RS = ROW + 4, F4 = ROW / 4;
  extern __shared__ float4 lds4[];
  float* lds = reinterpret_cast<float*>(lds4);
  const int tid = threadIdx.x;
  const int g0 = blockIdx.x * 256;
  const int g = g0 + tid;
  const int nrows = min(256, N - g0);
  const size_t blk = (size_t)g0 * ROW;

  // per-Gaussian upstream gradient {v_x, v_y, v_depth, v_opac, v_a, v_b, v_c, v_r, v_g, v_b}: the segmented sum of the
  // group's partial records, long runs shared by the whole workgroup.  Its scratch aliases the (not yet loaded) SH image.
  static_assert(256 * (3 * KS + 4) * sizeof(float) >= sizeof(RunScan), "the run scan reuses the SH image");
  float v[10];
  group_sum_partials(splats, group_base, partials, g, N, v, *reinterpret_cast<RunScan*>(lds4), cam.long_run);
  {
    const int nf = nrows * F4;
    for (int f0 = tid; f0 < nf; f0 += 256 * 4) {
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (f0 + 256 * u < nf) {
          // fused: the row is read again by the Adam stream below, keep it cached; otherwise this is
          // the only read of the step
          const float* src = sh + blk + 4 * (size_t)(f0 + 256 * u);
          t[u] = FUSE_ADAM ? ld4(src) : ld4_nt(src);
        }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int f = f0 + 256 * u;
        if (f < nf) { const int row = f / F4, c4 = f - row * F4; st4(lds + row * RS + 4 * c4, t[u]); }
      }
    }
  }
  __syncthreads();

  float vm[3] = {0.f, 0.f, 0.f}, vls[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f};
  float m[3] = {0.f, 0.f, 0.f}, ls[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f};
  float ol = 0.f, vol = 0.f;
  if constexpr (COLOR_ONLY) {
    if (g == 0) {  // trailer of the colour-gradient block: this view's camera position
      v_sh[3 * (size_t)N] = cam.campos[0]; v_sh[3 * (size_t)N + 1] = cam.campos[1];
      v_sh[3 * (size_t)N + 2] = cam.campos[2]; v_sh[3 * (size_t)N + 3] = 0.f;
    }
  }
  if (g < N) {
    m[0] = means[3 * g]; m[1] = means[3 * g + 1]; m[2] = means[3 * g + 2];
    ol = opac_logit[g];
    vol = v[3] * sigmoid_deriv(ol);
    // ---- SH colour backward (B.5), coefficients from / gradients to the LDS row ----
    float dx = m[0] - cam.campos[0], dy = m[1] - cam.campos[1], dz = m[2] - cam.campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float d[3] = {dx * inv, dy * inv, dz * inv};
    float Y[16];
    sh_basis<DEG>(d[0], d[1], d[2], Y);
    float* row = lds + tid * RS;
    float ck[ROW];
#pragma unroll
    for (int i = 0; i < F4; i++) {
      const float4 t = ld4(row + 4 * i);
      ck[4 * i] = t.x; ck[4 * i + 1] = t.y; ck[4 * i + 2] = t.z; ck[4 * i + 3] = t.w;
    }
    float col[3] = {0.5f, 0.5f, 0.5f};
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
      for (int ch = 0; ch < 3; ch++) col[ch] += Y[k] * ck[3 * k + ch];
    float vr[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) vr[ch] = (col[ch] > 0.f) ? v[7 + ch] : 0.f;  // clamp gate
    if constexpr (DEG >= 1) {
      float dY[K][3];
      sh_basis_grad<DEG>(d[0], d[1], d[2], dY);
      float vd[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 1; k < K; k++) {
        const float s = ck[3 * k] * vr[0] + ck[3 * k + 1] * vr[1] + ck[3 * k + 2] * vr[2];
        vd[0] += dY[k][0] * s; vd[1] += dY[k][1] * s; vd[2] += dY[k][2] * s;
      }
      const float dot = d[0] * vd[0] + d[1] * vd[1] + d[2] * vd[2];
#pragma unroll
      for (int j = 0; j < 3; j++) vm[j] += (vd[j] - d[j] * dot) * inv;
    }
    if constexpr (COLOR_ONLY) { v_sh[3 * g] = vr[0]; v_sh[3 * g + 1] = vr[1]; v_sh[3 * g + 2] = vr[2]; }
    // own row <- SH gradient (only this thread ever touches this row before the barrier)
#pragma unroll
    for (int i = 0; i < (COLOR_ONLY ? 0 : F4); i++) {
      float4 t;   // element e of the row belongs to basis e / 3, channel e % 3; inactive bases get 0
      t.x = (4 * i) / 3 < K ? Y[(4 * i) / 3] * vr[(4 * i) % 3] : 0.f;
      t.y = (4 * i + 1) / 3 < K ? Y[(4 * i + 1) / 3] * vr[(4 * i + 1) % 3] : 0.f;
      t.z = (4 * i + 2) / 3 < K ? Y[(4 * i + 2) / 3] * vr[(4 * i + 2) % 3] : 0.f;
      t.w = (4 * i + 3) / 3 < K ? Y[(4 * i + 3) / 3] * vr[(4 * i + 3) % 3] : 0.f;
      st4(row + 4 * i, t);
    }
    // ---- geometry backward (B.8) ----
    if (splats[(size_t)g * TGS_SPLAT_FLOATS + 4] > 0.f) {
      ls[0] = log_scales[3 * g]; ls[1] = log_scales[3 * g + 1]; ls[2] = log_scales[3 * g + 2];
      const float4 q4 = ld4(quats + 4 * (size_t)g);
      q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
      geom_bwd(cam, m, ls, q, v, vm, vls, vq);
    } else if (FUSE_ADAM) {
      ls[0] = log_scales[3 * g]; ls[1] = log_scales[3 * g + 1]; ls[2] = log_scales[3 * g + 2];
      const float4 q4 = ld4(quats + 4 * (size_t)g);
      q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
    }
    if (v_xy) { v_xy[2 * g] = v[0]; v_xy[2 * g + 1] = v[1]; }
    if constexpr (!FUSE_ADAM) {
      v_means[3 * g] = vm[0]; v_means[3 * g + 1] = vm[1]; v_means[3 * g + 2] = vm[2];
      v_log_scales[3 * g] = vls[0]; v_log_scales[3 * g + 1] = vls[1]; v_log_scales[3 * g + 2] = vls[2];
      st4(v_quats + 4 * (size_t)g, make_float4(vq[0], vq[1], vq[2], vq[3]));
      v_opac_logit[g] = vol;
    } else {
      // Adam on the 11 non-SH parameters of this Gaussian (flat-buffer offsets from the layout)
      float* ea = exp_avg; float* es = exp_avg_sq;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const long long e = 3ll * g + j;
        float pm = m[j], M = ea[e], V = es[e];
        adam1(ad, ad.lr_means, pm, vm[j], M, V);
        means[3 * g + j] = pm; ea[e] = M; es[e] = V;
        m[j] = pm;   // the colour prefetch below looks from the NEXT camera at the UPDATED mean
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const long long e = ad.e_means + 3ll * g + j;
        float pl = ls[j], M = ea[e], V = es[e];
        adam1(ad, ad.lr_scales, pl, vls[j], M, V);
        log_scales[3 * g + j] = pl; ea[e] = M; es[e] = V;
        ls[j] = pl;
      }
      {
        const long long e = ad.e_scales + 4ll * g;
        float4 M = ld4_nt(ea + e), V = ld4_nt(es + e);
        float4 Q = make_float4(q[0], q[1], q[2], q[3]);
        adam1(ad, ad.lr_quats, Q.x, vq[0], M.x, V.x); adam1(ad, ad.lr_quats, Q.y, vq[1], M.y, V.y);
        adam1(ad, ad.lr_quats, Q.z, vq[2], M.z, V.z); adam1(ad, ad.lr_quats, Q.w, vq[3], M.w, V.w);
        st4_nt(quats + 4 * (size_t)g, Q); st4_nt(ea + e, M); st4_nt(es + e, V);
        q[0] = Q.x; q[1] = Q.y; q[2] = Q.z; q[3] = Q.w;
      }
      {
        const long long e = ad.e_quats + g;
        float po = ol, M = ea[e], V = es[e];
        adam1(ad, ad.lr_opac, po, vol, M, V);
        opac_logit[g] = po; ea[e] = M; es[e] = V;
        ol = po;
      }
    }
  }
  if constexpr (COLOR_ONLY) return;
  __syncthreads();
  // coalesced stream of the block's SH gradient image (2 float4 columns per thread per round so
  // that 6 independent loads are in flight)
  const int nf = nrows * F4;
  for (int f0 = tid; f0 < nf; f0 += 256 * 2) {
    if constexpr (!FUSE_ADAM) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int f = f0 + 256 * u;
        if (f < nf) { const int row = f / F4, c4 = f - row * F4; st4(v_sh + blk + 4 * (size_t)f, ld4(lds + row * RS + 4 * c4)); }
      }
    } else {
      float4 P[2], M[2], V[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int f = f0 + 256 * u;
        if (f < nf) {
          const size_t e = blk + 4 * (size_t)f;
          P[u] = ld4_nt(sh + e); M[u] = ld4_nt(exp_avg + ad.e_opac + e); V[u] = ld4_nt(exp_avg_sq + ad.e_opac + e);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int f = f0 + 256 * u;
        if (f < nf) {
          const int row = f / F4, c4 = f - row * F4;
          const float4 G = ld4(lds + row * RS + 4 * c4);
          const size_t e = blk + 4 * (size_t)f;
          const int c = 4 * c4;  // column of the first element inside the 3K-float row; DC = columns 0..2
          adam1(ad, c < 3 ? ad.lr_dc : ad.lr_rest, P[u].x, G.x, M[u].x, V[u].x);
          adam1(ad, c + 1 < 3 ? ad.lr_dc : ad.lr_rest, P[u].y, G.y, M[u].y, V[u].y);
          adam1(ad, c + 2 < 3 ? ad.lr_dc : ad.lr_rest, P[u].z, G.z, M[u].z, V[u].z);
          adam1(ad, ad.lr_rest, P[u].w, G.w, M[u].w, V[u].w);
          st4_nt(sh + e, P[u]); st4_nt(exp_avg + ad.e_opac + e, M[u]); st4_nt(exp_avg_sq + ad.e_opac + e, V[u]);
          if (nx.colors) st4(lds + row * RS + 4 * c4, P[u]);   // updated coefficients replace the consumed gradient
        }
      }
    }
  }
  if constexpr (FUSE_ADAM) {
    // Colour prefetch: the next step's K1 needs, of the 3K updated coefficients, only the colour they
    // give from the next camera -- evaluate it here, where the updated row is on chip, and K1 reads
    // 12 B instead of 12K B per Gaussian (tgs_project_bin_sort_colors).
    if (nx.colors) {
      __syncthreads();
      float rgb[3] = {0.f, 0.f, 0.f};
      if (g < N) {
        sh_color<DEG>(m, nx.campos, lds + tid * RS, rgb);
        nx.colors[3 * g] = rgb[0]; nx.colors[3 * g + 1] = rgb[1]; nx.colors[3 * g + 2] = rgb[2];
      }
      if (nx.fr.splats) {
        // the next view's K1 for this group: same code and the same inputs (the stepped parameters, the colour just
        // evaluated) as the stand-alone kernel would read back from memory -> bit-identical records and counts.
        // The group scan takes over the (consumed) SH image.
        __syncthreads();
        project_fwd_core<true>(nx.fr.cam, N, g, m, ls, q, ol, rgb, nx.fr.splats, nx.fr.radii, nx.fr.group_base,
                               nx.fr.tile_count, nx.fr.rank, nx.fr.status, nx.fr.capacity, nx.fr.sticky,
                               reinterpret_cast<GroupScan*>(lds4));
      }
      if (g == 0) *nx.tag = nx.tag_value;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Adam on the SH block from all-gathered colour gradients (data-parallel step)
// ---------------------------------------------------------------------------------------------
// The SH gradient of rank r is the outer product Y_k(dir_r(g)) * v_color_r[g,:] -- 3 numbers per
// Gaussian and rank plus a basis every rank can evaluate itself (means are replicated, the cameras
// of all ranks are known).  Exchanging v_color (all-gather, 12 B per Gaussian and rank) instead of
// the 12K-float gradient rows (all-reduce) cuts the bytes crossing xGMI ~2.5x at 8 ranks and ~4x
// at 2; the sum over ranks is rebuilt here in rank order (identical on every rank), written into an
// LDS image of the block's rows and streamed through the Adam update fully coalesced, exactly like
// the fused K8+Adam tail.  `means` must still hold the values the forward pass used.  Each rank's
// block carries its camera position behind the N colour gradients (written by K8), so no host data
// is needed.
template <int DEG>
__global__ __launch_bounds__(256) void k_adam_sh_gathered(
    int world, int row0, int N, int sh_stride, const float* __restrict__ means,
    float* __restrict__ sh, const float* __restrict__ v_color_all, AdamK ad_in,
    float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq) {
  // rows [row0, N) of the model; the gathered blocks hold exactly these rows (N - row0 colour gradients each)
  if (ad_in.guard && ad_in.guard[1]) return;   // a rank's frame overflowed (tgs_dp_agree_overflow): no update
  const AdamK ad = adam_resolve(ad_in);
  constexpr int K = (DEG + 1) * (DEG + 1);
  const int ROW = 3 * sh_stride, RS = ROW + 4, F4 = ROW / 4;
  extern __shared__ float4 lds4[];
  float* lds = reinterpret_cast<float*>(lds4);
  const int tid = threadIdx.x;
  const int g0 = row0 + blockIdx.x * 256;
  const int g = g0 + tid;
  const int nrows = min(256, N - g0);
  const size_t blk = (size_t)g0 * ROW;
  if (g < N) {
    const float m0 = means[3 * g], m1 = means[3 * g + 1], m2 = means[3 * g + 2];
    float acc[3 * K];
#pragma unroll
    for (int i = 0; i < 3 * K; i++) acc[i] = 0.f;
    const size_t nb = (size_t)(N - row0);
    const size_t blk_r = 3 * nb + 4;   // one rank's block: v_color[rows,3] | campos[3] | pad
    for (int r = 0; r < world; r++) {
      const float* vc = v_color_all + r * blk_r + 3 * (size_t)(g - row0);
      const float* cp = v_color_all + r * blk_r + 3 * nb;
      const float v0 = vc[0], v1 = vc[1], v2 = vc[2];
      const float dx = m0 - cp[0], dy = m1 - cp[1], dz = m2 - cp[2];
      const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
      float Y[16];
      sh_basis<DEG>(dx * inv, dy * inv, dz * inv, Y);
#pragma unroll
      for (int k = 0; k < K; k++) {
        acc[3 * k] = fmaf(Y[k], v0, acc[3 * k]);
        acc[3 * k + 1] = fmaf(Y[k], v1, acc[3 * k + 1]);
        acc[3 * k + 2] = fmaf(Y[k], v2, acc[3 * k + 2]);
      }
    }
    float* row = lds + tid * RS;
#pragma unroll
    for (int i = 0; i < 3 * K / 4; i++) st4(row + 4 * i, make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]));
    if constexpr ((3 * K) % 4 != 0) {   // K = 1 or 9: finish the last partial float4 with zeros
      float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < (3 * K) % 4; i++) t[i] = acc[(3 * K / 4) * 4 + i];
      st4(row + (3 * K / 4) * 4, make_float4(t[0], t[1], t[2], t[3]));
    }
    for (int i = (3 * K + 3) / 4; i < F4; i++) st4(row + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
  }
  __syncthreads();
  const int nf = nrows * F4;
  for (int f0 = tid; f0 < nf; f0 += 256 * 2) {
    float4 P[2], M[2], V[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int f = f0 + 256 * u;
      if (f < nf) {
        const size_t e = blk + 4 * (size_t)f;
        P[u] = ld4_nt(sh + e); M[u] = ld4_nt(exp_avg + ad.e_opac + e); V[u] = ld4_nt(exp_avg_sq + ad.e_opac + e);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int f = f0 + 256 * u;
      if (f < nf) {
        const int row = f / F4, c4 = f - row * F4;
        const float4 G = ld4(lds + row * RS + 4 * c4);
        const size_t e = blk + 4 * (size_t)f;
        const int c = 4 * c4;  // column inside the 3K-float row; DC = columns 0..2
        adam1(ad, c < 3 ? ad.lr_dc : ad.lr_rest, P[u].x, G.x, M[u].x, V[u].x);
        adam1(ad, c + 1 < 3 ? ad.lr_dc : ad.lr_rest, P[u].y, G.y, M[u].y, V[u].y);
        adam1(ad, c + 2 < 3 ? ad.lr_dc : ad.lr_rest, P[u].z, G.z, M[u].z, V[u].z);
        adam1(ad, ad.lr_rest, P[u].w, G.w, M[u].w, V[u].w);
        st4_nt(sh + e, P[u]); st4_nt(exp_avg + ad.e_opac + e, M[u]); st4_nt(exp_avg_sq + ad.e_opac + e, V[u]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Data-parallel step, fused tail (round 6): SH Adam from the gathered colour blocks + geometry Adam from the
// all-reduced gradients + the next view's colours and K1 in ONE launch over the whole model
// ---------------------------------------------------------------------------------------------
// = k_adam_sh_gathered (per row chunk) followed by k_adam_geom_project_next, for a workgroup's 256 rows: the updated
// SH rows stay in the LDS image the Adam stream has just produced, so the next camera's colour is evaluated there
// instead of from rows re-read from HBM, the gradients of the 11 geometry parameters are read once, and four to five
// launches become one.  Same adam1 / sh_color / project_fwd_core on the same values: bit-identical to the unfused
// sequence.  It can only start when the geometry all-reduce has landed -- which on a node is exactly what the chunked
// SH Adam hides under -- so the host selects it by world size (parallel.GradSync.fused_tail).
struct ChunkTable {
  int n;
  int begin[9];              // row chunk c = [begin[c], begin[c + 1]); multiples of 256 except the last end
  const float* blk[8];       // all-gathered colour blocks of the chunk: [world][3 rows + 4]
};

template <int DEG>
__global__ __launch_bounds__(256) void k_adam_sh_geom_next(
    CamK cam, int world, int N, int sh_stride, float* __restrict__ means, float* __restrict__ log_scales,
    float* __restrict__ quats, float* __restrict__ opac_logit, float* __restrict__ sh, const float* __restrict__ grads,
    ChunkTable ct, AdamK ad_in, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
    float* __restrict__ splats, int32_t* __restrict__ radii, int32_t* __restrict__ group_base,
    int32_t* __restrict__ tile_count, int32_t* __restrict__ rank, int32_t* __restrict__ status,
    long long capacity, int32_t* __restrict__ sticky, int32_t* __restrict__ tag, int32_t tag_value) {
  if (ad_in.guard && ad_in.guard[1]) return;   // a rank's frame overflowed: no update, the tag keeps its old value
  const AdamK ad = adam_resolve(ad_in);
  constexpr int K = (DEG + 1) * (DEG + 1);
  const int ROW = 3 * sh_stride, RS = ROW + 4, F4 = ROW / 4;
  extern __shared__ float4 lds4[];
  float* lds = reinterpret_cast<float*>(lds4);
  const int tid = threadIdx.x;
  const int g0 = blockIdx.x * 256;
  const int g = g0 + tid;
  const int nrows = min(256, N - g0);
  const size_t blk = (size_t)g0 * ROW;
  int row0 = 0, rows_c = N;
  const float* v_color_all = ct.blk[0];
#pragma unroll
  for (int c = 0; c < 8; c++)
    if (c < ct.n && g0 >= ct.begin[c] && g0 < ct.begin[c + 1]) { row0 = ct.begin[c]; rows_c = ct.begin[c + 1] - ct.begin[c]; v_color_all = ct.blk[c]; }
  float m[3] = {0.f, 0.f, 0.f}, ls[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f};
  float ol = 0.f;
  // ---- the SH gradient sum_r Y(dir_r) (x) v_color_r of this row, in rank order, into the LDS image (k_adam_sh_gathered) ----
  if (g < N) {
    m[0] = means[3 * g]; m[1] = means[3 * g + 1]; m[2] = means[3 * g + 2];   // the means the forward pass used
    float acc[3 * K];
#pragma unroll
    for (int i = 0; i < 3 * K; i++) acc[i] = 0.f;
    const size_t nb = (size_t)rows_c;
    const size_t blk_r = 3 * nb + 4;
    for (int r = 0; r < world; r++) {
      const float* vc = v_color_all + r * blk_r + 3 * (size_t)(g - row0);
      const float* cp = v_color_all + r * blk_r + 3 * nb;
      const float v0 = vc[0], v1 = vc[1], v2 = vc[2];
      const float dx = m[0] - cp[0], dy = m[1] - cp[1], dz = m[2] - cp[2];
      const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
      float Y[16];
      sh_basis<DEG>(dx * inv, dy * inv, dz * inv, Y);
#pragma unroll
      for (int k = 0; k < K; k++) {
        acc[3 * k] = fmaf(Y[k], v0, acc[3 * k]);
        acc[3 * k + 1] = fmaf(Y[k], v1, acc[3 * k + 1]);
        acc[3 * k + 2] = fmaf(Y[k], v2, acc[3 * k + 2]);
      }
    }
    float* row = lds + tid * RS;
#pragma unroll
    for (int i = 0; i < 3 * K / 4; i++) st4(row + 4 * i, make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]));
    if constexpr ((3 * K) % 4 != 0) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < (3 * K) % 4; i++) t[i] = acc[(3 * K / 4) * 4 + i];
      st4(row + (3 * K / 4) * 4, make_float4(t[0], t[1], t[2], t[3]));
    }
    for (int i = (3 * K + 3) / 4; i < F4; i++) st4(row + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
  }
  __syncthreads();
  // ---- Adam over the block's SH rows; the updated coefficients replace the consumed gradient in the image ----
  const int nf = nrows * F4;
  for (int f0 = tid; f0 < nf; f0 += 256 * 2) {
    float4 P[2], M[2], V[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int f = f0 + 256 * u;
      if (f < nf) {
        const size_t e = blk + 4 * (size_t)f;
        P[u] = ld4_nt(sh + e); M[u] = ld4_nt(exp_avg + ad.e_opac + e); V[u] = ld4_nt(exp_avg_sq + ad.e_opac + e);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int f = f0 + 256 * u;
      if (f < nf) {
        const int row = f / F4, c4 = f - row * F4;
        const float4 G = ld4(lds + row * RS + 4 * c4);
        const size_t e = blk + 4 * (size_t)f;
        const int c = 4 * c4;
        adam1(ad, c < 3 ? ad.lr_dc : ad.lr_rest, P[u].x, G.x, M[u].x, V[u].x);
        adam1(ad, c + 1 < 3 ? ad.lr_dc : ad.lr_rest, P[u].y, G.y, M[u].y, V[u].y);
        adam1(ad, c + 2 < 3 ? ad.lr_dc : ad.lr_rest, P[u].z, G.z, M[u].z, V[u].z);
        adam1(ad, ad.lr_rest, P[u].w, G.w, M[u].w, V[u].w);
        st4_nt(sh + e, P[u]); st4_nt(exp_avg + ad.e_opac + e, M[u]); st4_nt(exp_avg_sq + ad.e_opac + e, V[u]);
        st4(lds + row * RS + 4 * c4, P[u]);
      }
    }
  }
  __syncthreads();
  // ---- Adam on the 11 geometry parameters (k_adam_geom_project_next), then the next camera's colour from the LDS row ----
  if (g < N) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const long long e = 3ll * g + j;
      float P = m[j], M = exp_avg[e], V = exp_avg_sq[e];
      adam1(ad, ad.lr_means, P, grads[e], M, V);
      means[3 * g + j] = P; exp_avg[e] = M; exp_avg_sq[e] = V;
      m[j] = P;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const long long e = ad.e_means + 3ll * g + j;
      float P = log_scales[3 * g + j], M = exp_avg[e], V = exp_avg_sq[e];
      adam1(ad, ad.lr_scales, P, grads[e], M, V);
      log_scales[3 * g + j] = P; exp_avg[e] = M; exp_avg_sq[e] = V;
      ls[j] = P;
    }
    {
      const long long e = ad.e_scales + 4ll * g;
      float4 Q = ld4(quats + 4 * (size_t)g), G = ld4_nt(grads + e), M = ld4_nt(exp_avg + e), V = ld4_nt(exp_avg_sq + e);
      adam1(ad, ad.lr_quats, Q.x, G.x, M.x, V.x); adam1(ad, ad.lr_quats, Q.y, G.y, M.y, V.y);
      adam1(ad, ad.lr_quats, Q.z, G.z, M.z, V.z); adam1(ad, ad.lr_quats, Q.w, G.w, M.w, V.w);
      st4(quats + 4 * (size_t)g, Q); st4_nt(exp_avg + e, M); st4_nt(exp_avg_sq + e, V);
      q[0] = Q.x; q[1] = Q.y; q[2] = Q.z; q[3] = Q.w;
    }
    {
      const long long e = ad.e_quats + g;
      float P = opac_logit[g], M = exp_avg[e], V = exp_avg_sq[e];
      adam1(ad, ad.lr_opac, P, grads[e], M, V);
      opac_logit[g] = P; exp_avg[e] = M; exp_avg_sq[e] = V;
      ol = P;
    }
    sh_color<DEG>(m, cam.campos, lds + tid * RS, rgb);
  }
  __syncthreads();     // the group scan of K1 takes over the SH image
  project_fwd_core<true>(cam, N, g, m, ls, q, ol, rgb, splats, radii, group_base, tile_count, rank, status,
                         capacity, sticky, reinterpret_cast<GroupScan*>(lds4));
  if (g == 0) *tag = tag_value;
}

// ---------------------------------------------------------------------------------------------
// stand-alone SH evaluation (gsplat `spherical_harmonics` op: no +0.5, no clamp; coefficients only
// receive gradient, view directions are treated as constants)
// ---------------------------------------------------------------------------------------------
template <int DEG, bool BWD>
__global__ __launch_bounds__(256) void k_sh_op(int N, int sh_stride, const float* __restrict__ dirs,
                                               const float* __restrict__ coeffs_or_vout,
                                               float* __restrict__ out) {
  constexpr int K = (DEG + 1) * (DEG + 1);
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  float dx = dirs[3 * g], dy = dirs[3 * g + 1], dz = dirs[3 * g + 2];
  const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  float Y[16];
  sh_basis<DEG>(dx * inv, dy * inv, dz * inv, Y);
  if constexpr (!BWD) {
    const float* c = coeffs_or_vout + (size_t)g * sh_stride * 3;
    float r = 0.f, gg = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) { r += Y[k] * c[3 * k]; gg += Y[k] * c[3 * k + 1]; b += Y[k] * c[3 * k + 2]; }
    out[3 * g] = r; out[3 * g + 1] = gg; out[3 * g + 2] = b;
  } else {
    const float v0 = coeffs_or_vout[3 * g], v1 = coeffs_or_vout[3 * g + 1], v2 = coeffs_or_vout[3 * g + 2];
    float* o = out + (size_t)g * sh_stride * 3;
#pragma unroll
    for (int k = 0; k < K; k++) { o[3 * k] = Y[k] * v0; o[3 * k + 1] = Y[k] * v1; o[3 * k + 2] = Y[k] * v2; }
    for (int k = K; k < sh_stride; k++) { o[3 * k] = 0.f; o[3 * k + 1] = 0.f; o[3 * k + 2] = 0.f; }
  }
}

// the LDS-staged K8 variants exist for rows of 4 or 16 stored bases and any active degree they hold
inline bool lds_k8_ok(int sh_deg, int sh_stride) {
  return (sh_stride == 16 && sh_deg >= 0 && sh_deg <= 3) || (sh_stride == 4 && sh_deg >= 0 && sh_deg <= 1);
}
#define DISPATCH_DEG_KS(M, deg, ks)                                                              \
  do {                                                                                           \
    if ((ks) == 16) {                                                                            \
      switch (deg) { case 0: M(0, 16); break; case 1: M(1, 16); break; case 2: M(2, 16); break;  \
                     default: M(3, 16); break; }                                                 \
    } else {                                                                                     \
      if ((deg) == 0) M(0, 4); else M(1, 4);                                                     \
    }                                                                                            \
  } while (0)

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int tgs_project_fwd(const TgsCamera* cam, int N, const float* means,
                               const float* log_scales, const float* quats,
                               const float* opac_logit, const float* sh, int sh_stride, int sh_deg,
                               const float* colors_in, float* splats, int32_t* radii,
                               void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  TGS_CHECK_ARG(N >= 0, "N < 0");
  TGS_CHECK_ARG(cam->W <= 4080 && cam->H <= 4080, "image side > 4080 px (255 tiles)");
  if (N == 0) return TGS_OK;
  TGS_CHECK_ARG(means && log_scales && quats && opac_logit && splats, "null pointer");
  if (!sh) sh_deg = -1;
  TGS_CHECK_ARG(sh_deg <= 3, "sh_deg > 3");
  TGS_CHECK_ARG(sh_deg < 0 || sh_stride >= (sh_deg + 1) * (sh_deg + 1), "sh_stride too small");
  const CamK k = make_camk(cam);
  const dim3 grid((N + 255) / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(D)                                                                              \
  hipLaunchKernelGGL((k_project_fwd<D, false>), grid, block, 0, s, k, N, means, log_scales, quats, \
                     opac_logit, sh, sh_stride, colors_in, splats, radii, (int32_t*)nullptr,       \
                     (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, 0ll, (int32_t*)nullptr)
  switch (sh_deg) {
    case 0: LAUNCH(0); break;
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    default: LAUNCH(-1); break;
  }
#undef LAUNCH
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_reduce_partials(int N, const float* splats, const int32_t* group_base,
                                   const TgsCamera* cam, const float* partials, float* v_splats,
                                   void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  if (N <= 0) return TGS_OK;
  TGS_CHECK_ARG(splats && group_base && partials && v_splats, "null pointer");
  const CamK k = make_camk(cam);
  hipLaunchKernelGGL(k_reduce_partials, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     k, N, splats, group_base, partials, v_splats);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_project_bwd(const TgsCamera* cam, int N, const float* means,
                               const float* log_scales, const float* quats,
                               const float* opac_logit, const float* sh, int sh_stride, int sh_deg,
                               const float* splats, const int32_t* group_base,
                               const float* partials, const float* v_splats, float* v_means,
                               float* v_log_scales, float* v_quats, float* v_opac_logit,
                               float* v_sh, float* v_xy, const int32_t* skip_if_overflow,
                               void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  if (N <= 0) return TGS_OK;
  TGS_CHECK_ARG(means && log_scales && quats && opac_logit && splats, "null pointer");
  TGS_CHECK_ARG((partials && group_base) || v_splats, "need partials+group_base or v_splats");
  TGS_CHECK_ARG(v_means && v_log_scales && v_quats && v_opac_logit, "null output pointer");
  if (!sh) sh_deg = -1;
  TGS_CHECK_ARG(sh_deg <= 3, "sh_deg > 3");
  TGS_CHECK_ARG(sh_deg < 0 || sh_stride >= (sh_deg + 1) * (sh_deg + 1), "sh_stride too small");
  const CamK k = make_camk(cam);
  const dim3 grid((N + 255) / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
  // fast path: SH rows staged through LDS (storage of 4 or 16 bases, any active degree it holds)
  if (partials && sh && v_sh && sh_deg >= 0 && lds_k8_ok(sh_deg, sh_stride)) {
    AdamK none{};
    none.guard = skip_if_overflow;
    const size_t lds_bytes = 256 * (size_t)(3 * sh_stride + 4) * sizeof(float);
#define LAUNCH_LDS(D, KS)                                                                        \
  hipLaunchKernelGGL((k_project_bwd_lds<D, KS, false>), grid, block, lds_bytes, s, k, N,         \
                     const_cast<float*>(means), const_cast<float*>(log_scales),                  \
                     const_cast<float*>(quats), const_cast<float*>(opac_logit),                  \
                     const_cast<float*>(sh), splats, group_base, partials, v_means, v_log_scales,\
                     v_quats, v_opac_logit, v_sh, v_xy, none, (float*)nullptr, (float*)nullptr, NextView{})
    DISPATCH_DEG_KS(LAUNCH_LDS, sh_deg, sh_stride);
#undef LAUNCH_LDS
    TGS_CHECK_LAUNCH();
    return TGS_OK;
  }
#define LAUNCH(D)                                                                                \
  hipLaunchKernelGGL(k_project_bwd<D>, grid, block, 0, s, k, N, means, log_scales, quats,        \
                     opac_logit, sh, sh_stride, splats, group_base, partials, v_splats, v_means, \
                     v_log_scales, v_quats, v_opac_logit, v_sh, (float*)nullptr, v_xy, skip_if_overflow)
  switch (sh_deg) {
    case 0: LAUNCH(0); break;
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    default: LAUNCH(-1); break;
  }
#undef LAUNCH
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

static int project_bwd_adam_impl(const TgsCamera* cam, int N, int sh_stride, int sh_deg,
                                 float* params, float* exp_avg, float* exp_avg_sq,
                                 const TgsAdamSpec* spec, const float* splats,
                                 const int32_t* group_base, const float* partials, float* v_xy,
                                 const int32_t* skip_if_overflow, const NextView& nx, void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  if (N <= 0) return TGS_OK;
  TGS_CHECK_ARG(params && exp_avg && exp_avg_sq && spec && splats && group_base && partials, "null pointer");
  TGS_CHECK_ARG(sh_deg >= 0 && lds_k8_ok(sh_deg, sh_stride),
                "fused K8+Adam needs an SH tensor storing 4 or 16 bases per Gaussian (degree 1 or 3)");
  const CamK k = make_camk(cam);
  AdamK a = make_adamk(N, sh_stride, spec, 1.0f);
  a.guard = skip_if_overflow;
  float* means = params;
  float* log_scales = params + a.e_means;
  float* quats = params + a.e_scales;
  float* opac = params + a.e_quats;
  float* sh = params + a.e_opac;
  const dim3 grid((N + 255) / 256), block(256);
  const size_t lds_bytes = 256 * (size_t)(3 * sh_stride + 4) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_F(D, KS)                                                                          \
  hipLaunchKernelGGL((k_project_bwd_lds<D, KS, true>), grid, block, lds_bytes, s, k, N, means,   \
                     log_scales, quats, opac, sh, splats, group_base, partials, (float*)nullptr, \
                     (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, v_xy, a,\
                     exp_avg, exp_avg_sq, nx)
  DISPATCH_DEG_KS(LAUNCH_F, sh_deg, sh_stride);
#undef LAUNCH_F
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_project_bwd_adam(const TgsCamera* cam, int N, int sh_stride, int sh_deg,
                                    float* params, float* exp_avg, float* exp_avg_sq,
                                    const TgsAdamSpec* spec, const float* splats,
                                    const int32_t* group_base, const float* partials, float* v_xy,
                                    const int32_t* skip_if_overflow, void* stream) {
  return project_bwd_adam_impl(cam, N, sh_stride, sh_deg, params, exp_avg, exp_avg_sq, spec, splats,
                               group_base, partials, v_xy, skip_if_overflow, NextView{}, stream);
}

extern "C" int tgs_project_bwd_adam_next(const TgsCamera* cam, int N, int sh_stride, int sh_deg,
                                         float* params, float* exp_avg, float* exp_avg_sq,
                                         const TgsAdamSpec* spec, const float* splats,
                                         const int32_t* group_base, const float* partials,
                                         float* v_xy, const int32_t* skip_if_overflow,
                                         const TgsCamera* next_cam, float* colors_next,
                                         int32_t* color_tag, int32_t tag_value, void* stream) {
  TGS_CHECK_ARG(camera_ok(next_cam), "bad next camera");
  TGS_CHECK_ARG(colors_next && color_tag, "null colour prefetch buffer");
  const CamK kn = make_camk(next_cam);
  NextView nx;
  nx.campos[0] = kn.campos[0]; nx.campos[1] = kn.campos[1]; nx.campos[2] = kn.campos[2];
  nx.colors = colors_next; nx.tag = color_tag; nx.tag_value = tag_value;
  return project_bwd_adam_impl(cam, N, sh_stride, sh_deg, params, exp_avg, exp_avg_sq, spec, splats,
                               group_base, partials, v_xy, skip_if_overflow, nx, stream);
}

// Fused K8 + Adam + the next view's colours AND its K1 (front prefetch).  The next frame's counters are cleared
// here, before the kernel that counts into them; tgs_project_bin_sort_front finishes that frame.
extern "C" int tgs_project_bwd_adam_next_front(const TgsCamera* cam, int N, int sh_stride, int sh_deg,
                                               float* params, float* exp_avg, float* exp_avg_sq,
                                               const TgsAdamSpec* spec, const float* splats,
                                               const int32_t* group_base, const float* partials,
                                               float* v_xy, const int32_t* skip_if_overflow,
                                               const TgsCamera* next_cam, float* colors_next,
                                               int32_t* tag_word, int32_t tag_value,
                                               float* splats_next, int32_t* radii_next,
                                               int32_t* group_base_next, int32_t* tile_cursor_next,
                                               int64_t capacity_next, void* scratch_next,
                                               int32_t* status_next, int32_t* sticky_overflow,
                                               int counters_cleared, void* stream) {
  TGS_CHECK_ARG(camera_ok(next_cam), "bad next camera");
  TGS_CHECK_ARG(next_cam->W <= 4080 && next_cam->H <= 4080, "image side > 4080 px (255 tiles)");
  TGS_CHECK_ARG(colors_next && tag_word, "null colour prefetch buffer");
  TGS_CHECK_ARG(splats_next && group_base_next && tile_cursor_next && scratch_next && status_next, "null front buffer");
  TGS_CHECK_ARG(capacity_next >= 0 && capacity_next < (1ll << 31), "bad capacity");
  if (N <= 0) return TGS_OK;
  const CamK kn = make_camk(next_cam);
  const int T = kn.TW * kn.TH;
  const BinScratch sc = carve_scratch(scratch_next, capacity_next);
  if (!counters_cleared) {
    hipLaunchKernelGGL(k_clear_counters, dim3((max(TGS_XCC * T, 2) + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       tile_cursor_next, T, status_next, sticky_overflow);
    TGS_CHECK_LAUNCH();
  }
  NextView nx;
  nx.campos[0] = kn.campos[0]; nx.campos[1] = kn.campos[1]; nx.campos[2] = kn.campos[2];
  nx.colors = colors_next; nx.tag = tag_word; nx.tag_value = tag_value;
  nx.fr.cam = kn; nx.fr.splats = splats_next; nx.fr.radii = radii_next; nx.fr.group_base = group_base_next;
  nx.fr.tile_count = tile_cursor_next; nx.fr.rank = sc.rank; nx.fr.status = status_next;
  nx.fr.capacity = (long long)capacity_next; nx.fr.sticky = sticky_overflow;
  return project_bwd_adam_impl(cam, N, sh_stride, sh_deg, params, exp_avg, exp_avg_sq, spec, splats,
                               group_base, partials, v_xy, skip_if_overflow, nx, stream);
}

// Data-parallel counterpart: geometry Adam from the all-reduced gradients + the next view's K1 (see the kernel).
extern "C" int tgs_adam_geom_project_next(const TgsCamera* next_cam, int N, int sh_stride, int sh_deg,
                                          float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                          const TgsAdamSpec* spec, float grad_scale,
                                          const int32_t* skip_if_overflow, int32_t* tag_word, int32_t tag_value,
                                          float* splats_next, int32_t* radii_next, int32_t* group_base_next,
                                          int32_t* tile_cursor_next, int64_t capacity_next, void* scratch_next,
                                          int32_t* status_next, int32_t* sticky_overflow, int counters_cleared,
                                          void* stream) {
  TGS_CHECK_ARG(camera_ok(next_cam), "bad next camera");
  TGS_CHECK_ARG(next_cam->W <= 4080 && next_cam->H <= 4080, "image side > 4080 px (255 tiles)");
  TGS_CHECK_ARG(N >= 0 && capacity_next >= 0 && capacity_next < (1ll << 31), "bad size");
  if (N == 0) return TGS_OK;
  TGS_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && spec && tag_word, "null pointer");
  TGS_CHECK_ARG(splats_next && group_base_next && tile_cursor_next && scratch_next && status_next, "null front buffer");
  TGS_CHECK_ARG(sh_deg >= 0 && sh_deg <= 3 && sh_stride >= (sh_deg + 1) * (sh_deg + 1), "bad SH degree / stride");
  const CamK kn = make_camk(next_cam);
  const int T = kn.TW * kn.TH;
  hipStream_t s = (hipStream_t)stream;
  const BinScratch sc = carve_scratch(scratch_next, capacity_next);
  if (!counters_cleared) {
    hipLaunchKernelGGL(k_clear_counters, dim3((max(TGS_XCC * T, 2) + 255) / 256), dim3(256), 0, s, tile_cursor_next, T,
                       status_next, sticky_overflow);
    TGS_CHECK_LAUNCH();
  }
  AdamK a = make_adamk(N, sh_stride, spec, grad_scale);
  a.guard = skip_if_overflow;
  float* means = params;
  float* log_scales = params + a.e_means;
  float* quats = params + a.e_scales;
  float* opac = params + a.e_quats;
  const float* sh = params + a.e_opac;
  const dim3 grid((N + 255) / 256), block(256);
#define LAUNCH_G(D)                                                                                          \
  hipLaunchKernelGGL((k_adam_geom_project_next<D>), grid, block, 0, s, kn, N, sh_stride, means, log_scales,    \
                     quats, opac, sh, grads, a, exp_avg, exp_avg_sq, splats_next, radii_next, group_base_next, \
                     tile_cursor_next, sc.rank, status_next, (long long)capacity_next, sticky_overflow, tag_word, tag_value)
  switch (sh_deg) {
    case 0: LAUNCH_G(0); break;
    case 1: LAUNCH_G(1); break;
    case 2: LAUNCH_G(2); break;
    default: LAUNCH_G(3); break;
  }
#undef LAUNCH_G
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

// Can tgs_project_bin_sort_front (N Gaussians) clear the counters of a W x H frame on the side?
extern "C" int tgs_front_can_clear_next(int N, int W, int H) {
  // the scan launch brings its own workgroups for the clearing job (binning.hip: ScanFront.n_clear): any N > 0 will do
  (void)W; (void)H;
  return N > 0;
}

// The rest of a frame whose K1 tgs_project_bwd_adam_next_front (tag_word == tag_expect) has run: scan / fill / sort.
// If the tag does not match (the fused kernel was voided by its overflow guard, which also left the sticky word raised)
// the scan launch voids the frame: empty lists, status[1] = 1 (binning.hip, ScanFront) -- no K1 is re-run for a frame
// nobody will look at.  The caller replays through tgs_project_bin_sort once it has dealt with the overflow.
extern "C" int tgs_project_bin_sort_front(const TgsCamera* cam, int N, const float* means,
                                          const float* log_scales, const float* quats,
                                          const float* opac_logit, const float* sh, int sh_stride,
                                          int sh_deg, float* splats, int32_t* radii, int32_t* group_base,
                                          int32_t* tile_start, int64_t tile_start_len, int32_t* tile_cursor, int32_t* sorted_gid,
                                          int32_t* tile_order, int64_t capacity, void* scratch,
                                          int32_t* status, int32_t* sticky_overflow, int32_t max_list_hint,
                                          const int32_t* tag_word, int32_t tag_expect,
                                          const TgsCamera* next_cam, int32_t* next_tile_cursor,
                                          int32_t* next_status, void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  TGS_CHECK_ARG(N > 0 && capacity >= 0 && capacity < (1ll << 31), "bad size");
  BinFront front;
  front.tag_word = tag_word; front.tag_expect = tag_expect;
  front.next_tile_cursor = next_tile_cursor; front.next_status = next_status; front.next_T = 0;
  if (next_tile_cursor) {
    TGS_CHECK_ARG(camera_ok(next_cam) && next_status, "next frame's counters without its camera / status word");
    const CamK kn = make_camk(next_cam);
    front.next_T = kn.TW * kn.TH;
  }
  TGS_CHECK_ARG(group_base && tile_start && tile_cursor && sorted_gid && scratch && status && tag_word, "null pointer");
  TGS_CHECK_ARG(splats, "null pointer");
  (void)means; (void)log_scales; (void)quats; (void)opac_logit; (void)sh; (void)sh_stride; (void)sh_deg; (void)radii;
  const CamK k = make_camk(cam);
  return tgs_bin_finish(k, N, splats, group_base, tile_start, tile_start_len, tile_cursor, sorted_gid, tile_order,
                        capacity, scratch, status, sticky_overflow, max_list_hint, &front, (hipStream_t)stream);
}

// K1 + K3a fused, then scan / fill / sort: the whole front half of a frame in one call.
static int project_bin_sort_impl(const TgsCamera* cam, int N, const float* means,
                                 const float* log_scales, const float* quats,
                                 const float* opac_logit, const float* sh, int sh_stride,
                                 int sh_deg, float* splats, int32_t* radii, int32_t* group_base,
                                 int32_t* tile_start, int64_t tile_start_len, int32_t* tile_cursor, int32_t* sorted_gid,
                                 int32_t* tile_order, int64_t capacity, void* scratch,
                                 int32_t* status, int32_t* sticky_overflow, int32_t max_list_hint, const float* colors_in,
                                 const int32_t* color_tag, int32_t tag_expect, void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  TGS_CHECK_ARG(N >= 0 && capacity >= 0 && capacity < (1ll << 31), "bad size");
  TGS_CHECK_ARG(cam->W <= 4080 && cam->H <= 4080, "image side > 4080 px (255 tiles)");
  TGS_CHECK_ARG(group_base && tile_start && tile_cursor && sorted_gid && scratch && status, "null pointer");
  TGS_CHECK_ARG(N == 0 || (means && log_scales && quats && opac_logit && splats), "null pointer");
  if (!sh) sh_deg = -1;
  TGS_CHECK_ARG(sh_deg <= 3, "sh_deg > 3");
  TGS_CHECK_ARG(sh_deg < 0 || sh_stride >= (sh_deg + 1) * (sh_deg + 1), "sh_stride too small");
  const CamK k = make_camk(cam);
  const int T = k.TW * k.TH;
  TGS_CHECK_ARG(tile_start_len >= (int64_t)T + 1 + TGS_TILE_START_SCRATCH, "tile_start buffer shorter than tgs_tile_start_len(W, H)");
  hipStream_t s = (hipStream_t)stream;
  const BinScratch sc = carve_scratch(scratch, capacity);
  hipLaunchKernelGGL(k_clear_counters, dim3((max(TGS_XCC * T, 2) + 255) / 256), dim3(256), 0, s, tile_cursor, T, status,
                     sticky_overflow);
  TGS_CHECK_LAUNCH();
  if (N > 0) {
    const dim3 grid((N + 255) / 256), block(256);
#define LAUNCH(D)                                                                                  \
  hipLaunchKernelGGL((k_project_fwd<D, true>), grid, block, 0, s, k, N, means, log_scales, quats,  \
                     opac_logit, sh, sh_stride, (const float*)nullptr, splats, radii, group_base,   \
                     tile_cursor, sc.rank, status, (long long)capacity, sticky_overflow)
#define LAUNCH_PRE(D)                                                                              \
  hipLaunchKernelGGL((k_project_fwd_colors<D>), grid, block, 0, s, k, N, means, log_scales, quats, \
                     opac_logit, sh, sh_stride, colors_in, splats, radii, group_base, tile_cursor,  \
                     sc.rank, status, (long long)capacity, sticky_overflow, color_tag, tag_expect)
    if (colors_in && color_tag && sh_deg >= 0) {
      switch (sh_deg) {
        case 0: LAUNCH_PRE(0); break;
        case 1: LAUNCH_PRE(1); break;
        case 2: LAUNCH_PRE(2); break;
        default: LAUNCH_PRE(3); break;
      }
    } else {
      switch (sh_deg) {
        case 0: LAUNCH(0); break;
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(-1); break;
      }
    }
#undef LAUNCH
#undef LAUNCH_PRE
    TGS_CHECK_LAUNCH();
  }
  return tgs_bin_finish(k, N, splats, group_base, tile_start, tile_start_len, tile_cursor, sorted_gid, tile_order,
                        capacity, scratch, status, sticky_overflow, max_list_hint, nullptr, s);
}

extern "C" int tgs_project_bin_sort(const TgsCamera* cam, int N, const float* means,
                                    const float* log_scales, const float* quats,
                                    const float* opac_logit, const float* sh, int sh_stride,
                                    int sh_deg, float* splats, int32_t* radii, int32_t* group_base,
                                    int32_t* tile_start, int64_t tile_start_len, int32_t* tile_cursor, int32_t* sorted_gid,
                                    int32_t* tile_order, int64_t capacity, void* scratch,
                                    int32_t* status, int32_t* sticky_overflow, int32_t max_list_hint, void* stream) {
  return project_bin_sort_impl(cam, N, means, log_scales, quats, opac_logit, sh, sh_stride, sh_deg, splats,
                               radii, group_base, tile_start, tile_start_len, tile_cursor, sorted_gid, tile_order, capacity,
                               scratch, status, sticky_overflow, max_list_hint, nullptr, nullptr, 0, stream);
}

extern "C" int tgs_project_bin_sort_colors(const TgsCamera* cam, int N, const float* means,
                                           const float* log_scales, const float* quats,
                                           const float* opac_logit, const float* sh, int sh_stride,
                                           int sh_deg, float* splats, int32_t* radii,
                                           int32_t* group_base, int32_t* tile_start, int64_t tile_start_len,
                                           int32_t* tile_cursor, int32_t* sorted_gid,
                                           int32_t* tile_order, int64_t capacity, void* scratch,
                                           int32_t* status, int32_t* sticky_overflow, int32_t max_list_hint,
                                           const float* colors_in, const int32_t* color_tag,
                                           int32_t tag_expect, void* stream) {
  TGS_CHECK_ARG(sh && colors_in && color_tag, "null pointer (the SH rows stay the fallback)");
  return project_bin_sort_impl(cam, N, means, log_scales, quats, opac_logit, sh, sh_stride, sh_deg, splats,
                               radii, group_base, tile_start, tile_start_len, tile_cursor, sorted_gid, tile_order, capacity,
                               scratch, status, sticky_overflow, max_list_hint, colors_in, color_tag, tag_expect, stream);
}

extern "C" int tgs_sh_fwd(int N, int sh_deg, int sh_stride, const float* dirs, const float* coeffs,
                          float* colors, void* stream) {
  TGS_CHECK_ARG(N >= 0 && sh_deg >= 0 && sh_deg <= 3, "bad size / degree");
  TGS_CHECK_ARG(sh_stride >= (sh_deg + 1) * (sh_deg + 1), "sh_stride too small");
  if (N == 0) return TGS_OK;
  TGS_CHECK_ARG(dirs && coeffs && colors, "null pointer");
  const dim3 grid((N + 255) / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (sh_deg) {
    case 0: hipLaunchKernelGGL((k_sh_op<0, false>), grid, block, 0, s, N, sh_stride, dirs, coeffs, colors); break;
    case 1: hipLaunchKernelGGL((k_sh_op<1, false>), grid, block, 0, s, N, sh_stride, dirs, coeffs, colors); break;
    case 2: hipLaunchKernelGGL((k_sh_op<2, false>), grid, block, 0, s, N, sh_stride, dirs, coeffs, colors); break;
    default: hipLaunchKernelGGL((k_sh_op<3, false>), grid, block, 0, s, N, sh_stride, dirs, coeffs, colors); break;
  }
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_sh_bwd(int N, int sh_deg, int sh_stride, const float* dirs, const float* v_colors,
                          float* v_coeffs, void* stream) {
  TGS_CHECK_ARG(N >= 0 && sh_deg >= 0 && sh_deg <= 3, "bad size / degree");
  TGS_CHECK_ARG(sh_stride >= (sh_deg + 1) * (sh_deg + 1), "sh_stride too small");
  if (N == 0) return TGS_OK;
  TGS_CHECK_ARG(dirs && v_colors && v_coeffs, "null pointer");
  const dim3 grid((N + 255) / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (sh_deg) {
    case 0: hipLaunchKernelGGL((k_sh_op<0, true>), grid, block, 0, s, N, sh_stride, dirs, v_colors, v_coeffs); break;
    case 1: hipLaunchKernelGGL((k_sh_op<1, true>), grid, block, 0, s, N, sh_stride, dirs, v_colors, v_coeffs); break;
    case 2: hipLaunchKernelGGL((k_sh_op<2, true>), grid, block, 0, s, N, sh_stride, dirs, v_colors, v_coeffs); break;
    default: hipLaunchKernelGGL((k_sh_op<3, true>), grid, block, 0, s, N, sh_stride, dirs, v_colors, v_coeffs); break;
  }
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

// K8 of the data-parallel step: geometry gradients + clamp-gated colour gradient, no SH rows.
extern "C" int tgs_project_bwd_color(const TgsCamera* cam, int N, const float* means,
                                     const float* log_scales, const float* quats,
                                     const float* opac_logit, const float* sh, int sh_stride,
                                     int sh_deg, const float* splats, const int32_t* group_base,
                                     const float* partials, float* v_means, float* v_log_scales,
                                     float* v_quats, float* v_opac_logit, float* v_color,
                                     float* v_xy, const int32_t* skip_if_overflow, void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  if (N <= 0) return TGS_OK;
  TGS_CHECK_ARG(means && log_scales && quats && opac_logit && sh && splats && group_base && partials,
                "null pointer");
  TGS_CHECK_ARG(v_means && v_log_scales && v_quats && v_opac_logit && v_color, "null output pointer");
  TGS_CHECK_ARG(sh_deg >= 0 && sh_deg <= 3, "sh_deg out of range");
  TGS_CHECK_ARG(sh_stride >= (sh_deg + 1) * (sh_deg + 1), "sh_stride too small");
  const CamK k = make_camk(cam);
  const dim3 grid((N + 255) / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (lds_k8_ok(sh_deg, sh_stride)) {
    AdamK none{};
    none.guard = skip_if_overflow;
    const size_t lds_bytes = 256 * (size_t)(3 * sh_stride + 4) * sizeof(float);
#define LAUNCH_C(D, KS)                                                                          \
  hipLaunchKernelGGL((k_project_bwd_lds<D, KS, false, true>), grid, block, lds_bytes, s, k, N,   \
                     const_cast<float*>(means), const_cast<float*>(log_scales),                  \
                     const_cast<float*>(quats), const_cast<float*>(opac_logit),                  \
                     const_cast<float*>(sh), splats, group_base, partials, v_means, v_log_scales,\
                     v_quats, v_opac_logit, v_color, v_xy, none, (float*)nullptr, (float*)nullptr, NextView{})
    DISPATCH_DEG_KS(LAUNCH_C, sh_deg, sh_stride);
#undef LAUNCH_C
    TGS_CHECK_LAUNCH();
    return TGS_OK;
  }
#define LAUNCH(D)                                                                                \
  hipLaunchKernelGGL(k_project_bwd<D>, grid, block, 0, s, k, N, means, log_scales, quats,        \
                     opac_logit, sh, sh_stride, splats, group_base, partials,                    \
                     (const float*)nullptr, v_means, v_log_scales, v_quats, v_opac_logit,        \
                     (float*)nullptr, v_color, v_xy, skip_if_overflow)
  switch (sh_deg) {
    case 0: LAUNCH(0); break;
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    default: LAUNCH(3); break;
  }
#undef LAUNCH
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

// The same for the model rows [row_begin, row_end) only (one chunk of a pipelined exchange): all per-Gaussian
// arrays are the whole model's, v_color_rows is the CHUNK's block [3 (row_end - row_begin) + 4] with its own
// trailer.  A workgroup owns 256 consecutive rows and a binning group is TGS_GROUP rows, so a chunk that starts
// on a multiple of TGS_GROUP is the whole-model launch restricted to some of its workgroups -- same arithmetic.
extern "C" int tgs_project_bwd_color_rows(const TgsCamera* cam, int N, int row_begin, int row_end,
                                          const float* means, const float* log_scales, const float* quats,
                                          const float* opac_logit, const float* sh, int sh_stride,
                                          int sh_deg, const float* splats, const int32_t* group_base,
                                          const float* partials, float* v_means, float* v_log_scales,
                                          float* v_quats, float* v_opac_logit, float* v_color_rows,
                                          float* v_xy, const int32_t* skip_if_overflow, void* stream) {
  TGS_CHECK_ARG(row_begin >= 0 && row_begin <= row_end && row_end <= N, "bad row range");
  TGS_CHECK_ARG(row_begin % TGS_GROUP == 0, "row_begin must be a multiple of TGS_GROUP");
  if (row_end == row_begin) return TGS_OK;
  TGS_CHECK_ARG(means && log_scales && quats && opac_logit && sh && splats && group_base && partials,
                "null pointer");
  TGS_CHECK_ARG(v_means && v_log_scales && v_quats && v_opac_logit && v_color_rows, "null output pointer");
  const size_t r = (size_t)row_begin;
  return tgs_project_bwd_color(cam, row_end - row_begin, means + 3 * r, log_scales + 3 * r, quats + 4 * r,
                               opac_logit + r, sh + r * (size_t)sh_stride * 3, sh_stride, sh_deg,
                               splats + r * TGS_SPLAT_FLOATS, group_base + r / TGS_GROUP, partials,
                               v_means + 3 * r, v_log_scales + 3 * r, v_quats + 4 * r, v_opac_logit + r,
                               v_color_rows, v_xy ? v_xy + 2 * r : nullptr, skip_if_overflow, stream);
}

// Sync-free intersection budget under data parallelism: every rank's colour-gradient block carries
// its frame's overflow flag in the pad slot; after the all-gather each rank derives the SAME verdict.
static __global__ void k_dp_agree_overflow(int world, size_t blk, const float* __restrict__ v_color_all,
                                           int32_t* __restrict__ status_out, int32_t* __restrict__ sticky) {
  int any = 0;
  for (int r = threadIdx.x; r < world; r += TGS_WAVE) any |= v_color_all[r * blk + blk - 1] != 0.f;
  any = __ballot(any) != 0ull;
  // a sticky word that is already raised voids this step too: by an earlier overflow (then every rank's flag is set
  // anyway) or by a peer-exchange wait that timed out in front of this launch (tgs_peer_wait's poison word)
  if (sticky && *sticky) any = 1;
  if (threadIdx.x == 0) {
    status_out[0] = 0;
    status_out[1] = any;
    if (any && sticky) *sticky = 1;
  }
}

extern "C" int tgs_dp_agree_overflow(int world, int N, const float* v_color_all, int32_t* status_out,
                                     int32_t* sticky_overflow, void* stream) {
  TGS_CHECK_ARG(world >= 1 && N >= 0, "bad size");
  TGS_CHECK_ARG(v_color_all && status_out, "null pointer");
  hipLaunchKernelGGL(k_dp_agree_overflow, dim3(1), dim3(TGS_WAVE), 0, (hipStream_t)stream, world,
                     3 * (size_t)N + 4, v_color_all, status_out, sticky_overflow);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_adam_step_sh_gathered_rows(int world, int N, int row_begin, int row_end, int sh_stride,
                                              int sh_deg, float* params, const float* v_color_rows_all,
                                              float* exp_avg, float* exp_avg_sq, const TgsAdamSpec* spec,
                                              float grad_scale, const int32_t* skip_if_overflow, void* stream) {
  TGS_CHECK_ARG(world >= 1 && N >= 0, "bad size");
  TGS_CHECK_ARG(row_begin >= 0 && row_begin <= row_end && row_end <= N, "bad row range");
  if (row_end == row_begin) return TGS_OK;
  TGS_CHECK_ARG(params && v_color_rows_all && exp_avg && exp_avg_sq && spec, "null pointer");
  TGS_CHECK_ARG(sh_deg >= 0 && sh_deg <= 3 && sh_stride >= (sh_deg + 1) * (sh_deg + 1), "bad SH degree / stride");
  TGS_CHECK_ARG((3 * sh_stride) % 4 == 0, "SH row (3*sh_stride floats) must be a multiple of 16 bytes");
  AdamK a = make_adamk(N, sh_stride, spec, grad_scale);   // segment offsets of the WHOLE model
  a.guard = skip_if_overflow;
  const dim3 grid((row_end - row_begin + 255) / 256), block(256);
  const size_t lds_bytes = 256 * (size_t)(3 * sh_stride + 4) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_G(D)                                                                              \
  hipLaunchKernelGGL((k_adam_sh_gathered<D>), grid, block, lds_bytes, s, world, row_begin, row_end, sh_stride, \
                     params, params + a.e_opac, v_color_rows_all, a, exp_avg, exp_avg_sq)
  switch (sh_deg) {
    case 0: LAUNCH_G(0); break;
    case 1: LAUNCH_G(1); break;
    case 2: LAUNCH_G(2); break;
    default: LAUNCH_G(3); break;
  }
#undef LAUNCH_G
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_adam_step_sh_gathered(int world, int N, int sh_stride, int sh_deg, float* params,
                                         const float* v_color_all, float* exp_avg,
                                         float* exp_avg_sq, const TgsAdamSpec* spec,
                                         float grad_scale, const int32_t* skip_if_overflow, void* stream) {
  TGS_CHECK_ARG(world >= 1 && N >= 0, "bad size");
  return tgs_adam_step_sh_gathered_rows(world, N, 0, N, sh_stride, sh_deg, params, v_color_all, exp_avg,
                                        exp_avg_sq, spec, grad_scale, skip_if_overflow, stream);
}

// The fused tail of a data-parallel step (k_adam_sh_geom_next): tgs_adam_step_sh_gathered_rows for every row chunk +
// tgs_adam_geom_project_next, in one launch.  chunk_begin[n_chunks + 1] (host): the chunks' row ranges, consecutive from
// 0 to N, every begin a multiple of TGS_GROUP; chunk_blocks[n_chunks] (host array of DEVICE pointers): the all-gathered
// colour blocks [world][3 rows + 4] of each chunk.
extern "C" int tgs_adam_sh_gathered_geom_project_next(
    const TgsCamera* next_cam, int world, int N, int sh_stride, int sh_deg, float* params, const float* grads,
    int n_chunks, const int32_t* chunk_begin, const float* const* chunk_blocks, float* exp_avg, float* exp_avg_sq,
    const TgsAdamSpec* spec, float grad_scale, const int32_t* skip_if_overflow, int32_t* tag_word, int32_t tag_value,
    float* splats_next, int32_t* radii_next, int32_t* group_base_next, int32_t* tile_cursor_next,
    int64_t capacity_next, void* scratch_next, int32_t* status_next, int32_t* sticky_overflow, int counters_cleared,
    void* stream) {
  TGS_CHECK_ARG(camera_ok(next_cam), "bad next camera");
  TGS_CHECK_ARG(next_cam->W <= 4080 && next_cam->H <= 4080, "image side > 4080 px (255 tiles)");
  TGS_CHECK_ARG(world >= 1 && N >= 0 && capacity_next >= 0 && capacity_next < (1ll << 31), "bad size");
  if (N == 0) return TGS_OK;
  TGS_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && spec && tag_word, "null pointer");
  TGS_CHECK_ARG(splats_next && group_base_next && tile_cursor_next && scratch_next && status_next, "null front buffer");
  TGS_CHECK_ARG(sh_deg >= 0 && sh_deg <= 3 && sh_stride >= (sh_deg + 1) * (sh_deg + 1), "bad SH degree / stride");
  TGS_CHECK_ARG((3 * sh_stride) % 4 == 0, "SH row (3*sh_stride floats) must be a multiple of 16 bytes");
  TGS_CHECK_ARG(256 * (size_t)(3 * sh_stride + 4) * sizeof(float) >= sizeof(GroupScan), "SH storage too small for the fused tail (needs >= 4 bases)");
  TGS_CHECK_ARG(n_chunks >= 1 && n_chunks <= 8 && chunk_begin && chunk_blocks, "1 .. 8 row chunks");
  ChunkTable ct;
  ct.n = n_chunks;
  for (int c = 0; c <= 8; c++) ct.begin[c] = c <= n_chunks ? chunk_begin[c] : N;
  for (int c = 0; c < 8; c++) ct.blk[c] = c < n_chunks ? chunk_blocks[c] : nullptr;
  TGS_CHECK_ARG(ct.begin[0] == 0 && ct.begin[n_chunks] == N, "the chunks must cover rows [0, N)");
  for (int c = 0; c < n_chunks; c++) {
    TGS_CHECK_ARG(ct.begin[c] % TGS_GROUP == 0 && ct.begin[c] < ct.begin[c + 1] && ct.blk[c], "bad row chunk");
  }
  const CamK kn = make_camk(next_cam);
  const int T = kn.TW * kn.TH;
  hipStream_t s = (hipStream_t)stream;
  const BinScratch sc = carve_scratch(scratch_next, capacity_next);
  if (!counters_cleared) {
    hipLaunchKernelGGL(k_clear_counters, dim3((max(TGS_XCC * T, 2) + 255) / 256), dim3(256), 0, s, tile_cursor_next, T,
                       status_next, sticky_overflow);
    TGS_CHECK_LAUNCH();
  }
  AdamK a = make_adamk(N, sh_stride, spec, grad_scale);
  a.guard = skip_if_overflow;
  const dim3 grid((N + 255) / 256), block(256);
  const size_t lds_bytes = 256 * (size_t)(3 * sh_stride + 4) * sizeof(float);
#define LAUNCH_T(D)                                                                                            \
  hipLaunchKernelGGL((k_adam_sh_geom_next<D>), grid, block, lds_bytes, s, kn, world, N, sh_stride, params,       \
                     params + a.e_means, params + a.e_scales, params + a.e_quats, params + a.e_opac, grads, ct, a, \
                     exp_avg, exp_avg_sq, splats_next, radii_next, group_base_next, tile_cursor_next, sc.rank,     \
                     status_next, (long long)capacity_next, sticky_overflow, tag_word, tag_value)
  switch (sh_deg) {
    case 0: LAUNCH_T(0); break;
    case 1: LAUNCH_T(1); break;
    case 2: LAUNCH_T(2); break;
    default: LAUNCH_T(3); break;
  }
#undef LAUNCH_T
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}
