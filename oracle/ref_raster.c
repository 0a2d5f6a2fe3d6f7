/* TEST INFRASTRUCTURE ONLY -- scalar C restatement of SURVEY.md Appendix B (B.1-B.8).
 *
 * PARITY UNPINNED: the reference tree has no rasterizer source (reference .gitmodules:7-9 is
 * an empty submodule; the only call site is scripts/train_bunny_real.sh:52), so this file
 * restates the *published* 3DGS / EWA algorithm with the constants of Appendix B.  It is an
 * independent second restatement (the first is oracle/torch_oracle.py, whose autograd is the
 * gradient ground truth); tests cross-check the two.  It is also the "port" CPU baseline that
 * bench.py times next to the GPU numbers.  Never linked into the product library.
 *
 * Build: see oracle/Makefile (REAL = float -> libref_raster_f32.so, double -> _f64.so).
 */
#include <tgmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

#define BLK 16
#define ALPHA_MIN ((real)(1.0 / 255.0))
#define ALPHA_MAX ((real)0.999)
#define T_STOP ((real)1e-4)
#define BLUR ((real)0.3)

static const double C0 = 0.28209479177387814, C1 = 0.4886025119029199;
static const double C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                             -1.0925484305920792, 0.5462742152960396};
static const double C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                             0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                             -0.5900435899266435};

/* camera block: viewmat[16] row-major, fx, fy, cx, cy, near, pix_center, bg[3]  (25 reals) */
typedef struct { real V[16], fx, fy, cx, cy, near_, pc, bg[3]; } cam_t;

static void sh_eval(int deg, real x, real y, real z, real *Y) { /* B.5 */
    Y[0] = (real)C0;
    if (deg < 1) return;
    Y[1] = (real)(-C1) * y; Y[2] = (real)C1 * z; Y[3] = (real)(-C1) * x;
    if (deg < 2) return;
    real xx = x * x, yy = y * y, zz = z * z;
    Y[4] = (real)C2[0] * x * y; Y[5] = (real)C2[1] * y * z;
    Y[6] = (real)C2[2] * (2 * zz - xx - yy);
    Y[7] = (real)C2[3] * x * z; Y[8] = (real)C2[4] * (xx - yy);
    if (deg < 3) return;
    Y[9] = (real)C3[0] * y * (3 * xx - yy); Y[10] = (real)C3[1] * x * y * z;
    Y[11] = (real)C3[2] * y * (4 * zz - xx - yy);
    Y[12] = (real)C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
    Y[13] = (real)C3[4] * x * (4 * zz - xx - yy);
    Y[14] = (real)C3[5] * z * (xx - yy); Y[15] = (real)C3[6] * x * (xx - 3 * yy);
}

/* dY_k/d(x,y,z) */
static void sh_grad(int deg, real x, real y, real z, real (*dY)[3]) {
    for (int k = 0; k < 16; k++) dY[k][0] = dY[k][1] = dY[k][2] = 0;
    if (deg < 1) return;
    dY[1][1] = (real)(-C1); dY[2][2] = (real)C1; dY[3][0] = (real)(-C1);
    if (deg < 2) return;
    dY[4][0] = (real)C2[0] * y; dY[4][1] = (real)C2[0] * x;
    dY[5][1] = (real)C2[1] * z; dY[5][2] = (real)C2[1] * y;
    dY[6][0] = (real)(-2 * C2[2]) * x; dY[6][1] = (real)(-2 * C2[2]) * y; dY[6][2] = (real)(4 * C2[2]) * z;
    dY[7][0] = (real)C2[3] * z; dY[7][2] = (real)C2[3] * x;
    dY[8][0] = (real)(2 * C2[4]) * x; dY[8][1] = (real)(-2 * C2[4]) * y;
    if (deg < 3) return;
    real xx = x * x, yy = y * y, zz = z * z;
    dY[9][0] = (real)(6 * C3[0]) * x * y; dY[9][1] = (real)(3 * C3[0]) * (xx - yy);
    dY[10][0] = (real)C3[1] * y * z; dY[10][1] = (real)C3[1] * x * z; dY[10][2] = (real)C3[1] * x * y;
    dY[11][0] = (real)(-2 * C3[2]) * x * y; dY[11][1] = (real)C3[2] * (4 * zz - xx - 3 * yy);
    dY[11][2] = (real)(8 * C3[2]) * y * z;
    dY[12][0] = (real)(-6 * C3[3]) * x * z; dY[12][1] = (real)(-6 * C3[3]) * y * z;
    dY[12][2] = (real)C3[3] * (6 * zz - 3 * xx - 3 * yy);
    dY[13][0] = (real)C3[4] * (4 * zz - 3 * xx - yy); dY[13][1] = (real)(-2 * C3[4]) * x * y;
    dY[13][2] = (real)(8 * C3[4]) * x * z;
    dY[14][0] = (real)(2 * C3[5]) * x * z; dY[14][1] = (real)(-2 * C3[5]) * y * z;
    dY[14][2] = (real)C3[5] * (xx - yy);
    dY[15][0] = (real)(3 * C3[6]) * (xx - yy); dY[15][1] = (real)(-6 * C3[6]) * x * y;
}

static void quat_rot(const real *q, real *R, real *qn, real *norm) { /* B.2 */
    real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    real w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z; *norm = n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* B.1-B.5.  Per Gaussian outputs; culled -> radius 0, tiles_hit 0. */
void ref_project_fwd(int N, const real *means, const real *log_scales, const real *quats,
                     const real *opac_logit, const real *sh, int K_stride, int sh_deg,
                     const cam_t *cam, int W, int H, real glob_scale,
                     real *xy, real *depth, int32_t *radius, real *conic, real *rgb, real *opac,
                     int32_t *rect, int32_t *tiles_hit) {
    const real *V = cam->V;
    int TW = (W + BLK - 1) / BLK, TH = (H + BLK - 1) / BLK;
    real campos[3];
    for (int j = 0; j < 3; j++) campos[j] = -(V[0 + j] * V[3] + V[4 + j] * V[7] + V[8 + j] * V[11]);
    real limx = (real)1.3 * (W / (real)2) / cam->fx, limy = (real)1.3 * (H / (real)2) / cam->fy;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        const real *m = means + 3 * i;
        radius[i] = 0; tiles_hit[i] = 0;
        rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0; depth[i] = 0;
        conic[3 * i] = conic[3 * i + 1] = conic[3 * i + 2] = 0;
        opac[i] = 1 / (1 + exp(-opac_logit[i]));
        if (sh) {
            real dx = m[0] - campos[0], dy = m[1] - campos[1], dz = m[2] - campos[2];
            real dn = sqrt(dx * dx + dy * dy + dz * dz);
            real Y[16];
            sh_eval(sh_deg, dx / dn, dy / dn, dz / dn, Y);
            int K = (sh_deg + 1) * (sh_deg + 1);
            for (int ch = 0; ch < 3; ch++) {
                real acc = 0;
                for (int k = 0; k < K; k++) acc += Y[k] * sh[((size_t)i * K_stride + k) * 3 + ch];
                acc += (real)0.5;
                rgb[3 * i + ch] = acc > 0 ? acc : 0;
            }
        }
        real tx = V[0] * m[0] + V[1] * m[1] + V[2] * m[2] + V[3];
        real ty = V[4] * m[0] + V[5] * m[1] + V[6] * m[2] + V[7];
        real tz = V[8] * m[0] + V[9] * m[1] + V[10] * m[2] + V[11];
        depth[i] = tz;
        if (!(tz > cam->near_)) continue;
        real R[9], qn[4], nrm;
        quat_rot(quats + 4 * i, R, qn, &nrm);
        real s[3];
        for (int j = 0; j < 3; j++) s[j] = exp(log_scales[3 * i + j]) * glob_scale;
        real M[9], S[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[3 * r + c] = R[3 * r + c] * s[c];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
            real a = 0; for (int k = 0; k < 3; k++) a += M[3 * r + k] * M[3 * c + k];
            S[3 * r + c] = a;
        }
        real ux = tx / tz, uy = ty / tz;
        real ucx = ux < -limx ? -limx : (ux > limx ? limx : ux);
        real ucy = uy < -limy ? -limy : (uy > limy ? limy : uy);
        real txc = tz * ucx, tyc = tz * ucy;
        real J[6] = {cam->fx / tz, 0, -cam->fx * txc / (tz * tz), 0, cam->fy / tz, -cam->fy * tyc / (tz * tz)};
        real Tm[6];
        for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++)
            Tm[3 * r + c] = J[3 * r] * V[c] + J[3 * r + 1] * V[4 + c] + J[3 * r + 2] * V[8 + c];
        real TS[6];
        for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) {
            real a = 0; for (int k = 0; k < 3; k++) a += Tm[3 * r + k] * S[3 * k + c];
            TS[3 * r + c] = a;
        }
        real c00 = TS[0] * Tm[0] + TS[1] * Tm[1] + TS[2] * Tm[2] + BLUR;
        real c01 = TS[0] * Tm[3] + TS[1] * Tm[4] + TS[2] * Tm[5];
        real c11 = TS[3] * Tm[3] + TS[4] * Tm[4] + TS[5] * Tm[5] + BLUR;
        real det = c00 * c11 - c01 * c01;
        xy[2 * i] = cam->fx * tx / tz + cam->cx;
        xy[2 * i + 1] = cam->fy * ty / tz + cam->cy;
        if (!(det > 0)) continue;
        conic[3 * i] = c11 / det; conic[3 * i + 1] = -c01 / det; conic[3 * i + 2] = c00 / det;
        real mid = (real)0.5 * (c00 + c11);
        real disc = mid * mid - det; if (disc < (real)0.1) disc = (real)0.1;
        real lam1 = mid + sqrt(disc);
        int rad = (int)ceil(3 * sqrt(lam1));
        real u = xy[2 * i], v = xy[2 * i + 1];
        int x0 = clampi((int)((u - rad) / BLK), 0, TW), x1 = clampi((int)((u + rad) / BLK) + 1, 0, TW);
        int y0 = clampi((int)((v - rad) / BLK), 0, TH), y1 = clampi((int)((v + rad) / BLK) + 1, 0, TH);
        int hit = (x1 - x0) * (y1 - y0);
        rect[4 * i] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
        if (hit <= 0) continue;
        radius[i] = rad; tiles_hit[i] = hit;
    }
}

typedef struct { uint32_t dbits; int32_t gid; } pair_t;
static int pair_cmp(const void *a, const void *b) {
    const pair_t *p = a, *q = b;
    if (p->dbits != q->dbits) return p->dbits < q->dbits ? -1 : 1;
    return (p->gid > q->gid) - (p->gid < q->gid);
}

/* B.6 keys + stable sort.  depth32 = fp32 depth (bits are the key).  tile_start has T+1 entries;
 * sorted_gid has sum(tiles_hit) entries. */
void ref_bin_sort(int N, const int32_t *rect, const int32_t *tiles_hit, const float *depth32,
                  int TW, int TH, int32_t *sorted_gid, int64_t *tile_start) {
    int T = TW * TH;
    int64_t *cnt = calloc((size_t)T + 1, sizeof(int64_t));
    for (int i = 0; i < N; i++) {
        if (!tiles_hit[i]) continue;
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; y++)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; x++) cnt[y * TW + x]++;
    }
    tile_start[0] = 0;
    for (int t = 0; t < T; t++) tile_start[t + 1] = tile_start[t] + cnt[t];
    int64_t nI = tile_start[T];
    pair_t *pairs = malloc((size_t)(nI > 0 ? nI : 1) * sizeof(pair_t));
    memset(cnt, 0, ((size_t)T + 1) * sizeof(int64_t));
    for (int i = 0; i < N; i++) {
        if (!tiles_hit[i]) continue;
        uint32_t bits; memcpy(&bits, depth32 + i, 4);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; y++)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; x++) {
                int t = y * TW + x;
                pair_t *p = pairs + tile_start[t] + cnt[t]++;
                p->dbits = bits; p->gid = i;
            }
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (int t = 0; t < T; t++) {
        int64_t s = tile_start[t], n = tile_start[t + 1] - s;
        if (n > 1) qsort(pairs + s, (size_t)n, sizeof(pair_t), pair_cmp);
        for (int64_t k = 0; k < n; k++) sorted_gid[s + k] = pairs[s + k].gid;
    }
    free(pairs); free(cnt);
}

/* B.6 forward blend.  Images row-major: rgb [H,W,3] (incl. bg), depth_acc/final_T [H,W], final_idx int32 [H,W]. */
void ref_blend_fwd_range(const real *xy, const real *conic, const real *opac, const real *rgb, const real *depth,
                   const int32_t *sorted_gid, const int64_t *tile_start, const cam_t *cam, int W, int H,
                   real *o_rgb, real *o_depth, real *o_T, int32_t *o_idx, int t0, int t1) {
    int TW = (W + BLK - 1) / BLK;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = t0; t < t1; t++) {
        int ty = t / TW, tx = t % TW;
        int64_t s = tile_start[t], e = tile_start[t + 1];
        for (int py = ty * BLK; py < (ty + 1) * BLK && py < H; py++)
            for (int px = tx * BLK; px < (tx + 1) * BLK && px < W; px++) {
                real fxp = px + cam->pc, fyp = py + cam->pc;
                real T = 1, C[3] = {0, 0, 0}, D = 0; int last = -1;
                for (int64_t k = s; k < e; k++) {
                    int g = sorted_gid[k];
                    real dx = xy[2 * g] - fxp, dy = xy[2 * g + 1] - fyp;
                    real sig = (real)0.5 * (conic[3 * g] * dx * dx + conic[3 * g + 2] * dy * dy) + conic[3 * g + 1] * dx * dy;
                    if (sig < 0) continue;
                    real al = opac[g] * exp(-sig); if (al > ALPHA_MAX) al = ALPHA_MAX;
                    if (al < ALPHA_MIN) continue;
                    real Tn = T * (1 - al);
                    if (Tn <= T_STOP) break;
                    real w = al * T;
                    C[0] += w * rgb[3 * g]; C[1] += w * rgb[3 * g + 1]; C[2] += w * rgb[3 * g + 2];
                    D += w * depth[g]; T = Tn; last = (int)(k - s);
                }
                size_t p = (size_t)py * W + px;
                for (int c = 0; c < 3; c++) o_rgb[3 * p + c] = C[c] + T * cam->bg[c];
                o_depth[p] = D; o_T[p] = T; o_idx[p] = last;
            }
    }
}

void ref_blend_fwd(const real *xy, const real *conic, const real *opac, const real *rgb, const real *depth,
                   const int32_t *sorted_gid, const int64_t *tile_start, const cam_t *cam, int W, int H,
                   real *o_rgb, real *o_depth, real *o_T, int32_t *o_idx) {
    int TW = (W + BLK - 1) / BLK, TH = (H + BLK - 1) / BLK;
    ref_blend_fwd_range(xy, conic, opac, rgb, depth, sorted_gid, tile_start, cam, W, H, o_rgb, o_depth, o_T, o_idx, 0, TW * TH);
}

/* Per-pixel decision margin of the forward blend (test aid for full-size parity): the smallest
 * relative distance of any threshold test that B.6 actually evaluates for the pixel (alpha vs 1/255,
 * T' vs 1e-4) from its threshold.  A pixel with a tiny margin is decision-ambiguous under fp32
 * rounding; parity is asserted on the others.  Same loop as ref_blend_fwd_range. */
void ref_blend_margin_range(const real *xy, const real *conic, const real *opac,
                   const int32_t *sorted_gid, const int64_t *tile_start, const cam_t *cam, int W, int H,
                   double *o_margin, int t0, int t1) {
    int TW = (W + BLK - 1) / BLK;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = t0; t < t1; t++) {
        int ty = t / TW, tx = t % TW;
        int64_t s = tile_start[t], e = tile_start[t + 1];
        for (int py = ty * BLK; py < (ty + 1) * BLK && py < H; py++)
            for (int px = tx * BLK; px < (tx + 1) * BLK && px < W; px++) {
                real fxp = px + cam->pc, fyp = py + cam->pc;
                real T = 1; double m = 1e30;
                for (int64_t k = s; k < e; k++) {
                    int g = sorted_gid[k];
                    real dx = xy[2 * g] - fxp, dy = xy[2 * g + 1] - fyp;
                    real sig = (real)0.5 * (conic[3 * g] * dx * dx + conic[3 * g + 2] * dy * dy) + conic[3 * g + 1] * dx * dy;
                    if (sig < 0) continue;
                    real araw = opac[g] * exp(-sig), al = araw > ALPHA_MAX ? ALPHA_MAX : araw;
                    double ma = fabs((double)araw - (double)ALPHA_MIN) / (double)ALPHA_MIN;
                    if (ma < m) m = ma;
                    if (al < ALPHA_MIN) continue;
                    real Tn = T * (1 - al);
                    double mt = fabs((double)Tn - (double)T_STOP) / (double)T_STOP;
                    if (mt < m) m = mt;
                    if (Tn <= T_STOP) break;
                    T = Tn;
                }
                o_margin[(size_t)py * W + px] = m;
            }
    }
}

/* Test aid for the product's tight tile rectangle: for every Gaussian, the
 * largest o*exp(-sigma) over the pixel centres of all tiles that lie in the NORMATIVE B.4 rect
 * `rect` but outside `tight` (both x0,y0,x1,y1).  Output preservation requires it to stay below
 * 1/255 (B.6 skips such pixels).  o_count receives the number of dropped (tile, Gaussian) pairs. */
void ref_dropped_pairs_max_alpha(int N, const real *xy, const real *conic, const real *opac,
                   const int32_t *rect, const int32_t *tiles_hit, const int32_t *tight,
                   const cam_t *cam, int W, int H, real *o_max_alpha, int64_t *o_count) {
    int64_t count = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : count)
    for (int i = 0; i < N; i++) {
        real best = 0;
        if (tiles_hit[i])
            for (int ty = rect[4 * i + 1]; ty < rect[4 * i + 3]; ty++)
                for (int tx = rect[4 * i]; tx < rect[4 * i + 2]; tx++) {
                    if (tx >= tight[4 * i] && tx < tight[4 * i + 2] && ty >= tight[4 * i + 1] && ty < tight[4 * i + 3])
                        continue;
                    count++;
                    for (int py = ty * BLK; py < (ty + 1) * BLK && py < H; py++)
                        for (int px = tx * BLK; px < (tx + 1) * BLK && px < W; px++) {
                            real dx = xy[2 * i] - (px + cam->pc), dy = xy[2 * i + 1] - (py + cam->pc);
                            real sig = (real)0.5 * (conic[3 * i] * dx * dx + conic[3 * i + 2] * dy * dy) + conic[3 * i + 1] * dx * dy;
                            if (sig < 0) continue;
                            real al = opac[i] * exp(-sig);
                            if (al > best) best = al;
                        }
                }
        o_max_alpha[i] = best;
    }
    *o_count = count;
}

/* Test aid for gradient parity at full size: classifies GAUSSIANS by the decision margins of the pixels
 * they (nearly) contribute to.  pix_margin [H,W] is the per-pixel map of ref_blend_margin_range (the caller
 * zeroes it in tiles whose fp32 depth order differs from the fp64 order).  For every Gaussian the minimum of
 * pix_margin over the pixel centres of its NORMATIVE B.4 rect where o*exp(-sigma) >= near * 1/255 (near < 1:
 * contributing, or close enough to the threshold to contribute under fp32 rounding) -- 1e30 if there is no
 * such pixel.  A flipped alpha_min / T_stop decision at a pixel changes the transmittance every later
 * Gaussian of that pixel sees, so a Gaussian's gradient is only comparable at 1e-4 if ALL the pixels it
 * reaches are decision-clear; occlusion (T_stop) is ignored, which can only classify too many as unclear. */
void ref_gaussian_min_margin(int N, const real *xy, const real *conic, const real *opac,
                   const int32_t *rect, const int32_t *tiles_hit, const cam_t *cam, int W, int H,
                   const double *pix_margin, double near, double *o_min, int64_t *o_npix) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < N; i++) {
        double best = 1e30; int64_t np_ = 0;
        if (tiles_hit[i])
            for (int py = rect[4 * i + 1] * BLK; py < rect[4 * i + 3] * BLK && py < H; py++)
                for (int px = rect[4 * i] * BLK; px < rect[4 * i + 2] * BLK && px < W; px++) {
                    real dx = xy[2 * i] - (px + cam->pc), dy = xy[2 * i + 1] - (py + cam->pc);
                    real sig = (real)0.5 * (conic[3 * i] * dx * dx + conic[3 * i + 2] * dy * dy) + conic[3 * i + 1] * dx * dy;
                    if (sig < 0) continue;
                    if ((double)(opac[i] * exp(-sig)) < near * (double)ALPHA_MIN) continue;
                    double m = pix_margin[(size_t)py * W + px];
                    if (m < best) best = m;
                    np_++;
                }
        o_min[i] = best; o_npix[i] = np_;
    }
}

/* Test aid for the product's tile lists: the largest o*exp(-sigma) of Gaussian gid[k] over the pixel
 * centres of tile tile[k], for a list of (tile, Gaussian) pairs -- the pairs of the oracle's normative lists
 * that the product's lists do not hold.  Output preservation requires it to stay below 1/255. */
void ref_pairs_max_alpha(int64_t n_pairs, const int32_t *gid, const int32_t *tile, const real *xy, const real *conic,
                   const real *opac, const cam_t *cam, int W, int H, real *o_max_alpha) {
    int TW = (W + BLK - 1) / BLK;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t k = 0; k < n_pairs; k++) {
        int i = gid[k], ty = tile[k] / TW, tx = tile[k] % TW;
        real best = 0;
        for (int py = ty * BLK; py < (ty + 1) * BLK && py < H; py++)
            for (int px = tx * BLK; px < (tx + 1) * BLK && px < W; px++) {
                real dx = xy[2 * i] - (px + cam->pc), dy = xy[2 * i + 1] - (py + cam->pc);
                real sig = (real)0.5 * (conic[3 * i] * dx * dx + conic[3 * i + 2] * dy * dy) + conic[3 * i + 1] * dx * dy;
                if (sig < 0) continue;
                real al = opac[i] * exp(-sig);
                if (al > best) best = al;
            }
        o_max_alpha[k] = best;
    }
}

/* B.7 backward blend.  v_rgb_img [H,W,3], v_depth_img [H,W] (w.r.t. depth_acc), v_alpha_img [H,W].
 * Accumulates (+=) into v_xy [N,2], v_conic [N,3], v_opac [N], v_rgb [N,3], v_depth [N] (caller zeroes). */
void ref_blend_bwd_range(const real *xy, const real *conic, const real *opac, const real *rgb, const real *depth,
                   const int32_t *sorted_gid, const int64_t *tile_start, const cam_t *cam, int W, int H,
                   const real *f_T, const int32_t *f_idx,
                   const real *v_rgb_img, const real *v_depth_img, const real *v_alpha_img,
                   real *v_xy, real *v_conic, real *v_opac, real *v_rgb, real *v_depth, int t0, int t1) {
    int TW = (W + BLK - 1) / BLK;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = t0; t < t1; t++) {
        int ty = t / TW, tx = t % TW;
        int64_t s = tile_start[t];
        for (int py = ty * BLK; py < (ty + 1) * BLK && py < H; py++)
            for (int px = tx * BLK; px < (tx + 1) * BLK && px < W; px++) {
                size_t p = (size_t)py * W + px;
                real fxp = px + cam->pc, fyp = py + cam->pc;
                real Tfin = f_T[p], T = Tfin;
                real vC[3] = {v_rgb_img[3 * p], v_rgb_img[3 * p + 1], v_rgb_img[3 * p + 2]};
                real vD = v_depth_img[p], vA = v_alpha_img[p];
                real S[3] = {0, 0, 0}, SD = 0;
                real bgdot = cam->bg[0] * vC[0] + cam->bg[1] * vC[1] + cam->bg[2] * vC[2];
                for (int k = f_idx[p]; k >= 0; k--) {
                    int g = sorted_gid[s + k];
                    real dx = xy[2 * g] - fxp, dy = xy[2 * g + 1] - fyp;
                    real a = conic[3 * g], b = conic[3 * g + 1], c = conic[3 * g + 2];
                    real sig = (real)0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy;
                    if (sig < 0) continue;
                    real ex = exp(-sig), al = opac[g] * ex; if (al > ALPHA_MAX) al = ALPHA_MAX;
                    if (al < ALPHA_MIN) continue;
                    real ra = 1 / (1 - al);
                    T *= ra;
                    real w = al * T;
                    real valpha = 0;
                    for (int ch = 0; ch < 3; ch++) {
                        real cg = rgb[3 * g + ch];
#pragma omp atomic
                        v_rgb[3 * g + ch] += w * vC[ch];
                        valpha += (cg * T - S[ch] * ra) * vC[ch];
                        S[ch] += w * cg;
                    }
#pragma omp atomic
                    v_depth[g] += w * vD;
                    valpha += (depth[g] * T - SD * ra) * vD;
                    SD += w * depth[g];
                    valpha += Tfin * ra * (vA - bgdot);
                    real vsig = -opac[g] * ex * valpha;
#pragma omp atomic
                    v_opac[g] += ex * valpha;
#pragma omp atomic
                    v_conic[3 * g] += (real)0.5 * vsig * dx * dx;
#pragma omp atomic
                    v_conic[3 * g + 1] += vsig * dx * dy;
#pragma omp atomic
                    v_conic[3 * g + 2] += (real)0.5 * vsig * dy * dy;
#pragma omp atomic
                    v_xy[2 * g] += vsig * (a * dx + b * dy);
#pragma omp atomic
                    v_xy[2 * g + 1] += vsig * (b * dx + c * dy);
                }
            }
    }
}

void ref_blend_bwd(const real *xy, const real *conic, const real *opac, const real *rgb, const real *depth,
                   const int32_t *sorted_gid, const int64_t *tile_start, const cam_t *cam, int W, int H,
                   const real *f_T, const int32_t *f_idx,
                   const real *v_rgb_img, const real *v_depth_img, const real *v_alpha_img,
                   real *v_xy, real *v_conic, real *v_opac, real *v_rgb, real *v_depth) {
    int TW = (W + BLK - 1) / BLK, TH = (H + BLK - 1) / BLK;
    ref_blend_bwd_range(xy, conic, opac, rgb, depth, sorted_gid, tile_start, cam, W, H, f_T, f_idx,
                        v_rgb_img, v_depth_img, v_alpha_img, v_xy, v_conic, v_opac, v_rgb, v_depth, 0, TW * TH);
}

/* Test aid: the UN-CANCELLED magnitude of every screen-space gradient of ref_blend_bwd_range -- the same
 * walk with the absolute value of every term that is added anywhere (inside dL/dalpha too): m_* >= |v_*|.
 * A floating-point evaluation of a sum is accurate relative to the sum of the magnitudes of its terms, not
 * relative to a result that cancels; with random upstream gradients a Gaussian's ~200 pixel contributions
 * cancel to 1e-3 of their magnitude for one Gaussian in a thousand, which is all the tail of a plain relative
 * error shows.  The full-size gradient tests therefore bound |HIP - oracle| by tol * m (the componentwise
 * forward-error measure of a sum).  Accumulates (+=); the caller zeroes. */
void ref_blend_bwd_mass_range(const real *xy, const real *conic, const real *opac, const real *rgb, const real *depth,
                   const int32_t *sorted_gid, const int64_t *tile_start, const cam_t *cam, int W, int H,
                   const real *f_T, const int32_t *f_idx,
                   const real *v_rgb_img, const real *v_depth_img, const real *v_alpha_img,
                   real *m_xy, real *m_conic, real *m_opac, real *m_rgb, real *m_depth, int t0, int t1) {
    int TW = (W + BLK - 1) / BLK;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = t0; t < t1; t++) {
        int ty = t / TW, tx = t % TW;
        int64_t s = tile_start[t];
        for (int py = ty * BLK; py < (ty + 1) * BLK && py < H; py++)
            for (int px = tx * BLK; px < (tx + 1) * BLK && px < W; px++) {
                size_t p = (size_t)py * W + px;
                real fxp = px + cam->pc, fyp = py + cam->pc;
                real Tfin = f_T[p], T = Tfin;
                real vC[3] = {fabs(v_rgb_img[3 * p]), fabs(v_rgb_img[3 * p + 1]), fabs(v_rgb_img[3 * p + 2])};
                real vD = fabs(v_depth_img[p]), vA = fabs(v_alpha_img[p]);
                real S[3] = {0, 0, 0}, SD = 0;
                real bgdot = fabs(cam->bg[0]) * vC[0] + fabs(cam->bg[1]) * vC[1] + fabs(cam->bg[2]) * vC[2];
                for (int k = f_idx[p]; k >= 0; k--) {
                    int g = sorted_gid[s + k];
                    real dx = xy[2 * g] - fxp, dy = xy[2 * g + 1] - fyp;
                    real a = conic[3 * g], b = conic[3 * g + 1], c = conic[3 * g + 2];
                    real sig = (real)0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy;
                    if (sig < 0) continue;
                    real ex = exp(-sig), al = opac[g] * ex; if (al > ALPHA_MAX) al = ALPHA_MAX;
                    if (al < ALPHA_MIN) continue;
                    real ra = 1 / (1 - al);
                    T *= ra;
                    real w = al * T;
                    real valpha = 0;
                    for (int ch = 0; ch < 3; ch++) {
                        real cg = fabs(rgb[3 * g + ch]);
#pragma omp atomic
                        m_rgb[3 * g + ch] += w * vC[ch];
                        valpha += (cg * T + S[ch] * ra) * vC[ch];
                        S[ch] += w * cg;
                    }
#pragma omp atomic
                    m_depth[g] += w * vD;
                    valpha += (fabs(depth[g]) * T + SD * ra) * vD;
                    SD += w * fabs(depth[g]);
                    valpha += Tfin * ra * (vA + bgdot);
                    real vsig = opac[g] * ex * valpha;
#pragma omp atomic
                    m_opac[g] += ex * valpha;
#pragma omp atomic
                    m_conic[3 * g] += (real)0.5 * vsig * dx * dx;
#pragma omp atomic
                    m_conic[3 * g + 1] += vsig * fabs(dx * dy);
#pragma omp atomic
                    m_conic[3 * g + 2] += (real)0.5 * vsig * dy * dy;
#pragma omp atomic
                    m_xy[2 * g] += vsig * (fabs(a * dx) + fabs(b * dy));
#pragma omp atomic
                    m_xy[2 * g + 1] += vsig * (fabs(b * dx) + fabs(c * dy));
                }
            }
    }
}

/* B.8 projection + SH backward.  Inputs: per-Gaussian v_xy, v_conic, v_opac, v_rgb, v_depth.
 * Outputs (overwritten): v_means [N,3], v_log_scales [N,3], v_quats [N,4], v_opac_logit [N],
 * v_sh [N,K_stride,3]. */
void ref_project_bwd(int N, const real *means, const real *log_scales, const real *quats,
                     const real *opac_logit, const real *sh, int K_stride, int sh_deg,
                     const cam_t *cam, int W, int H, real glob_scale, const int32_t *radius,
                     const real *v_xy, const real *v_conic, const real *v_opac, const real *v_rgb,
                     const real *v_depth,
                     real *v_means, real *v_ls, real *v_quats, real *v_ol, real *v_sh) {
    const real *V = cam->V;
    real campos[3];
    for (int j = 0; j < 3; j++) campos[j] = -(V[0 + j] * V[3] + V[4 + j] * V[7] + V[8 + j] * V[11]);
    real limx = (real)1.3 * (W / (real)2) / cam->fx, limy = (real)1.3 * (H / (real)2) / cam->fy;
    int K = (sh_deg + 1) * (sh_deg + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        const real *m = means + 3 * i;
        real vm[3] = {0, 0, 0};
        for (int j = 0; j < 3; j++) { v_ls[3 * i + j] = 0; }
        for (int j = 0; j < 4; j++) v_quats[4 * i + j] = 0;
        real o = 1 / (1 + exp(-opac_logit[i]));
        v_ol[i] = v_opac[i] * o * (1 - o);
        if (sh) { /* B.5 backward (colour is defined for every Gaussian, visible or not) */
            real dx = m[0] - campos[0], dy = m[1] - campos[1], dz = m[2] - campos[2];
            real dn = sqrt(dx * dx + dy * dy + dz * dz);
            real d[3] = {dx / dn, dy / dn, dz / dn};
            real Y[16], dY[16][3];
            sh_eval(sh_deg, d[0], d[1], d[2], Y);
            sh_grad(sh_deg, d[0], d[1], d[2], dY);
            real vd[3] = {0, 0, 0};
            for (int k = 0; k < K_stride; k++) for (int ch = 0; ch < 3; ch++) v_sh[((size_t)i * K_stride + k) * 3 + ch] = 0;
            for (int ch = 0; ch < 3; ch++) {
                real acc = 0;
                for (int k = 0; k < K; k++) acc += Y[k] * sh[((size_t)i * K_stride + k) * 3 + ch];
                if (!(acc + (real)0.5 > 0)) continue; /* clamp active -> zero gradient */
                real g = v_rgb[3 * i + ch];
                for (int k = 0; k < K; k++) {
                    v_sh[((size_t)i * K_stride + k) * 3 + ch] = Y[k] * g;
                    real ck = sh[((size_t)i * K_stride + k) * 3 + ch] * g;
                    vd[0] += dY[k][0] * ck; vd[1] += dY[k][1] * ck; vd[2] += dY[k][2] * ck;
                }
            }
            real dot = d[0] * vd[0] + d[1] * vd[1] + d[2] * vd[2];
            for (int j = 0; j < 3; j++) vm[j] += (vd[j] - d[j] * dot) / dn;
        }
        if (radius[i] > 0) {
            real tx = V[0] * m[0] + V[1] * m[1] + V[2] * m[2] + V[3];
            real ty = V[4] * m[0] + V[5] * m[1] + V[6] * m[2] + V[7];
            real tz = V[8] * m[0] + V[9] * m[1] + V[10] * m[2] + V[11];
            real R[9], qn[4], nrm;
            quat_rot(quats + 4 * i, R, qn, &nrm);
            real s[3];
            for (int j = 0; j < 3; j++) s[j] = exp(log_scales[3 * i + j]) * glob_scale;
            real M[9], S[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[3 * r + c] = R[3 * r + c] * s[c];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
                real a = 0; for (int k = 0; k < 3; k++) a += M[3 * r + k] * M[3 * c + k];
                S[3 * r + c] = a;
            }
            real ux = tx / tz, uy = ty / tz;
            int inx = (ux >= -limx && ux <= limx), iny = (uy >= -limy && uy <= limy);
            real ucx = ux < -limx ? -limx : (ux > limx ? limx : ux);
            real ucy = uy < -limy ? -limy : (uy > limy ? limy : uy);
            real fx = cam->fx, fy = cam->fy;
            real J[6] = {fx / tz, 0, -fx * ucx / tz, 0, fy / tz, -fy * ucy / tz};
            real Tm[6];
            for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++)
                Tm[3 * r + c] = J[3 * r] * V[c] + J[3 * r + 1] * V[4 + c] + J[3 * r + 2] * V[8 + c];
            /* conic -> covariance gradient (full symmetric matrix G) */
            real a = 0, b = 0, c = 0;
            {
                real TS[6];
                for (int r = 0; r < 2; r++) for (int cc = 0; cc < 3; cc++) {
                    real acc = 0; for (int k = 0; k < 3; k++) acc += Tm[3 * r + k] * S[3 * k + cc];
                    TS[3 * r + cc] = acc;
                }
                real c00 = TS[0] * Tm[0] + TS[1] * Tm[1] + TS[2] * Tm[2] + BLUR;
                real c01 = TS[0] * Tm[3] + TS[1] * Tm[4] + TS[2] * Tm[5];
                real c11 = TS[3] * Tm[3] + TS[4] * Tm[4] + TS[5] * Tm[5] + BLUR;
                real det = c00 * c11 - c01 * c01;
                a = c11 / det; b = -c01 / det; c = c00 / det;
            }
            real va = v_conic[3 * i], vb = v_conic[3 * i + 1], vc = v_conic[3 * i + 2];
            real G00 = -(a * a * va + a * b * vb + b * b * vc);
            real G11 = -(b * b * va + b * c * vb + c * c * vc);
            real G01 = (real)-0.5 * (2 * a * b * va + (a * c + b * b) * vb + 2 * b * c * vc);
            real G[4] = {G00, G01, G01, G11};
            /* v_Sigma = Tm^T G Tm (3x3, symmetric);  v_Tm = 2 G Tm Sigma */
            real GT[6];
            for (int r = 0; r < 2; r++) for (int cc = 0; cc < 3; cc++)
                GT[3 * r + cc] = G[2 * r] * Tm[cc] + G[2 * r + 1] * Tm[3 + cc];
            real vS[9];
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++)
                vS[3 * r + cc] = Tm[r] * GT[cc] + Tm[3 + r] * GT[3 + cc];
            real vTm[6];
            for (int r = 0; r < 2; r++) for (int cc = 0; cc < 3; cc++) {
                real acc = 0; for (int k = 0; k < 3; k++) acc += GT[3 * r + k] * S[3 * k + cc];
                vTm[3 * r + cc] = 2 * acc;
            }
            /* v_J = v_Tm Rw^T */
            real vJ[6];
            for (int r = 0; r < 2; r++) for (int cc = 0; cc < 3; cc++)
                vJ[3 * r + cc] = vTm[3 * r] * V[4 * cc] + vTm[3 * r + 1] * V[4 * cc + 1] + vTm[3 * r + 2] * V[4 * cc + 2];
            real vt[3] = {0, 0, 0};
            real tz2 = tz * tz;
            vt[2] += vJ[0] * (-fx / tz2) + vJ[4] * (-fy / tz2);
            if (inx) { vt[0] += vJ[2] * (-fx / tz2); vt[2] += vJ[2] * (2 * fx * tx / (tz2 * tz)); }
            else vt[2] += vJ[2] * (fx * ucx / tz2);
            if (iny) { vt[1] += vJ[5] * (-fy / tz2); vt[2] += vJ[5] * (2 * fy * ty / (tz2 * tz)); }
            else vt[2] += vJ[5] * (fy * ucy / tz2);
            /* mean2d and depth */
            vt[0] += v_xy[2 * i] * fx / tz; vt[2] += -v_xy[2 * i] * fx * tx / tz2;
            vt[1] += v_xy[2 * i + 1] * fy / tz; vt[2] += -v_xy[2 * i + 1] * fy * ty / tz2;
            vt[2] += v_depth[i];
            for (int j = 0; j < 3; j++) vm[j] += V[j] * vt[0] + V[4 + j] * vt[1] + V[8 + j] * vt[2];
            /* Sigma = M M^T: v_M = 2 v_Sigma M */
            real vM[9];
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) {
                real acc = 0; for (int k = 0; k < 3; k++) acc += vS[3 * r + k] * M[3 * k + cc];
                vM[3 * r + cc] = 2 * acc;
            }
            real vR[9];
            for (int j = 0; j < 3; j++) {
                real vs = R[j] * vM[j] + R[3 + j] * vM[3 + j] + R[6 + j] * vM[6 + j];
                v_ls[3 * i + j] = vs * s[j];
                vR[j] = vM[j] * s[j]; vR[3 + j] = vM[3 + j] * s[j]; vR[6 + j] = vM[6 + j] * s[j];
            }
            real w = qn[0], x = qn[1], y = qn[2], z = qn[3];
            real vq[4];
            vq[0] = 2 * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
            vq[1] = 2 * (-2 * x * (vR[4] + vR[8]) + y * (vR[1] + vR[3]) + z * (vR[2] + vR[6]) + w * (vR[7] - vR[5]));
            vq[2] = 2 * (x * (vR[1] + vR[3]) - 2 * y * (vR[0] + vR[8]) + z * (vR[5] + vR[7]) + w * (vR[2] - vR[6]));
            vq[3] = 2 * (x * (vR[2] + vR[6]) + y * (vR[5] + vR[7]) - 2 * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
            real dot = qn[0] * vq[0] + qn[1] * vq[1] + qn[2] * vq[2] + qn[3] * vq[3];
            for (int j = 0; j < 4; j++) v_quats[4 * i + j] = (vq[j] - qn[j] * dot) / nrm;
        }
        for (int j = 0; j < 3; j++) v_means[3 * i + j] = vm[j];
    }
}

int ref_real_bytes(void) { return (int)sizeof(real); }
#ifdef _OPENMP
#include <omp.h>
int ref_num_threads(void) { return omp_get_max_threads(); }
#else
int ref_num_threads(void) { return 1; }
#endif
