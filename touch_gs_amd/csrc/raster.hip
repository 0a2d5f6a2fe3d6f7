// raster.hip -- K6 per-tile front-to-back RGB+depth compositing (forward) and K7 its backward
// with the tactile depth/uncertainty loss fused in.   gfx950, wave64.
//
// Spec: SURVEY.md App. B.6 (forward) / B.7 (backward).  Mirrors the op the reference's training
// loop reaches through `ns-train depth-gaussian-splatting` (scripts/train_bunny_real.sh:52):
// gsplat `rasterize_gaussians` (App. A.2) -- but RGB and depth are composited in ONE pass instead
// of the two full rasterizations Splatfacto issues.
//
// MI355X mapping (not a warp-shaped CUDA tiling; the forward's default since round 3 is the 4x4-block form
// further down, k_raster_fwd_blocks -- this describes the quadrant form, which the backward uses):
//  * one wave64 owns one 16x16 tile.  Lane l owns 4 pixels, one per 8x8 quadrant ("slot" k):
//    (x, y) = (8*(k&1) + (l&7), 8*(k>>1) + (l>>3)).  No workgroup barriers, no cross-wave traffic.
//  * the tile's Gaussian list is staged 64 records at a time through LDS.  While staging, lane j
//    turns Gaussian j into a tile-centred quadratic  s(u,v) = c0 + c1 u + c2 v + c3 u^2 + c4 uv +
//    c5 v^2  with log2(e) and -log2(opacity) folded in, so that alpha = exp2(-s) costs 5 FMAs + one
//    v_exp per pixel, and into a 4-bit quadrant mask (which 8x8 quadrants can reach
//    alpha >= 1/255).  The blend loop reads the record back as wave-uniform broadcast
//    ds_read_b128 and skips whole quadrants with scalar branches; the per-pixel update is
//    branch-free (v_cndmask), so the 4 pixels of a lane pipeline through the exp unit.
//  * backward (k_raster_bwd, round 4): walks the list BACK TO FRONT from the per-pixel stop positions the forward
//    leaves (the forward's alpha compare on the same bits: identical decisions; T in front of a Gaussian =
//    T behind * rcp(1 - alpha)); k_raster_bwd_f2b is the front-to-back form of rounds 1-3 (colour behind =
//    final - prefix: 300x less accurate for occluded Gaussians, see the comment at k_raster_bwd).  Per-pixel state
//    is (T, sum behind . v) only; each lane accumulates, over its <= 4 pixels,
//    8 sums per Gaussian (v_rgb, v_depth, and S / Sx / Sy / Quv of q = alpha * v_alpha, from which the
//    six pixel-coordinate moments follow linearly); the 64-lane sums go through LDS: every lane scatters
//    its 8 values into a padded [8][68] image, lane (part, c) forms a weighted sum of 16 lane
//    contributions read as four conflict-free ds_read_b128 (the weights turn S, Sx, Sy into the moments),
//    two lane-swap folds (v_permlane16/32_swap) combine the four parts and term c lands in slot c
//    of an LDS transpose buffer (16 FMAs + two 3-instruction lane-swap folds + 9 LDS instructions; a multiplexed DPP
//    butterfly needs 37 VALU and the kernel is VALU bound); at the end of the batch lane j converts Gaussian j's moments into (v_xy, v_conic, v_opacity) and stores
//    one 48-B partial record.  There are NO float atomics: cross-tile accumulation is a segmented
//    sum in K8 (deterministic, and it avoids cross-XCD memory-side atomics).
//  * blockIdx -> tile mapping (tile_order, built by the sort launch; tgs_common.h): block b runs on XCD b % 8; the
//    row-major tiles are dealt to the XCDs in granules of 8 consecutive tiles (horizontal neighbours share most of
//    their Gaussians and stay in one L2; any view is balanced across the XCDs), each XCD visiting its tiles longest
//    list first.
#include <stdlib.h>
#include <type_traits>
#include "tgs_common.h"

namespace {

constexpr float ALPHA_MAX = 0.999f;
// K7 runs the copy of its loop without the 0.999 clamp only for batches whose opacities all stay below
// this bound: exp2(-s) = o e^-sigma can exceed o by rounding (sigma >= 0 only up to a few ulp of the
// tile-centred quadratic), and the forward clamps whenever e > 0.999 -- the 1 % margin keeps the two
// kernels' transmittance sequences bit-identical (and e == 1, T' = 0 out of the rcp).
constexpr float CLAMP_FREE_OPACITY = 0.99f;
constexpr float T_STOP = 1e-4f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LOG2_255 = 7.994353436858858f;   // alpha >= 1/255  <=>  s <= log2(255)

__device__ __forceinline__ int xcd_tile(int b, int T) {   // schedule without a tile_order
  return tgs_xcd_slot_tile(T, b & 7, b >> 3);
}

// v[l] + v[l ^ 16] and v[l] + v[l ^ 32] with the gfx950 lane-swap instructions (3 VALU each: a
// copy, v_permlane{16,32}_swap_b32 and the add) instead of ds_bpermute + per-iteration index
// arithmetic.  permlane16_swap exchanges the odd rows of its first operand with the even rows of the
// second, permlane32_swap the upper half of the first with the lower half of the second.
typedef unsigned tgs_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fold_xor16(float v) {
  const tgs_u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float fold_xor32(float v) {
  const tgs_u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

struct LossK {
  const float* gt_rgb;
  const float* gt_depth;
  const float* unc;
  float l1w, dw, uw, eps;
  int on;
};

// Gaussian -> tile-centred quadratic + quadrant mask.  (gx, gy) = centre relative to the tile
// centre, in pixel-centre coordinates (pixel (i,j) of the tile sits at u = i - 7.5, v = j - 7.5).
struct TileRec {
  float4 a;  // c0 c1 c2 c3
  float4 b;  // c4 c5 depth r        (the ten floats the blend loops read are contiguous:
  float4 c;  // g b - mask(bits)      ds_read_b128 x2 + ds_read_b64 off one base address)
};

// Gaussian centre relative to the centre of tile (tx, ty).  The record holds the screen position
// relative to the origin of the Gaussian's own tile rect (K1, project.hip: a small number, so its fp32
// ulp is ~1e-6 px instead of the 2.4e-4 px of an absolute 4K coordinate); the tile centre relative to
// that origin, 16 (tx - x0) + 7.5 + pix_center, is exact.
__device__ __forceinline__ void centre_rel(float4 r0, float4 r2, int tx, int ty, float pix_center, float& gx, float& gy) {
  const unsigned rect = __float_as_uint(r2.z);
  const int dx = tx - (int)(rect & 255u), dy = ty - (int)((rect >> 8) & 255u);
  gx = r0.x - ((float)(dx * TGS_BLOCK) + (7.5f + pix_center));
  gy = r0.y - ((float)(dy * TGS_BLOCK) + (7.5f + pix_center));
}

template <bool QUAD_MASK = true>
__device__ __forceinline__ TileRec make_tile_rec(float4 r0, float4 r1, float4 r2, float gx, float gy) {
  // r0 = {x - 16 x0, y - 16 y0, depth, opac}  r1 = {a, b, c, r}  r2 = {g, b, rect, off};  (gx, gy) = centre_rel()
  const float A = r1.x, B = r1.y, Cc = r1.z;
  const float L = -__log2f(r0.w);
  TileRec t;
  t.a.x = LOG2E * (0.5f * A * gx * gx + B * gx * gy + 0.5f * Cc * gy * gy) + L;
  t.a.y = -LOG2E * (A * gx + B * gy);
  t.a.z = -LOG2E * (B * gx + Cc * gy);
  t.a.w = 0.5f * LOG2E * A;
  t.b.x = LOG2E * B;
  t.b.y = 0.5f * LOG2E * Cc;
  t.b.z = r0.z;
  t.b.w = r1.w;
  // quadrant mask: does {alpha >= 1/255} = {s <= log2 255} reach any pixel centre of the 8x8
  // quadrant?  Exact test: minimise the convex quadratic s over the quadrant's rectangle of pixel
  // centres (interior point if the centre projects inside, else the best point on the 4 edges).
  unsigned mask = 0u;
  const float slack = LOG2_255 + 1e-3f;
  if (QUAD_MASK && L < LOG2_255) {
    const float c0 = t.a.x, c1 = t.a.y, c2 = t.a.z, c3 = t.a.w, c4 = t.b.x, c5 = t.b.y;
    // approximate reciprocals suffice: the test carries a 1e-3 slack and only decides skipping
    const float ic3 = __builtin_amdgcn_rcpf(c3), ic5 = __builtin_amdgcn_rcpf(c5);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float u0 = (k & 1) ? 0.5f : -7.5f, u1 = u0 + 7.f;
      const float v0 = (k >> 1) ? 0.5f : -7.5f, v1 = v0 + 7.f;
      float best;
      if (gx >= u0 && gx <= u1 && gy >= v0 && gy <= v1) {
        best = L;  // the centre itself (s = L there) lies inside the rectangle
      } else {
        // edges v = const: s(u) = c3 u^2 + (c1 + c4 v) u + (c0 + c2 v + c5 v^2), minimised at clamped u*
        best = 3.0e38f;
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float v = e ? v1 : v0;
          const float lin = fmaf(c4, v, c1);
          const float us = fminf(fmaxf(-0.5f * lin * ic3, u0), u1);
          best = fminf(best, fmaf(us, fmaf(c3, us, lin), fmaf(v, fmaf(c5, v, c2), c0)));
          const float u = e ? u1 : u0;
          const float lin2 = fmaf(c4, u, c2);
          const float vs = fminf(fmaxf(-0.5f * lin2 * ic5, v0), v1);
          best = fminf(best, fmaf(vs, fmaf(c5, vs, lin2), fmaf(u, fmaf(c3, u, c1), c0)));
        }
      }
      mask |= (best <= slack) ? (1u << k) : 0u;
    }
  }
  t.c = make_float4(r2.x, r2.y, 0.f, __uint_as_float(mask));
  return t;
}

// per-lane pixel constants of slot k
struct PixConst {
  float u[2], v[2], uu[2], vv[2], uv[4];
};

__device__ __forceinline__ PixConst make_pix_const(int lane) {
  PixConst p;
  p.u[0] = (float)(lane & 7) - 7.5f; p.u[1] = p.u[0] + 8.f;
  p.v[0] = (float)(lane >> 3) - 7.5f; p.v[1] = p.v[0] + 8.f;
#pragma unroll
  for (int i = 0; i < 2; i++) { p.uu[i] = p.u[i] * p.u[i]; p.vv[i] = p.v[i] * p.v[i]; }
#pragma unroll
  for (int k = 0; k < 4; k++) p.uv[k] = p.u[k & 1] * p.v[k >> 1];
  return p;
}

__device__ __forceinline__ float eval_s(const float4& a, const float4& b, const PixConst& p, int k) {
  float s = fmaf(a.y, p.u[k & 1], a.x);
  s = fmaf(a.z, p.v[k >> 1], s);
  s = fmaf(a.w, p.uu[k & 1], s);
  s = fmaf(b.x, p.uv[k], s);
  s = fmaf(b.y, p.vv[k >> 1], s);
  return s;
}

// Entry of the per-batch quadrant bitmaps K6 leaves for K7 (4 x 64 bits per batch of 64 list
// positions): batches of one tile start at positions start, start + 64, ..; (position >> 6) + tile is
// strictly increasing over the batches of a tile and from one tile to the next, hence unique, and
// < (#intersections >> 6) + #tiles + 1.
__device__ __forceinline__ size_t slot_ok_index(int batch_pos, int tile) {
  return (size_t)(batch_pos >> 6) + (size_t)tile;
}

// smax of a stopped pixel: 0xC0000000 | list position (relative to the tile's first entry, < 2^30).  As a
// float that is <= -2 or NaN -- dead either way (`s <= smax` and `smax > 0` are false).  A pixel that never
// stopped keeps smax = log2(255) > 0 and reports TGS_NO_STOP.
#define TGS_NO_STOP 0x7fffffff
__device__ __forceinline__ float stop_code(int pos) { return __uint_as_float(0xC0000000u | (unsigned)pos); }
__device__ __forceinline__ int stop_of(float smax) {
  return smax > 0.f ? TGS_NO_STOP : (int)(__float_as_uint(smax) & 0x3fffffffu);
}
// 4x4-block form: the code of a stop at entry j (< 64) of the CURRENT batch -- 0xE0000000 | j (= -2^65 (1 + ..): dead
// like stop_code) -- and, at the end of the batch, its conversion to stop_code(batch base + j).  "Fresh" codes are the
// words 0xE00000xx; the out-of-image sentinel -3e38 = 0xFF61B1E6 and finished codes 0xC....... are left alone, and
// list positions stay below 2^29.
__device__ __forceinline__ float batch_stop_code(int j) { return __uint_as_float(0xE0000000u | (unsigned)j); }
__device__ __forceinline__ float finish_batch_stop(float smax, int batch_rel) {
  const unsigned b = __float_as_uint(smax);
  return (b & 0xffffff00u) == 0xE0000000u ? stop_code(batch_rel + (int)(b & 0xffu)) : smax;
}

// The frame's deepest walk -- how far into its list the busiest tile's pixels reach, = the length of K7's longest
// chain -- is what decides whether K7 splits its tiles over four waves (frame_is_chain_bound).  Every K6 block folds
// its tile's walk into two of the words behind the tile starts (maximum and sum, TGS_WALK_AT / TGS_WALKSUM_AT(T, XCC
// id): one pair per L2, the scan kernel zeroed them); a pixel that never stopped walks the whole list.  Pixels outside the image keep smax > 0.
// (TGS_WALK_AT / TGS_WALKSUM_AT / TGS_SLOTCTR_AT: layout of the scratch ints behind the tile starts, tgs_common.h)
// The host entry points take tile_start as a mutable buffer of tgs_tile_start_len ints (tgs.h); the kernels keep a const
// __restrict__ view of it for the starts (scalar loads) and reach the scratch words behind them through this one cast.
__device__ __forceinline__ int32_t* frame_scratch(const int32_t* tile_start) { return const_cast<int32_t*>(tile_start); }
__device__ __forceinline__ void publish_walk(const int32_t* __restrict__ tile_start, int T_total, const float (&smax)[4],
                                             int n, int lane) {
  int wl = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)   // a pixel outside the image (smax = the -3e38 sentinel, never a stop code) walks nothing
    wl = max(wl, smax[k] == -3.0e38f ? 0 : min(stop_of(smax[k]), n));
  wl = wave_minmax_i<true>(wl);
  if (lane == 0 && wl > 0) {
    const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & (TGS_WALK_WORDS - 1));
    atomicMax(frame_scratch(tile_start) + TGS_WALK_AT(T_total, xcc), wl);
    atomicAdd(frame_scratch(tile_start) + TGS_WALKSUM_AT(T_total, xcc), wl);   // sum of the walks = K7's work
  }
}

// ---------------------------------------------------------------------------------------------
// K6 forward
// ---------------------------------------------------------------------------------------------
// One compositing step of pixel slot k against the staged Gaussian (qa, qb, qc); shared verbatim
// by the forward and the backward so that both take bit-identical threshold decisions.
//   smax : log2(255) while the pixel is live, -inf once it has stopped (folds `live` into the
//          alpha >= 1/255 test).  The sigma < 0 skip of App. B.6 cannot trigger: K1 only emits
//          positive-definite conics, for which sigma >= 0 up to rounding.
//   returns the contributing alpha (0 if skipped or if this Gaussian stops the pixel).
//   MAYCLAMP = false: the caller knows that no Gaussian of the batch has opacity > 0.999, so
//          exp2(-s) = o e^-sigma <= 0.999 and min(0.999, .) is the identity -- bit-identical, one VALU less.
//   okb  : (WANT_OK) lane mask of the pixels that passed the alpha >= 1/255 test while live -- the ballot
//          sits next to the compare so that it IS the compare's scalar result (no extra VALU)
//   okb  : WANT_OK = 1: lane mask of `ok`; WANT_OK = 2: lane mask of `go` (both ballots sit next to their
//          compares, so they ARE the compares' scalar results: no extra VALU)
//   stopv: what smax becomes when THIS Gaussian stops the pixel -- a negative float (so the pixel is dead for
//          every later `s <= smax`, s >= 0 up to rounding) that carries the list position: stop_code(position)
//          where the position is wave-uniform (quadrant forms: an SGPR operand), batch_stop_code(j) in the
//          4x4-block form, whose rows walk different entries -- there the batch-relative index j is the lane's
//          own (it addressed the record) and one v_or with a literal makes the code; the batch's base is added
//          once per batch (finish_batch_stops).  The forward stores the position per pixel (stop_pos) and the
//          backward, which walks the list back to front, starts each pixel there.
//          Round 4 kept stop_code(position) in a spare slot of the staged record instead: no VALU at all, but the
//          block form's per-lane record read grew from 40 to 44 B (ds_read_b96 for b64) and that loop sits at ~85 %
//          of the CU's LDS bandwidth: K6 +10 % (156 vs 141.7 us 8-view mean, same box, profiles/r5_ab_runs.txt).
template <bool MAYCLAMP = true, int WANT_OK = 0>
__device__ __forceinline__ float blend_step(float s, float& T, float& smax, float& Tnew, bool& go,
                                            unsigned long long& okb, float stopv = -3.0e38f) {
  const float e = __builtin_amdgcn_exp2f(-s);
  const float al = MAYCLAMP ? fminf(ALPHA_MAX, e) : e;
  const bool ok = s <= smax;
  if constexpr (WANT_OK == 1) okb = __builtin_amdgcn_ballot_w64(ok);
  Tnew = fmaf(-al, T, T);
  const bool above = Tnew > T_STOP;
  if constexpr (WANT_OK == 2) okb = __builtin_amdgcn_ballot_w64(ok) & __builtin_amdgcn_ballot_w64(above);
  go = ok & above;
  smax = (ok != go) ? stopv : smax;  // stop (ok and not go; go implies ok): T' <= 1e-4, this Gaussian excluded
  return go ? al : 0.f;
}

template <bool WANT_IDX, bool WANT_OK>
__global__ __launch_bounds__(64) void k_raster_fwd(
    CamK cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ sorted_gid,
    const int32_t* __restrict__ tile_start, float* __restrict__ out_rgb,
    float* __restrict__ out_depth, float* __restrict__ final_T, int32_t* __restrict__ final_idx,
    const int32_t* __restrict__ tile_order, unsigned long long* __restrict__ slot_ok,
    int32_t* __restrict__ stop_pos) {
  const int tile = tile_order ? tile_order[blockIdx.x] : xcd_tile(blockIdx.x, T_total);
  if (tile >= T_total) return;
  const int lane = threadIdx.x;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  const PixConst pc = make_pix_const(lane);
  int pxi[4], pyi[4];
  float smax[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    pxi[k] = tx * TGS_BLOCK + 8 * (k & 1) + (lane & 7);
    pyi[k] = ty * TGS_BLOCK + 8 * (k >> 1) + (lane >> 3);
    smax[k] = ((pxi[k] < cam.W) && (pyi[k] < cam.H)) ? LOG2_255 : -3.0e38f;
  }
  float T[4] = {1.f, 1.f, 1.f, 1.f};
  float Cr[4] = {0.f, 0.f, 0.f, 0.f}, Cg[4] = {0.f, 0.f, 0.f, 0.f}, Cb[4] = {0.f, 0.f, 0.f, 0.f};
  float D[4] = {0.f, 0.f, 0.f, 0.f};
  int last[4] = {-1, -1, -1, -1};

  __shared__ float4 recs[64 * 3];
  const int start = tile_start[tile], end = tile_start[tile + 1];

  for (int base = start; base < end; base += 64) {
    unsigned slot_live = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) slot_live |= (__ballot(smax[k] > 0.f) != 0ull) ? (1u << k) : 0u;
    if (slot_live == 0u) break;
    __syncthreads();
    unsigned my_mask = 0u;
    if (base + lane < end) {
      // no register prefetch of the next batch: 12 fewer VGPRs buy two more resident waves per
      // SIMD, which hide the gather latency (and the long dependent chains of the blend) better;
      // prefetching only the next batch's ids (1 VGPR) measured K7 -1 %, K6 +1 % (round 3): not kept
      const float* r = splats + (size_t)sorted_gid[base + lane] * TGS_SPLAT_FLOATS;
      const float4 q0 = ld4(r), q1 = ld4(r + 4), q2 = ld4(r + 8);
      float gx, gy;
      centre_rel(q0, q2, tx, ty, cam.pix_center, gx, gy);
      const TileRec t = make_tile_rec(q0, q1, q2, gx, gy);
      recs[lane * 3] = t.a; recs[lane * 3 + 1] = t.b; recs[lane * 3 + 2] = t.c;
      my_mask = __float_as_uint(t.c.w);
    }
    __syncthreads();
    // quadrant masks of the whole batch as four wave-uniform 64-bit ballots (bit j = Gaussian j
    // reaches quadrant k): Gaussians that reach no live quadrant are skipped by scalar bit
    // scanning, without touching LDS or the VALU
    unsigned long long qm[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
      qm[k] = ((slot_live >> k) & 1u) ? __ballot((my_mask >> k) & 1u) : 0ull;
    // okm[k]: bit j = some pixel of quadrant k passed the alpha >= 1/255 test on Gaussian j while live,
    // i.e. Gaussian j changed the state of quadrant k (contribution or stop).  The backward walks the
    // same list with the same decisions and skips every (Gaussian, quadrant) whose bit is clear: ~7 % of
    // the quadrant evaluations at cfg3 reach only pixels that have already stopped.
    unsigned long long okm[4] = {0ull, 0ull, 0ull, 0ull};
    unsigned long long rem = qm[0] | qm[1] | qm[2] | qm[3];
    while (rem) {
      const int j = __builtin_ctzll(rem);
      rem &= rem - 1;
      const unsigned m = (unsigned)((qm[0] >> j) & 1ull) | ((unsigned)((qm[1] >> j) & 1ull) << 1) |
                         ((unsigned)((qm[2] >> j) & 1ull) << 2) | ((unsigned)((qm[3] >> j) & 1ull) << 3);
      const float4 qa = recs[j * 3], qb = recs[j * 3 + 1], qc = recs[j * 3 + 2];
      const int pos = base - start + j;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (m & (1u << k)) {  // wave-uniform
          const float s = eval_s(qa, qb, pc, k);
          float Tn; bool go;
          unsigned long long okb = 0ull;
          const float al = blend_step<true, WANT_OK ? 1 : 0>(s, T[k], smax[k], Tn, go, okb, stop_code(pos));
          const float w = al * T[k];
          Cr[k] = fmaf(w, qb.w, Cr[k]); Cg[k] = fmaf(w, qc.x, Cg[k]);
          Cb[k] = fmaf(w, qc.y, Cb[k]); D[k] = fmaf(w, qb.z, D[k]);
          T[k] = go ? Tn : T[k];
          if (WANT_IDX) last[k] = go ? pos : last[k];
          if constexpr (WANT_OK) okm[k] |= (okb != 0ull) ? (1ull << j) : 0ull;
        }
      }
    }
    if (WANT_OK && lane == 0) {
      unsigned long long* o = slot_ok + 4 * slot_ok_index(base, tile);
      o[0] = okm[0]; o[1] = okm[1]; o[2] = okm[2]; o[3] = okm[3];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (pxi[k] < cam.W && pyi[k] < cam.H) {
      const size_t p = (size_t)pyi[k] * cam.W + pxi[k];
      out_rgb[3 * p] = Cr[k] + T[k] * cam.bg[0];
      out_rgb[3 * p + 1] = Cg[k] + T[k] * cam.bg[1];
      out_rgb[3 * p + 2] = Cb[k] + T[k] * cam.bg[2];
      out_depth[p] = D[k];
      final_T[p] = T[k];
      if (WANT_IDX) final_idx[p] = last[k];
      if (stop_pos) stop_pos[p] = stop_of(smax[k]);
    }
  }
  // the walk statistics are K7's (chain-bound test): a render-only forward (no stop_pos) publishes nothing -- and
  // does not disturb the sums a training forward on the same lists has left (they accumulate: one publishing forward
  // per backward; 1.5 us of K6 at cfg3)
  if (stop_pos) publish_walk(tile_start, T_total, smax, end - start, lane);
}

// ---------------------------------------------------------------------------------------------
// K6, 4x4-block form (round 3)
// ---------------------------------------------------------------------------------------------
// The quadrant form above evaluates a Gaussian on a whole 8x8 quadrant as soon as it reaches one of its
// pixels: 63 % of the evaluated lanes are pixels the Gaussian does not touch.  A wave
// instruction covers 64 pixels whatever the mask, so finer culling needs DIFFERENT Gaussians in different
// parts of one instruction.  Here the four 16-lane rows of the wave (DPP rows) own the four 4x4 blocks
// of the current quadrant and every row walks its OWN block's list: per batch of 64 staged Gaussians a
// 16-bit block mask per Gaussian (exact: the u-extent of the ellipse {alpha >= 1/255} over each block
// row's v-range against the blocks' pixel-centre rectangles) is compacted into 16 index lists in LDS;
// quadrant s then runs max(len of its 4 lists) iterations -- never more than the quadrant form's
// |union| -- in which lane l reads the Gaussian its row is at (per-lane LDS address, padded with a null
// record).  No per-Gaussian bit scan and no quadrant tests: ~3 scalar instructions per iteration
// instead of ~10.  Same eval_s / blend_step on the same pixel constants in the same list order: the
// images are bit-identical to the quadrant form (test_k6_block_form_is_bit_identical).
// Only the forward can do this: the backward's per-Gaussian sums would have to be accumulated across
// rows and iterations (5.1: ds_add_f32 costs 194 cycles).
__device__ __forceinline__ unsigned row_bits(float ulo, float uhi) {
  // blocks bx (pixel centres u in [4 bx - 7.5, 4 bx - 4.5]) that meet [ulo, uhi]: bx in [lo, h1 - 1];
  // an empty range (h1 <= lo) gives 0 by itself
  const int lo = min(max((int)ceilf(fminf(fmaf(ulo, 0.25f, 1.125f), 8.f)), 0), 4);
  const int h1 = min(max((int)floorf(fmaxf(fmaf(uhi, 0.25f, 2.875f), -8.f)), 0), 4);   // hi + 1
  return ((1u << h1) - 1u) & ~((1u << lo) - 1u);
}

// 16-bit mask of the 4x4 blocks (bit 4 by + bx) in which alpha >= 1/255 can hold at a pixel centre.
// (gx, gy) = centre relative to the tile centre, (A, B, C) = conic, L = -log2(opacity).
__device__ __forceinline__ unsigned block_mask16(float gx, float gy, float A, float B, float C, float L) {
  if (!(L < LOG2_255)) return 0u;
  // region: A du^2 + 2 B du dv + C dv^2 <= t2,  t2 = 2 ln(255 o) (+ slack)
  const float t2 = 2.0f * (LOG2_255 + 2e-3f - L) * 0.6931471805599453f;
  const float M = 0.02f;                                      // margin in pixels (rounding of the roots)
  const float iA = __builtin_amdgcn_rcpf(A), iC = __builtin_amdgcn_rcpf(C);
  const float det = fmaxf(A * C - B * B, 1e-30f);
  const float dumax = sqrtf(t2 * C * __builtin_amdgcn_rcpf(det));   // horizontal half extent of the ellipse
  const float vr = gy - B * iC * dumax, vl = gy + B * iC * dumax;    // rows of its rightmost / leftmost point
  // On the line v the ellipse spans  du = (-B dv -+ sqrt(B^2 dv^2 - A (C dv^2 - t2))) / A.  Its right end
  // u_hi(v) is concave in v with its maximum at vr, so over a block row's v-range [v0, v1] the rightmost
  // point lies on the line v = clamp(vr, v0, v1) (likewise the leftmost on clamp(vl, v0, v1)): two line
  // evaluations per block row give the exact u-range of (ellipse n strip).  A line the ellipse does not
  // reach gives sqrt(negative) = NaN, which fminf / fmaxf replace by +-3e38: an empty range in row_bits.
  unsigned mask = 0u;
#pragma unroll
  for (int row = 0; row < 4; row++) {
    const float v0 = (float)(4 * row) - 7.5f, v1 = v0 + 3.f;
    const float dvh = fminf(fmaxf(vr, v0), v1) - gy, dvl = fminf(fmaxf(vl, v0), v1) - gy;
    const float hbh = B * dvh, hbl = B * dvl;
    const float rh = __builtin_amdgcn_sqrtf(fmaf(hbh, hbh, -A * fmaf(C * dvh, dvh, -t2)));
    const float rl = __builtin_amdgcn_sqrtf(fmaf(hbl, hbl, -A * fmaf(C * dvl, dvl, -t2)));
    const float hi = fmaf(rh - hbh, iA, gx), lo = fmaf(-hbl - rl, iA, gx);
    // NaN (row not reached) must give an empty range: fminf / fmaxf return the non-NaN operand
    mask |= row_bits(fminf(lo - M, 3.0e38f), fmaxf(hi + M, -3.0e38f)) << (4 * row);
  }
  return mask;
}

#ifndef TGS_BLK_U
#define TGS_BLK_U 2        // list entries (iterations) per index read; a quadrant's trip count is rounded up to it
#endif
constexpr int BLK_U = TGS_BLK_U;
typedef std::conditional<BLK_U == 4, unsigned int, std::conditional<BLK_U == 2, unsigned short, unsigned char>::type>::type BLK_T;

#ifndef TGS_BLK_WAVES
#define TGS_BLK_WAVES 4    // minimum waves per SIMD asked of the register allocator: 4 (89 VGPRs) and 6 (80) measure the same
#endif
// One 8x8 quadrant of a tile, one pixel per lane: the forward's form for tiles with LONG lists.  A tile's pixels are
// independent of each other, so the four quadrants of such a tile go to four blocks of the SAME launch (the tile's own
// block takes quadrant 0, three extra blocks at the end of the grid the others) with nothing to exchange: the wave that
// would walk ~1600 entries for 256 pixels walks the ~half of them that reach its quadrant, for 64, while the rest of the
// GPU -- idle in an object-centric frame, whose longest tile IS the launch -- hosts the other three.  Same
// staging, same blend_step on the same bits per pixel: images, final_T and stop positions bit for bit those of the other forms.
template <bool WANT_IDX>
__device__ __forceinline__ void raster_fwd_quadrant(
    const CamK& cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ sorted_gid,
    const int32_t* __restrict__ tile_start, float* __restrict__ out_rgb, float* __restrict__ out_depth,
    float* __restrict__ final_T, int32_t* __restrict__ final_idx, int32_t* __restrict__ stop_pos,
    int tile, int k, float4* __restrict__ recs) {
  const int lane = threadIdx.x;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  const PixConst pc = make_pix_const(lane);
  const float pu = (k & 1) ? pc.u[1] : pc.u[0], pv = (k >> 1) ? pc.v[1] : pc.v[0];
  const float puu = (k & 1) ? pc.uu[1] : pc.uu[0], pvv = (k >> 1) ? pc.vv[1] : pc.vv[0];
  const float puv = k == 0 ? pc.uv[0] : (k == 1 ? pc.uv[1] : (k == 2 ? pc.uv[2] : pc.uv[3]));
  const int px = tx * TGS_BLOCK + 8 * (k & 1) + (lane & 7);
  const int py = ty * TGS_BLOCK + 8 * (k >> 1) + (lane >> 3);
  const bool inb = px < cam.W && py < cam.H;
  float smax = inb ? LOG2_255 : -3.0e38f;
  float T = 1.f, Cr = 0.f, Cg = 0.f, Cb = 0.f, D = 0.f;
  int last = -1;
  const int start = tile_start[tile], end = tile_start[tile + 1];
  for (int base = start; base < end; base += 64) {
    if (__ballot(smax > 0.f) == 0ull) break;
    __syncthreads();
    unsigned my_mask = 0u;
    if (base + lane < end) {
      const float* r = splats + (size_t)sorted_gid[base + lane] * TGS_SPLAT_FLOATS;
      const float4 q0 = ld4(r), q1 = ld4(r + 4), q2 = ld4(r + 8);
      float gx, gy;
      centre_rel(q0, q2, tx, ty, cam.pix_center, gx, gy);
      const TileRec t = make_tile_rec(q0, q1, q2, gx, gy);
      recs[lane * 3] = t.a; recs[lane * 3 + 1] = t.b; recs[lane * 3 + 2] = t.c;
      my_mask = __float_as_uint(t.c.w);
    }
    __syncthreads();
    unsigned long long rem = __ballot((my_mask >> k) & 1u);
    while (rem) {
      const int j = __builtin_ctzll(rem);
      rem &= rem - 1;
      const float4 qa = recs[j * 3], qb = recs[j * 3 + 1], qc = recs[j * 3 + 2];
      float s = fmaf(qa.y, pu, qa.x);          // eval_s for slot k
      s = fmaf(qa.z, pv, s);
      s = fmaf(qa.w, puu, s);
      s = fmaf(qb.x, puv, s);
      s = fmaf(qb.y, pvv, s);
      float Tn; bool go;
      unsigned long long okb = 0ull;
      const float al = blend_step<true, 0>(s, T, smax, Tn, go, okb, stop_code(base - start + j));
      const float w = al * T;
      Cr = fmaf(w, qb.w, Cr); Cg = fmaf(w, qc.x, Cg);
      Cb = fmaf(w, qc.y, Cb); D = fmaf(w, qb.z, D);
      T = go ? Tn : T;
      if (WANT_IDX) last = go ? (base - start + j) : last;
    }
  }
  if (inb) {
    const size_t p = (size_t)py * cam.W + px;
    out_rgb[3 * p] = Cr + T * cam.bg[0];
    out_rgb[3 * p + 1] = Cg + T * cam.bg[1];
    out_rgb[3 * p + 2] = Cb + T * cam.bg[2];
    out_depth[p] = D;
    final_T[p] = T;
    if (WANT_IDX) final_idx[p] = last;
    if (stop_pos) stop_pos[p] = stop_of(smax);
  }
  // the quadrant's walk: the maximum is the tile's once all four have reported; a quarter each into the sum
  int wl = wave_minmax_i<true>(inb ? min(stop_of(smax), end - start) : 0);
  if (lane == 0 && wl > 0 && stop_pos) {   // (render-only forwards publish nothing, as in the unsplit forms)
    const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & (TGS_WALK_WORDS - 1));
    atomicMax(frame_scratch(tile_start) + TGS_WALK_AT(T_total, xcc), wl);
    atomicAdd(frame_scratch(tile_start) + TGS_WALKSUM_AT(T_total, xcc), (wl + 3) >> 2);
  }
}

// Which tiles the forward splits into quadrant blocks: lists longer than max(256, factor x I / 4096) -- factor (default 2)
// times what a wave slot would hold if the frame's I intersections were spread evenly -- among the first `heads` entries
// of the schedule (longest lists first: extra blocks are only launched for those).  cfg3 never qualifies.
struct SplitRule { int factor, n_slots, heads, floor; };
__device__ __forceinline__ bool tile_is_split(int n, int I, SplitRule r) {
  return r.factor > 0 && n > max(r.floor, (int)(((long long)I * r.factor) >> 12));
}

template <bool WANT_IDX>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TGS_BLK_WAVES, 8))) void k_raster_fwd_blocks(
    CamK cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ sorted_gid,
    const int32_t* __restrict__ tile_start, float* __restrict__ out_rgb,
    float* __restrict__ out_depth, float* __restrict__ final_T, int32_t* __restrict__ final_idx,
    const int32_t* __restrict__ tile_order, int32_t* __restrict__ stop_pos, SplitRule split) {
  __shared__ float4 recs[65 * 3];                 // 64 staged Gaussians + the null record (alpha = 0)
  int slot = blockIdx.x, part = -1;
  if (slot >= split.n_slots) {                    // extra blocks: quadrants 1..3 of the schedule's first `heads` tiles
    const int e = slot - split.n_slots;
    slot = e / 3; part = 1 + (e - 3 * slot);
  }
  const int tile = tile_order ? tile_order[slot] : xcd_tile(slot, T_total);
  if (tile >= T_total) return;
  if (split.factor > 0) {
    const bool sp = slot < split.heads && tile_is_split(tile_start[tile + 1] - tile_start[tile], tile_start[T_total], split);
    if (part >= 0 && !sp) return;
    if (sp) {
      raster_fwd_quadrant<WANT_IDX>(cam, T_total, splats, sorted_gid, tile_start, out_rgb, out_depth, final_T, final_idx,
                                    stop_pos, tile, part < 0 ? 0 : part, recs);
      return;
    }
  }
  const int lane = threadIdx.x;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  // lane -> pixel: DPP row g = lane >> 4 owns block (g & 1, g >> 1) of the quadrant, lane & 15 = pixel in it
  const int g = lane >> 4, lx = 4 * (g & 1) + (lane & 3), ly = 4 * (g >> 1) + ((lane >> 2) & 3);
  PixConst pc;
  pc.u[0] = (float)lx - 7.5f; pc.u[1] = pc.u[0] + 8.f;
  pc.v[0] = (float)ly - 7.5f; pc.v[1] = pc.v[0] + 8.f;
#pragma unroll
  for (int i = 0; i < 2; i++) { pc.uu[i] = pc.u[i] * pc.u[i]; pc.vv[i] = pc.v[i] * pc.v[i]; }
#pragma unroll
  for (int k = 0; k < 4; k++) pc.uv[k] = pc.u[k & 1] * pc.v[k >> 1];
  int pxi[4], pyi[4];
  float smax[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    pxi[k] = tx * TGS_BLOCK + 8 * (k & 1) + lx;
    pyi[k] = ty * TGS_BLOCK + 8 * (k >> 1) + ly;
    smax[k] = ((pxi[k] < cam.W) && (pyi[k] < cam.H)) ? LOG2_255 : -3.0e38f;
  }
  float T[4] = {1.f, 1.f, 1.f, 1.f};
  float Cr[4] = {0.f, 0.f, 0.f, 0.f}, Cg[4] = {0.f, 0.f, 0.f, 0.f}, Cb[4] = {0.f, 0.f, 0.f, 0.f};
  float D[4] = {0.f, 0.f, 0.f, 0.f};
  int last[4] = {-1, -1, -1, -1};

  __shared__ unsigned int lists4[16 * 16];        // 16 lists of 64 one-byte indices, padded with 64 = null
  unsigned char* lists = reinterpret_cast<unsigned char*>(lists4);
  if (lane == 0) {
    recs[64 * 3] = make_float4(3.0e38f, 0.f, 0.f, 0.f);
    recs[64 * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    recs[64 * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int start = tile_start[tile], end = tile_start[tile + 1];
  // my row's list in quadrant k: block (2 (k & 1) + (g & 1), 2 (k >> 1) + (g >> 1))
  int myblock[4];
#pragma unroll
  for (int k = 0; k < 4; k++) myblock[k] = 4 * (2 * (k >> 1) + (g >> 1)) + 2 * (k & 1) + (g & 1);

  unsigned live16 = 0xffffu;
  for (int base = start; base < end; base += 64) {
    // live blocks: bit 4 by + bx = some pixel of the block is still live.  A stale (larger) set only costs
    // iterations on stopped pixels, so the 16 block tests are refreshed every 4th batch; the exit test
    // (nothing live in the tile) runs on every batch.
    unsigned long long lv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) lv[k] = __ballot(smax[k] > 0.f);
    if ((lv[0] | lv[1] | lv[2] | lv[3]) == 0ull) break;
    if ((((base - start) >> 6) & 3) == 0) {
      live16 = 0u;
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          live16 |= ((lv[k] >> (16 * r)) & 0xffffull) ? (1u << (4 * (2 * (k >> 1) + (r >> 1)) + 2 * (k & 1) + (r & 1))) : 0u;
    }
    __syncthreads();
    unsigned my_mask = 0u;
    float my_opac = 0.f;
    if (base + lane < end) {
      const float* r = splats + (size_t)sorted_gid[base + lane] * TGS_SPLAT_FLOATS;
      const float4 q0 = ld4(r), q1 = ld4(r + 4), q2 = ld4(r + 8);
      float gx, gy;
      centre_rel(q0, q2, tx, ty, cam.pix_center, gx, gy);
      const TileRec t = make_tile_rec<false>(q0, q1, q2, gx, gy);
      recs[lane * 3] = t.a; recs[lane * 3 + 1] = t.b; recs[lane * 3 + 2] = t.c;
      my_mask = block_mask16(gx, gy, q1.x, q1.y, q1.z, -__log2f(q0.w)) & live16;
      my_opac = q0.w;
    }
    const bool clampy = __ballot(my_opac > CLAMP_FREE_OPACITY) != 0ull;
    reinterpret_cast<uint4*>(lists4)[lane] = make_uint4(0x40404040u, 0x40404040u, 0x40404040u, 0x40404040u);
    __syncthreads();
    int cnt[16];
#pragma unroll
    for (int b = 0; b < 16; b++) {
      const bool in = (my_mask >> b) & 1u;
      const unsigned long long bal = __ballot(in);
      cnt[b] = __popcll(bal);
      if (in) lists[b * 64 + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (unsigned char)lane;
    }
    __syncthreads();
    auto walk = [&](auto mayclamp) {
      constexpr bool MAYCLAMP = decltype(mayclamp)::value;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int b0 = 8 * (k >> 1) + 2 * (k & 1);                 // blocks b0, b0 + 1, b0 + 4, b0 + 5
        const int n = max(max(cnt[b0], cnt[b0 + 1]), max(cnt[b0 + 4], cnt[b0 + 5]));
        const BLK_T* mylist = reinterpret_cast<const BLK_T*>(lists4) + myblock[k] * (64 / BLK_U);
        for (int i4 = 0; BLK_U * i4 < n; i4++) {
          const unsigned idx4 = mylist[i4];
#pragma unroll
          for (int e = 0; e < BLK_U; e++) {
            const int j = (idx4 >> (8 * e)) & 0xffu;               // 64 = null record: s = 3e38, alpha = 0
            const float4 qa = recs[j * 3], qb = recs[j * 3 + 1], qc = recs[j * 3 + 2];
            const float s = eval_s(qa, qb, pc, k);
            float Tn; bool go;
            unsigned long long okb = 0ull;
            const float al = blend_step<MAYCLAMP, 0>(s, T[k], smax[k], Tn, go, okb, batch_stop_code(j));
            const float w = al * T[k];
            Cr[k] = fmaf(w, qb.w, Cr[k]); Cg[k] = fmaf(w, qc.x, Cg[k]);
            Cb[k] = fmaf(w, qc.y, Cb[k]); D[k] = fmaf(w, qb.z, D[k]);
            T[k] = go ? Tn : T[k];          // (fmaf(-al, T, T) instead of the select: no difference, same box)
            if (WANT_IDX) last[k] = go ? (base - start + j) : last[k];
          }
        }
      }
    };
    // batches without an opacity above CLAMP_FREE_OPACITY (nearly all) run the copy of the loops without the
    // v_min of the 0.999 clamp (identity there: bit-identical) -- a per-BATCH choice, no branch inside the loops
    if (clampy) walk(std::true_type{}); else walk(std::false_type{});   // 144 -> 139 us at cfg3 (same box)
    // pixels that stopped in this batch: batch-relative code -> list position (4 selects per batch, not per entry)
#pragma unroll
    for (int k = 0; k < 4; k++) smax[k] = finish_batch_stop(smax[k], base - start);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (pxi[k] < cam.W && pyi[k] < cam.H) {
      const size_t p = (size_t)pyi[k] * cam.W + pxi[k];
      out_rgb[3 * p] = Cr[k] + T[k] * cam.bg[0];
      out_rgb[3 * p + 1] = Cg[k] + T[k] * cam.bg[1];
      out_rgb[3 * p + 2] = Cb[k] + T[k] * cam.bg[2];
      out_depth[p] = D[k];
      final_T[p] = T[k];
      if (WANT_IDX) final_idx[p] = last[k];
      if (stop_pos) stop_pos[p] = stop_of(smax[k]);
    }
  }
  // the walk statistics are K7's (chain-bound test): a render-only forward (no stop_pos) publishes nothing -- and
  // does not disturb the sums a training forward on the same lists has left (they accumulate: one publishing forward
  // per backward; 1.5 us of K6 at cfg3)
  if (stop_pos) publish_walk(tile_start, T_total, smax, end - start, lane);
}

// ---------------------------------------------------------------------------------------------
// K7 backward
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t pair_index(const int32_t* __restrict__ group_base, int gid,
                                             float4 r2, int tx, int ty) {
  int x0, y0, w, h;
  unpack_rect(__float_as_uint(r2.z), x0, y0, w, h);
  return (size_t)group_base[gid / TGS_GROUP] + (size_t)__float_as_int(r2.w) +
         (size_t)((ty - y0) * w + (tx - x0));
}

__global__ __launch_bounds__(64) void k_raster_bwd_f2b(
    CamK cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const int32_t* __restrict__ sorted_gid, const int32_t* __restrict__ tile_start,
    const float* __restrict__ out_rgb, const float* __restrict__ out_depth,
    const float* __restrict__ final_T, const float* __restrict__ v_rgb,
    const float* __restrict__ v_depth, const float* __restrict__ v_alpha, LossK loss,
    float* __restrict__ partials, float* __restrict__ tile_loss,
    const int32_t* __restrict__ tile_order, const unsigned long long* __restrict__ slot_ok) {
  const int tile = tile_order ? tile_order[blockIdx.x] : xcd_tile(blockIdx.x, T_total);
  if (tile >= T_total) return;
  const int lane = threadIdx.x;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  const PixConst pc = make_pix_const(lane);
  const int start = tile_start[tile], end = tile_start[tile + 1];
  const int n = end - start;

  // The list is walked FRONT TO BACK exactly like the forward (same blend_step, hence the same
  // transmittance sequence and the same threshold decisions); the colour accumulated BEHIND a
  // Gaussian, which B.7 needs, is (final colour - prefix), both projected on the pixel's upstream
  // gradient:  S.v = Cv_total - Pv.   Per-pixel state: T, XP = X + Pv, smax, with
  // X = T_final*(v_A - bg.v_C) - Cv_total  (the running sum starts at X: one add less per slot).
  float T[4] = {1.f, 1.f, 1.f, 1.f};
  float smax[4], vCr[4], vCg[4], vCb[4], vD[4], XP[4];
  float l_l1 = 0.f, l_dep = 0.f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int px = tx * TGS_BLOCK + 8 * (k & 1) + (lane & 7);
    const int py = ty * TGS_BLOCK + 8 * (k >> 1) + (lane >> 3);
    smax[k] = -3.0e38f; vCr[k] = vCg[k] = vCb[k] = vD[k] = XP[k] = 0.f;
    if (px < cam.W && py < cam.H) {
      smax[k] = LOG2_255;
      const size_t p = (size_t)py * cam.W + px;
      const float Tf = final_T[p];
      const float o0 = out_rgb[3 * p], o1 = out_rgb[3 * p + 1], o2 = out_rgb[3 * p + 2];
      const float od = out_depth[p];
      float vA = v_alpha ? v_alpha[p] : 0.f;
      if (v_rgb) { vCr[k] = v_rgb[3 * p]; vCg[k] = v_rgb[3 * p + 1]; vCb[k] = v_rgb[3 * p + 2]; }
      if (v_depth) vD[k] = v_depth[p];
      if (loss.on) {
        if (loss.gt_rgb) {
          const float d0 = o0 - loss.gt_rgb[3 * p];
          const float d1 = o1 - loss.gt_rgb[3 * p + 1];
          const float d2 = o2 - loss.gt_rgb[3 * p + 2];
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
            const float dhat = od * ia;
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
      // colour part of the output without the background term, projected on v
      const float cvtot = vCr[k] * (o0 - Tf * cam.bg[0]) + vCg[k] * (o1 - Tf * cam.bg[1]) +
                          vCb[k] * (o2 - Tf * cam.bg[2]) + vD[k] * od;
      XP[k] = Tf * (vA - bgdot) - cvtot;
    }
  }
  if (tile_loss) {
    const float a = wave_sum(l_l1), b = wave_sum(l_dep);
    if (lane == 0) { tile_loss[2 * tile] = a; tile_loss[2 * tile + 1] = b; }
  }
  if (n == 0) return;

  __shared__ float4 recs[64 * 3];
  __shared__ float4 sums[64 * 4];  // [Gaussian j][16 slots], slot c < 14 = total of term c (table below)
  // Transposed reduction scratch: row r holds the 64 lanes' values of per-lane accumulator r
  //   r = 0..2 v_rgb, 3 v_depth, 4 S = sum q, 5 Sx = sum q over the lane's right-hand pixels (slots 1, 3),
  //   6 Sy = sum q over its lower pixels (slots 2, 3), 7 Quv = sum q u v.
  // Reader lane (part, c) takes 16 consecutive lane values of row ROW[c] (writer lanes 16 part .. +15) and
  // forms a WEIGHTED sum: the pixel-coordinate moments the conversion to (v_xy, v_conic) needs are linear
  // in S, Sx, Sy with coefficients that depend on the writer lane only (u = u0 + 8 [right], v = v0 + 8
  // [lower], u0 = (l & 7) - 7.5, v0 = (l >> 3) - 7.5), so the blend loop accumulates 3 instead of 6
  // values per pixel slot and the moments appear here at the price of 16 FMAs instead of 15 adds:
  //   c  0..3  row 0..3, w = 1                          v_rgb, v_depth
  //   c  4     S, 1            -> Q0
  //   c  5, 6  S, u0 | Sx, 8   -> Qu  (sum of the two)
  //   c  7, 8  S, v0 | Sy, 8   -> Qv
  //   c  9,10  S, u0^2 | Sx, 16 u0 + 64   -> Quu
  //   c 11,12  S, v0^2 | Sy, 16 v0 + 64   -> Qvv
  //   c 13     Quv, 1
  // Bank conflicts: ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32)
  // (MI355X_MICROARCH.md), NOT in quarter waves.  part = (lane >> 4) ^ [4 <= c < 12] makes every such group
  // read ONE part of <= 8 different rows, whose 68-float stride puts them on disjoint bank quads; lanes
  // l, l^16, l^32, l^48 still hold the four parts of one c, which is all the two folds need.
  constexpr int RED_RS = 68;
  constexpr int RED_ROWS = 8;
  __shared__ float4 red4[RED_ROWS * RED_RS / 4];
  float* red = reinterpret_cast<float*>(red4);
  const int red_c = lane & 15;
  const int red_part = (lane >> 4) ^ ((red_c >= 4 && red_c < 12) ? 1 : 0);
  float wt[16];
  const float* red_rd;
  {
    // row of term c, packed 4 bits each: c = 0 .. 15 -> 0 1 2 3 4 4 5 4 6 4 5 4 6 7 0 0
    const unsigned long long ROWS = 0x0076454645443210ull;
    red_rd = red + (int)((ROWS >> (4 * red_c)) & 15ull) * RED_RS + red_part * 16;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const float u0 = (float)(i & 7) - 7.5f;
      const float v0 = (float)(2 * red_part + (i >> 3)) - 7.5f;
      float w = 1.f;                                   // c 0..4, 13
      w = (red_c == 5) ? u0 : w;
      w = (red_c == 7) ? v0 : w;
      w = (red_c == 6 || red_c == 8) ? 8.f : w;
      w = (red_c == 9) ? u0 * u0 : w;
      w = (red_c == 11) ? v0 * v0 : w;
      w = (red_c == 10) ? 16.f * u0 + 64.f : w;
      w = (red_c == 12) ? 16.f * v0 + 64.f : w;
      w = (red_c >= 14) ? 0.f : w;
      wt[i] = w;
    }
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int base = start;
  for (; base < end; base += 64) {
    unsigned slot_live = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) slot_live |= (__ballot(smax[k] > 0.f) != 0ull) ? (1u << k) : 0u;
    if (slot_live == 0u) break;
    const int cnt = min(64, end - base);
    size_t P = 0;
    float4 a0 = z4, a1 = z4;
    unsigned my_mask = 0u;
    __syncthreads();
    if (lane < cnt) {
      const int gid = sorted_gid[base + lane];
      const float* r = splats + (size_t)gid * TGS_SPLAT_FLOATS;
      a0 = ld4(r); a1 = ld4(r + 4);
      const float4 a2 = ld4(r + 8);
      float gx, gy;
      centre_rel(a0, a2, tx, ty, cam.pix_center, gx, gy);
      a0.x = gx; a0.y = gy;                       // kept for the conversion of the moments below
      const TileRec t = make_tile_rec(a0, a1, a2, gx, gy);
      recs[lane * 3] = t.a; recs[lane * 3 + 1] = t.b; recs[lane * 3 + 2] = t.c;
      my_mask = __float_as_uint(t.c.w);
      P = pair_index(group_base, gid, a2, tx, ty);
    }
    sums[lane * 4] = z4; sums[lane * 4 + 1] = z4; sums[lane * 4 + 2] = z4; sums[lane * 4 + 3] = z4;
    __syncthreads();

    unsigned long long qm[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
      qm[k] = ((slot_live >> k) & 1u) ? __ballot((my_mask >> k) & 1u) : 0ull;
    if (slot_ok) {   // quadrants in which the forward saw no live pixel reach alpha >= 1/255: nothing to do
      const unsigned long long* o = slot_ok + 4 * slot_ok_index(base, tile);
#pragma unroll
      for (int k = 0; k < 4; k++) qm[k] &= __builtin_nontemporal_load(o + k);
    }
    auto walk = [&](auto mayclamp) {
    constexpr bool MAYCLAMP = decltype(mayclamp)::value;
    unsigned long long rem = qm[0] | qm[1] | qm[2] | qm[3];
    while (rem) {
      const int j = __builtin_ctzll(rem);
      rem &= rem - 1;
      const unsigned m = (unsigned)((qm[0] >> j) & 1ull) | ((unsigned)((qm[1] >> j) & 1ull) << 1) |
                         ((unsigned)((qm[2] >> j) & 1ull) << 2) | ((unsigned)((qm[3] >> j) & 1ull) << 3);
      const float4 qa = recs[j * 3], qb = recs[j * 3 + 1], qc = recs[j * 3 + 2];
      // acc: 0..2 v_rgb, 3 v_depth, 4 S = sum q, 5 Sx, 6 Sy, 7 Quv   (rows of the reduction scratch)
      float acc[RED_ROWS];
#pragma unroll
      for (int c = 0; c < RED_ROWS; c++) acc[c] = 0.f;
      // lanes with a contributing pixel, as a SCALAR mask: the ballots inside blend_step are the compares' own
      // results, so keeping `any` as a mask costs one s_or per quadrant -- as a per-lane bool it cost three scalar
      // instructions per quadrant and a v_cndmask + v_cmp per Gaussian for the final ballot (K7 396 -> 383 us)
      unsigned long long any = 0ull;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (m & (1u << k)) {  // wave-uniform
          const float s = eval_s(qa, qb, pc, k);
          float Tn; bool go;
          unsigned long long okb;
          const float al = blend_step<MAYCLAMP, 2>(s, T[k], smax[k], Tn, go, okb);   // 0 unless this Gaussian contributes
          const float w = al * T[k];
          // alpha / (1 - alpha) = alpha T / T'  (T' = T (1 - alpha) is already there; w = 0 if skipped)
          const float kap = w * __builtin_amdgcn_rcpf(Tn);
          acc[0] = fmaf(w, vCr[k], acc[0]); acc[1] = fmaf(w, vCg[k], acc[1]);
          acc[2] = fmaf(w, vCb[k], acc[2]); acc[3] = fmaf(w, vD[k], acc[3]);
          float cv = qb.w * vCr[k];
          cv = fmaf(qc.x, vCg[k], cv); cv = fmaf(qc.y, vCb[k], cv); cv = fmaf(qb.z, vD[k], cv);
          // q = opacity e^-sigma dL/dalpha, dL/dalpha = T cv + (X + Pv)/(1 - alpha)   (B.7).  With
          // z = w cv:  q = z + alpha/(1-alpha) (X + Pv + z)  -- exact when alpha is not clamped;
          // under the 0.999 clamp (opacity > 0.999 and sigma ~ 0) q is rescaled by e^-s / 0.999.
          const float z = w * cv;
          XP[k] += z;
          float q = fmaf(kap, XP[k], z);               // = 0 automatically when al == 0
          if constexpr (MAYCLAMP) q *= fmaxf(__builtin_amdgcn_exp2f(-s) * (1.0f / ALPHA_MAX), 1.0f);
          T[k] = fmaf(-al, T[k], T[k]);                // unchanged when al == 0
          acc[4] += q;
          if (k & 1) acc[5] += q;
          if (k >> 1) acc[6] += q;
          acc[7] = fmaf(q, pc.uv[k], acc[7]);
          any |= okb;
        }
      }
      if (any != 0ull) {
        // LDS operations of one wave execute in order, so the (single-wave) workgroup needs no
        // barrier between the scatter and the transposed read -- only the compiler must keep them
        // in order, which the possible aliasing already forces
#pragma unroll
        for (int c = 0; c < RED_ROWS; c++) red[c * RED_RS + lane] = acc[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4 r0 = ld4(red_rd), r1 = ld4(red_rd + 4), r2 = ld4(red_rd + 8), r3 = ld4(red_rd + 12);
        // four independent chains of four
        float o0 = r0.x * wt[0], o1 = r1.x * wt[4], o2 = r2.x * wt[8], o3 = r3.x * wt[12];
        o0 = fmaf(r0.y, wt[1], o0); o1 = fmaf(r1.y, wt[5], o1); o2 = fmaf(r2.y, wt[9], o2); o3 = fmaf(r3.y, wt[13], o3);
        o0 = fmaf(r0.z, wt[2], o0); o1 = fmaf(r1.z, wt[6], o1); o2 = fmaf(r2.z, wt[10], o2); o3 = fmaf(r3.z, wt[14], o3);
        o0 = fmaf(r0.w, wt[3], o0); o1 = fmaf(r1.w, wt[7], o1); o2 = fmaf(r2.w, wt[11], o2); o3 = fmaf(r3.w, wt[15], o3);
        float O = (o0 + o1) + (o2 + o3);
        O = fold_xor16(O);
        O = fold_xor32(O);
        if (lane < 16) reinterpret_cast<float*>(sums)[j * 16 + lane] = O;
        __builtin_amdgcn_wave_barrier();
      }
    }
    };
    // The 0.999 clamp can only bind for opacity > 0.999; batches without a Gaussian above 0.99 (the
    // margin covers rounding, see CLAMP_FREE_OPACITY; almost all of a typical scene) run a copy of the loop without the clamp (min) and its pass-through correction
    // (3 VALU) -- a per-BATCH choice, so the Gaussian loop itself has no extra branch.
    if (__ballot(lane < cnt && a0.w > CLAMP_FREE_OPACITY) != 0ull) walk(std::true_type{});
    else walk(std::false_type{});
    __syncthreads();
    if (lane < cnt) {
      const float4 s0 = sums[lane * 4], s1 = sums[lane * 4 + 1], s2 = sums[lane * 4 + 2], s3 = sums[lane * 4 + 3];
      // s0 = {v_r, v_g, v_b, v_depth}  s1 = {Q0, Qu', Qu", Qv'}  s2 = {Qv", Quu', Quu", Qvv'}  s3 = {Qvv", Quv, -, -}
      const float gx = a0.x, gy = a0.y;           // centre relative to the tile centre (set while staging)
      const float A = a1.x, B = a1.y, Cc = a1.z;
      const float Q0 = s1.x, Qu = s1.y + s1.z, Qv = s1.w + s2.x, Quu = s2.y + s2.z, Qvv = s2.w + s3.x, Quv = s3.y;
      // v_sigma = -q, Delta = (gx - u, gy - v):  M* = sum v_sigma * Delta-monomials
      const float Mx = -(gx * Q0 - Qu), My = -(gy * Q0 - Qv);
      const float Mxx = -(gx * gx * Q0 - 2.f * gx * Qu + Quu);
      const float Mxy = -(gx * gy * Q0 - gx * Qv - gy * Qu + Quv);
      const float Myy = -(gy * gy * Q0 - 2.f * gy * Qv + Qvv);
      float* o = partials + P * TGS_PARTIAL_FLOATS;
      // {v_x, v_y, v_depth, v_opacity, v_a, v_b, v_c, v_r, v_g, v_b}
      st4(o, make_float4(A * Mx + B * My, B * Mx + Cc * My, s0.w, Q0 / a0.w));
      st4(o + 4, make_float4(0.5f * Mxx, Mxy, 0.5f * Myy, s0.x));
      st4(o + 8, make_float4(s0.y, s0.z, 0.f, 0.f));
    }
  }
  // every pixel of the tile has stopped: the rest of the list received no gradient
  for (int i = base + lane; i < end; i += 64) {
    const int gid = sorted_gid[i];
    const size_t P = pair_index(group_base, gid, ld4(splats + (size_t)gid * TGS_SPLAT_FLOATS + 8), tx, ty);
    float* o = partials + P * TGS_PARTIAL_FLOATS;
    st4(o, z4); st4(o + 4, z4); st4(o + 8, z4);
  }
}


// Which tiles go to the four-wave form of K7 (k_raster_bwd_quad).  Two launches in one stream run one after the other,
// each as long as its longest chain, so the split is decided per FRAME: a frame is chain-bound if its deepest walk
// (publish_walk: K6 leaves it behind the tile starts) exceeds factor / 2 times the load a wave slot would get if the
// frame's walks (their sum, next to it) were spread evenly over the 4096 slots K7 has (256 CUs x 4 SIMDs x 4 waves).  Then every tile that walks more than min_walk entries is the
// four-wave kernel's and the one-wave kernel only keeps the trivial rest; otherwise the four-wave launch returns at
// once.  Measured per view (tools/k7_quad_probe.py, four waves / one wave): 0.58 - 0.69 at 100 k object-centric
// Gaussians / 720p (ratios 5.6 - 12), 0.68 - 0.78 at 300 k (4.5 - 16); 0.97 - 1.30 at 1 M clustered / 1080p (1.7 - 7.9:
// the four-wave form does 1.5x the work) and 1.4 - 1.6 on the uniform cfg3 (0.6 - 1.3).  The default factor 8 (ratio > 4)
// takes the first two and leaves the others alone.  Both kernels evaluate the same predicate on the same words.
// factor 0: never.
struct QuadRule { int factor, min_walk, scan_min, scan_heads; };   // scan_min > 0: tiles of the schedule's first scan_heads slots that walk more than scan_min entries are k_raster_bwd_scan's
__device__ __forceinline__ bool frame_is_chain_bound(const int32_t* __restrict__ tile_start, int T, QuadRule q) {
  if (q.factor <= 0) return false;
  int walk = 0;
  long long work = 0;
#pragma unroll
  for (int i = 0; i < TGS_WALK_WORDS; i++) {
    walk = max(walk, tile_start[TGS_WALK_AT(T, i)]);
    work += tile_start[TGS_WALKSUM_AT(T, i)];
  }
  return (long long)walk * 8192 > work * q.factor;
}

// ---------------------------------------------------------------------------------------------
// K7 backward, back to front (round 4; the default)
// ---------------------------------------------------------------------------------------------
// k_raster_bwd_f2b above walks the list front to back and obtains the colour BEHIND a Gaussian as
// (final colour - prefix).  That difference carries an ABSOLUTE error of one ulp of the pixel's colour
// (~1e-7) whatever is left behind, while the quantity it feeds -- dL/dalpha of a Gaussian at transmittance
// T -- is itself O(T): the relative error of the geometry / opacity gradients of an occluded Gaussian is
// ~1e-7 / T, 2e-4 in the median at cfg3 (lists of ~400, most Gaussians sit behind T ~ 1e-3) against 1e-5 for
// the back-to-front order every CUDA rasterizer uses (tools/grad_err_k7.py, profiles/r4_grad_err_*.txt).
// This kernel walks back to front:
//   * the forward leaves, per pixel, the list position of the Gaussian that stopped it (stop_pos; free: the
//     position rides in the bits of the dead pixel's smax) -- positions below it with alpha >= 1/255
//     contributed, nothing else did.  The alpha test is the forward's own compare on the same bits
//     (eval_s on the same record), so the decisions are the forward's by construction; the T' <= 1e-4 test
//     is not re-evaluated at all.
//   * T before a Gaussian = T after it * rcp(1 - alpha) (relative error ~1 ulp per step); the sum behind it,
//     SX = sum_{j behind} w_j c_j.v - T_final (v_A - bg.v_C), accumulates small terms first.
//     dL/dalpha = T c.v - SX / (1 - alpha).
//   * per slot: 28 VALU against ~30 (no T_stop compare, no smax select; one more v_cmp for the position).
//   * everything else -- staging, quadrant masks, the weighted transposed LDS reduction, the partial record
//     -- is k_raster_bwd_f2b's.  Batches behind the last pixel's stop position are never staged; their
//     partial records are zero-filled up front.
// (5 waves per SIMD -- amdgpu_waves_per_eu(5, 8): 96 VGPRs + 14 spilled -- 386 -> 508 us, same-box A/B: 112 VGPRs / 4 waves stay)
#ifndef TGS_K7_FIRSTQ
#define TGS_K7_FIRSTQ 0    // 1: the first quadrant a Gaussian reaches assigns its eight lane sums instead of clearing + accumulating (A/B switch: slower)
#endif
#ifndef TGS_K7_RECPF
#define TGS_K7_RECPF 0     // 1: the next entry's staged record is read from LDS while the current entry is blended (A/B switch)
#endif
#ifndef TGS_K7_PAIR
#define TGS_K7_PAIR 0      // 1: two Gaussians per reduction round (A/B switch; tools/build_variant.sh k7pair -DTGS_K7_PAIR=1)
#endif
#if TGS_K7_RECPF
__attribute__((amdgpu_waves_per_eu(4, 8)))
#endif
__global__ __launch_bounds__(64) void k_raster_bwd(
    CamK cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const int32_t* __restrict__ sorted_gid, const int32_t* __restrict__ tile_start,
    const float* __restrict__ out_rgb, const float* __restrict__ out_depth,
    const float* __restrict__ final_T, const int32_t* __restrict__ stop_pos, const float* __restrict__ v_rgb,
    const float* __restrict__ v_depth, const float* __restrict__ v_alpha, LossK loss,
    float* __restrict__ partials, float* __restrict__ tile_loss,
    const int32_t* __restrict__ tile_order, QuadRule quad) {
  // the slot counters of the four-wave launch that follows in the stream (8 words behind the walk words): cleared by
  // block 0 BEFORE any early return -- block 0's tile is empty in most object-centric frames
  if (blockIdx.x == 0 && threadIdx.x < TGS_WALK_WORDS && quad.factor > 0)
    frame_scratch(tile_start)[TGS_SLOTCTR_AT(T_total, threadIdx.x)] = 0;
  const int tile = tile_order ? tile_order[blockIdx.x] : xcd_tile(blockIdx.x, T_total);
  if (tile >= T_total) return;
  const int lane = threadIdx.x;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  const PixConst pc = make_pix_const(lane);
  const int start = tile_start[tile], end = tile_start[tile + 1];
  const int n = end - start;

  // per-pixel state: T (after the Gaussians walked so far, i.e. behind the current one), SX, and lim = number
  // of leading list positions that may contribute
  float T[4] = {1.f, 1.f, 1.f, 1.f};
  float vCr[4], vCg[4], vCb[4], vD[4], SX[4];
  int lim[4];
  float l_l1 = 0.f, l_dep = 0.f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int px = tx * TGS_BLOCK + 8 * (k & 1) + (lane & 7);
    const int py = ty * TGS_BLOCK + 8 * (k >> 1) + (lane >> 3);
    lim[k] = 0; vCr[k] = vCg[k] = vCb[k] = vD[k] = SX[k] = 0.f;
    if (px < cam.W && py < cam.H) {
      const size_t p = (size_t)py * cam.W + px;
      const float Tf = final_T[p];
      T[k] = Tf;
      lim[k] = min(stop_pos[p], n);
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
      SX[k] = -Tf * (vA - bgdot);
    }
  }
  if (tile_loss) {
    const float a = wave_sum(l_l1), b = wave_sum(l_dep);
    if (lane == 0) { tile_loss[2 * tile] = a; tile_loss[2 * tile + 1] = b; }
  }
  if (n == 0) return;
  // per quadrant / per tile: how far into the list any pixel reaches (wave-uniform)
  int qlim[4];
#pragma unroll
  for (int k = 0; k < 4; k++) qlim[k] = wave_minmax_i<true>(lim[k]);
  const int tmax = max(max(qlim[0], qlim[1]), max(qlim[2], qlim[3]));
  // in a chain-bound frame every tile with a walk worth splitting is k_raster_bwd_quad's (four waves); its losses were
  // written above
  if (tmax > quad.min_walk && frame_is_chain_bound(tile_start, T_total, quad)) return;   // (k_raster_bwd_quad's or k_raster_bwd_scan's)
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // behind the last stop position nothing received a gradient
  for (int i = start + tmax + lane; i < end; i += 64) {
    const int gid = sorted_gid[i];
    const size_t P = pair_index(group_base, gid, ld4(splats + (size_t)gid * TGS_SPLAT_FLOATS + 8), tx, ty);
    float* o = partials + P * TGS_PARTIAL_FLOATS;
    st4(o, z4); st4(o + 4, z4); st4(o + 8, z4);
  }
  if (tmax == 0) return;

  __shared__ float4 recs[64 * 3];
  __shared__ float4 sums[64 * 4];  // [Gaussian j][16 slots], slot c < 14 = total of term c (table in k_raster_bwd_f2b)
  constexpr int RED_RS = 68;
  constexpr int RED_ROWS = 8;
  __shared__ float4 red4[(TGS_K7_PAIR ? 2 : 1) * RED_ROWS * RED_RS / 4];
  float* red = reinterpret_cast<float*>(red4);
  const int red_c = lane & 15;
  const int red_part = (lane >> 4) ^ ((red_c >= 4 && red_c < 12) ? 1 : 0);
  float wt[16];
  const float* red_rd;
  {
    const unsigned long long ROWS = 0x0076454645443210ull;
    red_rd = red + (int)((ROWS >> (4 * red_c)) & 15ull) * RED_RS + red_part * 16;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const float u0 = (float)(i & 7) - 7.5f;
      const float v0 = (float)(2 * red_part + (i >> 3)) - 7.5f;
      float w = 1.f;                                   // c 0..4, 13
      w = (red_c == 5) ? u0 : w;
      w = (red_c == 7) ? v0 : w;
      w = (red_c == 6 || red_c == 8) ? 8.f : w;
      w = (red_c == 9) ? u0 * u0 : w;
      w = (red_c == 11) ? v0 * v0 : w;
      w = (red_c == 10) ? 16.f * u0 + 64.f : w;
      w = (red_c == 12) ? 16.f * v0 + 64.f : w;
      w = (red_c >= 14) ? 0.f : w;
      wt[i] = w;
    }
  }
  for (int base = start + ((tmax - 1) & ~63); base >= start; base -= 64) {
    const int rel = base - start;
    const int cnt = min(64, tmax - rel);
    size_t P = 0;
    float4 a0 = z4, a1 = z4;
    unsigned my_mask = 0u;
    __syncthreads();
    if (lane < cnt) {
      const int gid = sorted_gid[base + lane];
      const float* r = splats + (size_t)gid * TGS_SPLAT_FLOATS;
      a0 = ld4(r); a1 = ld4(r + 4);
      const float4 a2 = ld4(r + 8);
      float gx, gy;
      centre_rel(a0, a2, tx, ty, cam.pix_center, gx, gy);
      a0.x = gx; a0.y = gy;                       // kept for the conversion of the moments below
      const TileRec t = make_tile_rec(a0, a1, a2, gx, gy);
      my_mask = __float_as_uint(t.c.w);
      recs[lane * 3] = t.a; recs[lane * 3 + 1] = t.b; recs[lane * 3 + 2] = t.c;
      P = pair_index(group_base, gid, a2, tx, ty);
    }
    sums[lane * 4] = z4; sums[lane * 4 + 1] = z4; sums[lane * 4 + 2] = z4; sums[lane * 4 + 3] = z4;
    __syncthreads();

    // quadrant k only needs the batch entries below its own furthest stop position
    unsigned long long qm[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int r = qlim[k] - rel;
      const unsigned long long keep = r >= 64 ? ~0ull : (r <= 0 ? 0ull : ((1ull << r) - 1ull));
      qm[k] = __ballot((my_mask >> k) & 1u) & keep;
    }
    const unsigned long long rem0 = qm[0] | qm[1] | qm[2] | qm[3];
    auto walk = [&](auto mayclamp) {
    constexpr bool MAYCLAMP = decltype(mayclamp)::value;
    unsigned long long rem = rem0;
    // one Gaussian against the (up to four) quadrants it reaches: the lane's 8 partial sums + "some pixel took it"
    auto blend = [&](int j, const float4 qa, const float4 qb, const float4 qc, float (&acc)[RED_ROWS], unsigned long long& any) {
      const unsigned m = (unsigned)((qm[0] >> j) & 1ull) | ((unsigned)((qm[1] >> j) & 1ull) << 1) |
                         ((unsigned)((qm[2] >> j) & 1ull) << 2) | ((unsigned)((qm[3] >> j) & 1ull) << 3);
      const int pos = rel + j;
      // acc: 0..2 v_rgb, 3 v_depth, 4 S = sum q, 5 Sx, 6 Sy, 7 Quv   (rows of the reduction scratch)
      any = 0ull;   // lanes with a contributing pixel, kept as a scalar mask (see k_raster_bwd_f2b)
      // One quadrant of the Gaussian.  FIRST (TGS_K7_FIRSTQ, an experiment that LOST): the first quadrant a Gaussian
      // reaches ASSIGNS the eight sums (w v instead of fma(w, v, 0): the same bits), which removes the eight v_mov that
      // clear them -- 6 of ~86 VALU instructions per pair -- at the price of a four-way scalar switch and ten copies of
      // this body instead of four: 383 -> 419 us (profiles/r5_ab_runs.txt).  Scalar branches are not free here.
      auto quad = [&](auto KC, auto FC) {
        constexpr int k = decltype(KC)::value;
        constexpr bool FIRST = decltype(FC)::value;
        const float s = eval_s(qa, qb, pc, k);
        const float e = __builtin_amdgcn_exp2f(-s);
        const float al0 = MAYCLAMP ? fminf(ALPHA_MAX, e) : e;
        // the forward's alpha test (same bits) below the pixel's stop position: exactly the contributions
        const bool ok = s <= LOG2_255, in = pos < lim[k];
        const bool go = ok & in;
        // each ballot sits next to its compare, so it IS the compare's scalar result (a ballot of `go` costs
        // a v_cndmask + v_cmp per slot)
        const unsigned long long gob = __builtin_amdgcn_ballot_w64(ok) & __builtin_amdgcn_ballot_w64(in);
        const float al = go ? al0 : 0.f;
        const float ra = __builtin_amdgcn_rcpf(1.0f - al);    // = 1 exactly when al == 0
        const float Tp = T[k] * ra;                           // transmittance in front of this Gaussian
        const float w = al * Tp;
        float cv = qb.w * vCr[k];
        cv = fmaf(qc.x, vCg[k], cv); cv = fmaf(qc.y, vCb[k], cv); cv = fmaf(qb.z, vD[k], cv);
        // q = opacity e^-sigma dL/dalpha = alpha (T c.v - SX / (1 - alpha))   (B.7); = 0 when al == 0.
        // Under the 0.999 clamp (opacity > 0.999 and sigma ~ 0) q is rescaled by e^-s / 0.999 (pass-through).
        const float z = w * cv;
        float q = fmaf(-(al * ra), SX[k], z);
        if constexpr (MAYCLAMP) q *= fmaxf(e * (1.0f / ALPHA_MAX), 1.0f);
        SX[k] += z;
        T[k] = Tp;
        if constexpr (FIRST) {
          acc[0] = w * vCr[k]; acc[1] = w * vCg[k]; acc[2] = w * vCb[k]; acc[3] = w * vD[k];
          acc[4] = q;
          acc[5] = (k & 1) ? q : 0.f;
          acc[6] = (k >> 1) ? q : 0.f;
          acc[7] = q * pc.uv[k];
        } else {
          acc[0] = fmaf(w, vCr[k], acc[0]); acc[1] = fmaf(w, vCg[k], acc[1]);
          acc[2] = fmaf(w, vCb[k], acc[2]); acc[3] = fmaf(w, vD[k], acc[3]);
          acc[4] += q;
          if (k & 1) acc[5] += q;
          if (k >> 1) acc[6] += q;
          acc[7] = fmaf(q, pc.uv[k], acc[7]);
        }
        any |= gob;
      };
#if TGS_K7_FIRSTQ
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      const std::true_type F{}; const std::false_type A{};
      // a four-way scalar switch on the first quadrant (separate code per case: an `if (first)` inside one body is
      // if-converted into eight more selects per quadrant)
      switch (__builtin_ctz(m | 16u)) {
        case 0: quad(I0{}, F); if (m & 2u) quad(I1{}, A); if (m & 4u) quad(I2{}, A); if (m & 8u) quad(I3{}, A); break;
        case 1: quad(I1{}, F); if (m & 4u) quad(I2{}, A); if (m & 8u) quad(I3{}, A); break;
        case 2: quad(I2{}, F); if (m & 8u) quad(I3{}, A); break;
        case 3: quad(I3{}, F); break;
        default:     // (cannot happen: the walk only visits entries with a quadrant bit) keep the sums defined
#pragma unroll
          for (int c = 0; c < RED_ROWS; c++) acc[c] = 0.f;
      }
#else
#pragma unroll
      for (int c = 0; c < RED_ROWS; c++) acc[c] = 0.f;
      if (m & 1u) quad(std::integral_constant<int, 0>{}, std::false_type{});
      if (m & 2u) quad(std::integral_constant<int, 1>{}, std::false_type{});
      if (m & 4u) quad(std::integral_constant<int, 2>{}, std::false_type{});
      if (m & 8u) quad(std::integral_constant<int, 3>{}, std::false_type{});
#endif
    };
    // the weighted transposed sum of one Gaussian's 64 x 8 lane values out of `rd` (see the kernel's head comment)
    auto reduce_rows = [&](const float* rd) {
      const float4 r0 = ld4(rd), r1 = ld4(rd + 4), r2 = ld4(rd + 8), r3 = ld4(rd + 12);
      float o0 = r0.x * wt[0], o1 = r1.x * wt[4], o2 = r2.x * wt[8], o3 = r3.x * wt[12];
      o0 = fmaf(r0.y, wt[1], o0); o1 = fmaf(r1.y, wt[5], o1); o2 = fmaf(r2.y, wt[9], o2); o3 = fmaf(r3.y, wt[13], o3);
      o0 = fmaf(r0.z, wt[2], o0); o1 = fmaf(r1.z, wt[6], o1); o2 = fmaf(r2.z, wt[10], o2); o3 = fmaf(r3.z, wt[14], o3);
      o0 = fmaf(r0.w, wt[3], o0); o1 = fmaf(r1.w, wt[7], o1); o2 = fmaf(r2.w, wt[11], o2); o3 = fmaf(r3.w, wt[15], o3);
      return (o0 + o1) + (o2 + o3);
    };
#if TGS_K7_RECPF
    // (A/B switch) the NEXT entry's staged record is requested from LDS before the current entry is blended: its three
    // broadcast ds_read_b128 are in flight under ~70 VALU instead of in front of them (+12 VGPRs: 124, still 4 waves / SIMD)
    if (rem == 0ull) return;
    int jn = 63 - __builtin_clzll(rem);
    rem &= ~(1ull << jn);
    float4 na = recs[jn * 3], nb = recs[jn * 3 + 1], nc = recs[jn * 3 + 2];
    for (;;) {
      const int j = jn;
      const float4 qa = na, qb = nb, qc = nc;
      const bool more = rem != 0ull;
      if (more) {
        jn = 63 - __builtin_clzll(rem);
        rem &= ~(1ull << jn);
        na = recs[jn * 3]; nb = recs[jn * 3 + 1]; nc = recs[jn * 3 + 2];
      }
      float acc[RED_ROWS];
      unsigned long long any;
      blend(j, qa, qb, qc, acc, any);
      if (any != 0ull) {
#pragma unroll
        for (int c = 0; c < RED_ROWS; c++) red[c * RED_RS + lane] = acc[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float O = reduce_rows(red_rd);
        O = fold_xor16(O);
        O = fold_xor32(O);
        if (lane < 16) reinterpret_cast<float*>(sums)[j * 16 + lane] = O;
        __builtin_amdgcn_wave_barrier();
      }
      if (!more) break;
    }
#else
    while (rem) {
      const int j = 63 - __builtin_clzll(rem);
      rem &= ~(1ull << j);
      float acc[RED_ROWS];
      unsigned long long any;
      blend(j, recs[j * 3], recs[j * 3 + 1], recs[j * 3 + 2], acc, any);
#if TGS_K7_PAIR
      // Two Gaussians per reduction round (VERDICT r4 next #9a): the second one's blend runs before the first one's
      // scatter -> transposed read round trip is waited for, and the two round trips overlap.  Same sums in the same
      // order per Gaussian: bit-identical.  Costs 8 + ~16 VGPRs (3 waves per SIMD) and a second scratch image.
      int j2 = -1;
      float acc2[RED_ROWS];
      unsigned long long any2 = 0ull;
      if (rem) {
        j2 = 63 - __builtin_clzll(rem);
        rem &= ~(1ull << j2);
        blend(j2, recs[j2 * 3], recs[j2 * 3 + 1], recs[j2 * 3 + 2], acc2, any2);
      }
      if (any != 0ull && any2 != 0ull) {
#pragma unroll
        for (int c = 0; c < RED_ROWS; c++) { red[c * RED_RS + lane] = acc[c]; red[RED_ROWS * RED_RS + c * RED_RS + lane] = acc2[c]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float O = reduce_rows(red_rd), O2 = reduce_rows(red_rd + RED_ROWS * RED_RS);
        O = fold_xor16(O); O2 = fold_xor16(O2);
        O = fold_xor32(O); O2 = fold_xor32(O2);
        if (lane < 16) { reinterpret_cast<float*>(sums)[j * 16 + lane] = O; reinterpret_cast<float*>(sums)[j2 * 16 + lane] = O2; }
        __builtin_amdgcn_wave_barrier();
        continue;
      }
      if (any2 != 0ull) {        // only the second one contributed: it takes the single path below
#pragma unroll
        for (int c = 0; c < RED_ROWS; c++) acc[c] = acc2[c];
        any = any2;
      }
      const int jr = (any2 != 0ull) ? j2 : j;
#else
      const int jr = j;
#endif
      if (any != 0ull) {
#pragma unroll
        for (int c = 0; c < RED_ROWS; c++) red[c * RED_RS + lane] = acc[c];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float O = reduce_rows(red_rd);
        O = fold_xor16(O);
        O = fold_xor32(O);
        if (lane < 16) reinterpret_cast<float*>(sums)[jr * 16 + lane] = O;
        __builtin_amdgcn_wave_barrier();
      }
    }
#endif
    };
    if (__ballot(lane < cnt && a0.w > CLAMP_FREE_OPACITY) != 0ull) walk(std::true_type{});
    else walk(std::false_type{});
    __syncthreads();
    if (lane < cnt) {
      const float4 s0 = sums[lane * 4], s1 = sums[lane * 4 + 1], s2 = sums[lane * 4 + 2], s3 = sums[lane * 4 + 3];
      const float gx = a0.x, gy = a0.y;           // centre relative to the tile centre (set while staging)
      const float A = a1.x, B = a1.y, Cc = a1.z;
      const float Q0 = s1.x, Qu = s1.y + s1.z, Qv = s1.w + s2.x, Quu = s2.y + s2.z, Qvv = s2.w + s3.x, Quv = s3.y;
      const float Mx = -(gx * Q0 - Qu), My = -(gy * Q0 - Qv);
      const float Mxx = -(gx * gx * Q0 - 2.f * gx * Qu + Quu);
      const float Mxy = -(gx * gy * Q0 - gx * Qv - gy * Qu + Quv);
      const float Myy = -(gy * gy * Q0 - 2.f * gy * Qv + Qvv);
      float* o = partials + P * TGS_PARTIAL_FLOATS;
      st4(o, make_float4(A * Mx + B * My, B * Mx + Cc * My, s0.w, Q0 / a0.w));
      st4(o + 4, make_float4(0.5f * Mxx, Mxy, 0.5f * Myy, s0.x));
      st4(o + 8, make_float4(s0.y, s0.z, 0.f, 0.f));
    }
  }
}


// ---------------------------------------------------------------------------------------------
// K7 in 4x4-block form (round 6: built and measured -- VERDICT r5 next #2; NOT the default)
// ---------------------------------------------------------------------------------------------
// The forward's block form (k_raster_fwd_blocks) carried over to the backward: the four 16-lane DPP rows of the
// wave own the four 4x4 blocks of the current quadrant and every row walks ITS block's list (back to front), so one
// wave instruction works on four different Gaussians and the lanes that evaluate a pixel the Gaussian does not touch
// drop from 63 % to ~50 %.  What the forward does not have is the per-Gaussian accumulation: here
// every iteration ends with a reduction of ten values over the row's 16 lanes (DPP: 4 steps each) and lane 0 of the
// row adds the row's totals to a ROW-PRIVATE accumulator acc[row][j][12] in LDS (two rows may meet the same Gaussian
// in one iteration, and a float LDS atomic costs 3 cycles per active lane, profiles/r4_lds_atomic_lanes.txt); at the end
// of the batch lane j adds the four rows' accumulators in row order and converts the moments as k_raster_bwd does.
// Same decisions as the forward (eval_s on the same pixel constants, `s <= log2 255` below the pixel's stop position);
// the sums are formed in a different order than k_raster_bwd's, so the two differ by rounding only.
// Cost model and measurement: DESIGN.md section 5.1e.
// sum over the 16 lanes of a DPP row, in every lane: four v_add_f32_dpp.  (Written as inline asm: from
// __builtin_amdgcn_update_dpp(0, v, ..) + add the compiler emits v_mov_b32_dpp + v_add_f32 -- 8 instead of 4 VALU per
// value, 80 instead of 40 per iteration of this kernel: the first build issued 462 M VALU instructions per cfg3 frame
// against the one-wave form's 233 M, profiles/r6_pmc_k7_forms_cfg3.json.)
// Ten values at once, stage by stage: a DPP operand must have been written at least two VALU slots earlier (the
// compiler's hazard recogniser does not look inside asm), which the nine other values of a stage provide; one s_nop
// covers the first stage's first operands.
#define TGS_DPP_STAGE(MOD)                                                                      \
  _Pragma("unroll") for (int i = 0; i < 10; i++)                                                 \
      asm volatile("v_add_f32_dpp %0, %1, %1 " MOD " row_mask:0xf bank_mask:0xf" : "=v"(t[i]) : "v"(c[i]));  \
  _Pragma("unroll") for (int i = 0; i < 10; i++) c[i] = t[i];
__device__ __forceinline__ void row_sum16x10(float (&c)[10]) {
  float t[10];
  asm volatile("s_nop 1");
  TGS_DPP_STAGE("quad_perm:[1,0,3,2]")
  TGS_DPP_STAGE("quad_perm:[2,3,0,1]")
  TGS_DPP_STAGE("row_half_mirror")
  TGS_DPP_STAGE("row_mirror")
}
#undef TGS_DPP_STAGE
template <int CTRL>
__device__ __forceinline__ int row_dpp_maxi(int v) {
  const int s = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
  return max(v, s);
}
__device__ __forceinline__ int row_max16i(int v) {
  v = row_dpp_maxi<0xB1>(v); v = row_dpp_maxi<0x4E>(v); v = row_dpp_maxi<0x141>(v); v = row_dpp_maxi<0x140>(v);
  return v;
}

__global__ __launch_bounds__(64) void k_raster_bwd_blocks(
    CamK cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const int32_t* __restrict__ sorted_gid, const int32_t* __restrict__ tile_start,
    const float* __restrict__ out_rgb, const float* __restrict__ out_depth,
    const float* __restrict__ final_T, const int32_t* __restrict__ stop_pos, const float* __restrict__ v_rgb,
    const float* __restrict__ v_depth, const float* __restrict__ v_alpha, LossK loss,
    float* __restrict__ partials, float* __restrict__ tile_loss, const int32_t* __restrict__ tile_order) {
  const int tile = tile_order ? tile_order[blockIdx.x] : xcd_tile(blockIdx.x, T_total);
  if (tile >= T_total) return;
  const int lane = threadIdx.x;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  // lane -> pixel as in k_raster_fwd_blocks: DPP row g owns block (g & 1, g >> 1) of every quadrant
  const int g = lane >> 4, lx = 4 * (g & 1) + (lane & 3), ly = 4 * (g >> 1) + ((lane >> 2) & 3);
  PixConst pc;
  pc.u[0] = (float)lx - 7.5f; pc.u[1] = pc.u[0] + 8.f;
  pc.v[0] = (float)ly - 7.5f; pc.v[1] = pc.v[0] + 8.f;
#pragma unroll
  for (int i = 0; i < 2; i++) { pc.uu[i] = pc.u[i] * pc.u[i]; pc.vv[i] = pc.v[i] * pc.v[i]; }
#pragma unroll
  for (int k = 0; k < 4; k++) pc.uv[k] = pc.u[k & 1] * pc.v[k >> 1];
  const int start = tile_start[tile], end = tile_start[tile + 1];
  const int n = end - start;

  float T[4] = {1.f, 1.f, 1.f, 1.f};
  float vCr[4], vCg[4], vCb[4], vD[4], SX[4];
  int lim[4];
  float l_l1 = 0.f, l_dep = 0.f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int px = tx * TGS_BLOCK + 8 * (k & 1) + lx;
    const int py = ty * TGS_BLOCK + 8 * (k >> 1) + ly;
    lim[k] = 0; vCr[k] = vCg[k] = vCb[k] = vD[k] = SX[k] = 0.f;
    if (px < cam.W && py < cam.H) {
      const size_t p = (size_t)py * cam.W + px;
      const float Tf = final_T[p];
      T[k] = Tf;
      lim[k] = min(stop_pos[p], n);
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
            if (1.f - Tf > 1e-10f) vA += -gdh * dhat * ia;
          }
        }
      }
      const float bgdot = cam.bg[0] * vCr[k] + cam.bg[1] * vCg[k] + cam.bg[2] * vCb[k];
      SX[k] = -Tf * (vA - bgdot);
    }
  }
  if (tile_loss) {
    const float a = wave_sum(l_l1), b = wave_sum(l_dep);
    if (lane == 0) { tile_loss[2 * tile] = a; tile_loss[2 * tile + 1] = b; }
  }
  if (n == 0) return;
  // how far into the list each of the 16 blocks reaches (wave-uniform: SGPRs), block of (row gg, quadrant k) = 4 (2 (k >> 1) + (gg >> 1)) + 2 (k & 1) + (gg & 1)
  int blim[16];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int r = row_max16i(lim[k]);
#pragma unroll
    for (int gg = 0; gg < 4; gg++)
      blim[4 * (2 * (k >> 1) + (gg >> 1)) + 2 * (k & 1) + (gg & 1)] = __builtin_amdgcn_readlane(r, 16 * gg);
  }
  int tmax = 0;
#pragma unroll
  for (int b = 0; b < 16; b++) tmax = max(tmax, blim[b]);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = start + tmax + lane; i < end; i += 64) {       // behind the last stop position nothing received a gradient
    const int gid = sorted_gid[i];
    const size_t P = pair_index(group_base, gid, ld4(splats + (size_t)gid * TGS_SPLAT_FLOATS + 8), tx, ty);
    float* o = partials + P * TGS_PARTIAL_FLOATS;
    st4(o, z4); st4(o + 4, z4); st4(o + 8, z4);
  }
  if (tmax == 0) return;

  __shared__ float4 recs[65 * 3];                 // 64 staged Gaussians + the null record (alpha = 0)
  __shared__ unsigned int lists4[16 * 16];        // 16 lists of 64 one-byte indices, padded with 64 = null
  __shared__ float4 acc4[4 * 64 * 3];             // [row][Gaussian j][12]: v_rgb 3, v_depth, Q0, Qu, Qv, Quu, Quv, Qvv, -, -
  unsigned char* lists = reinterpret_cast<unsigned char*>(lists4);
  float* acc = reinterpret_cast<float*>(acc4);
  if (lane == 0) {
    recs[64 * 3] = make_float4(3.0e38f, 0.f, 0.f, 0.f);
    recs[64 * 3 + 1] = z4;
    recs[64 * 3 + 2] = z4;
  }
  int myblock[4];
#pragma unroll
  for (int k = 0; k < 4; k++) myblock[k] = 4 * (2 * (k >> 1) + (g >> 1)) + 2 * (k & 1) + (g & 1);
  const bool row_head = (lane & 15) == 0;

  for (int base = start + ((tmax - 1) & ~63); base >= start; base -= 64) {
    const int rel = base - start;
    const int cnt_b = min(64, tmax - rel);
    size_t P = 0;
    float4 a0 = z4, a1 = z4;
    unsigned my_mask = 0u;
    __syncthreads();
    if (lane < cnt_b) {
      const int gid = sorted_gid[base + lane];
      const float* r = splats + (size_t)gid * TGS_SPLAT_FLOATS;
      a0 = ld4(r); a1 = ld4(r + 4);
      const float4 a2 = ld4(r + 8);
      float gx, gy;
      centre_rel(a0, a2, tx, ty, cam.pix_center, gx, gy);
      my_mask = block_mask16(gx, gy, a1.x, a1.y, a1.z, -__log2f(a0.w));
      a0.x = gx; a0.y = gy;                       // kept for the conversion of the moments below
      const TileRec t = make_tile_rec<false>(a0, a1, a2, gx, gy);
      recs[lane * 3] = t.a; recs[lane * 3 + 1] = t.b; recs[lane * 3 + 2] = t.c;
      P = pair_index(group_base, gid, a2, tx, ty);
      // a block only needs the entries below its own furthest stop position
      unsigned reach = 0u;
#pragma unroll
      for (int b = 0; b < 16; b++) reach |= (rel + lane < blim[b]) ? (1u << b) : 0u;
      my_mask &= reach;
    }
    const bool clampy = __ballot(lane < cnt_b && a0.w > CLAMP_FREE_OPACITY) != 0ull;
    reinterpret_cast<uint4*>(lists4)[lane] = make_uint4(0x40404040u, 0x40404040u, 0x40404040u, 0x40404040u);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      acc4[(r * 64 + lane) * 3] = z4; acc4[(r * 64 + lane) * 3 + 1] = z4; acc4[(r * 64 + lane) * 3 + 2] = z4;
    }
    __syncthreads();
    int cnt[16];
#pragma unroll
    for (int b = 0; b < 16; b++) {     // lists in REVERSE list order: entry 0 = the block's deepest Gaussian of this batch
      const bool in = (my_mask >> b) & 1u;
      const unsigned long long bal = __ballot(in);
      cnt[b] = __popcll(bal);
      if (in) lists[b * 64 + cnt[b] - 1 - (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (unsigned char)lane;
    }
    __syncthreads();
    auto walk = [&](auto mayclamp) {
      constexpr bool MAYCLAMP = decltype(mayclamp)::value;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int b0 = 8 * (k >> 1) + 2 * (k & 1);                 // blocks b0, b0 + 1, b0 + 4, b0 + 5
        const int nit = max(max(cnt[b0], cnt[b0 + 1]), max(cnt[b0 + 4], cnt[b0 + 5]));
        const BLK_T* mylist = reinterpret_cast<const BLK_T*>(lists4) + myblock[k] * (64 / BLK_U);
        for (int i4 = 0; BLK_U * i4 < nit; i4++) {
          const unsigned idx4 = mylist[i4];
#pragma unroll
          for (int e = 0; e < BLK_U; e++) {
            const int j = (idx4 >> (8 * e)) & 0xffu;               // 64 = null record: s = 3e38, alpha = 0
            const float4 qa = recs[j * 3], qb = recs[j * 3 + 1], qc = recs[j * 3 + 2];
            const float s = eval_s(qa, qb, pc, k);
            const float ex = __builtin_amdgcn_exp2f(-s);
            const float al0 = MAYCLAMP ? fminf(ALPHA_MAX, ex) : ex;
            const bool go = (s <= LOG2_255) & (rel + j < lim[k]);   // the forward's alpha test below the pixel's stop position
            const float al = go ? al0 : 0.f;
            const float ra = __builtin_amdgcn_rcpf(1.0f - al);
            const float Tp = T[k] * ra;
            const float w = al * Tp;
            float cv = qb.w * vCr[k];
            cv = fmaf(qc.x, vCg[k], cv); cv = fmaf(qc.y, vCb[k], cv); cv = fmaf(qb.z, vD[k], cv);
            const float z = w * cv;
            float q = fmaf(-(al * ra), SX[k], z);
            if constexpr (MAYCLAMP) q *= fmaxf(ex * (1.0f / ALPHA_MAX), 1.0f);
            SX[k] += z;
            T[k] = Tp;
            // the row's ten sums for ITS Gaussian (tile-centred moments, converted at the end of the batch)
            float c[10];
            c[0] = w * vCr[k]; c[1] = w * vCg[k]; c[2] = w * vCb[k]; c[3] = w * vD[k];
            c[4] = q; c[5] = q * pc.u[k & 1]; c[6] = q * pc.v[k >> 1];
            c[7] = q * pc.uu[k & 1]; c[8] = q * pc.uv[k]; c[9] = q * pc.vv[k >> 1];
            row_sum16x10(c);
            if (row_head && j < 64) {
              float* ac = acc + (g * 64 + j) * 12;
              float4 x = ld4(ac), y = ld4(ac + 4);
              float2 t2 = *reinterpret_cast<const float2*>(ac + 8);
              x.x += c[0]; x.y += c[1]; x.z += c[2]; x.w += c[3];
              y.x += c[4]; y.y += c[5]; y.z += c[6]; y.w += c[7];
              t2.x += c[8]; t2.y += c[9];
              st4(ac, x); st4(ac + 4, y);
              *reinterpret_cast<float2*>(ac + 8) = t2;
            }
          }
        }
      }
    };
    if (clampy) walk(std::true_type{}); else walk(std::false_type{});
    __syncthreads();
    if (lane < cnt_b) {
      float4 s0 = acc4[lane * 3], s1 = acc4[lane * 3 + 1], s2 = acc4[lane * 3 + 2];
#pragma unroll
      for (int r = 1; r < 4; r++) {        // the four rows' totals in row order
        const float4 t0 = acc4[(r * 64 + lane) * 3], t1 = acc4[(r * 64 + lane) * 3 + 1], t2 = acc4[(r * 64 + lane) * 3 + 2];
        s0.x += t0.x; s0.y += t0.y; s0.z += t0.z; s0.w += t0.w;
        s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
        s2.x += t2.x; s2.y += t2.y;
      }
      const float gx = a0.x, gy = a0.y;
      const float A = a1.x, B = a1.y, Cc = a1.z;
      const float Q0 = s1.x, Qu = s1.y, Qv = s1.z, Quu = s1.w, Quv = s2.x, Qvv = s2.y;
      const float Mx = -(gx * Q0 - Qu), My = -(gy * Q0 - Qv);
      const float Mxx = -(gx * gx * Q0 - 2.f * gx * Qu + Quu);
      const float Mxy = -(gx * gy * Q0 - gx * Qv - gy * Qu + Quv);
      const float Myy = -(gy * gy * Q0 - 2.f * gy * Qv + Qvv);
      float* o = partials + P * TGS_PARTIAL_FLOATS;
      st4(o, make_float4(A * Mx + B * My, B * Mx + Cc * My, s0.w, Q0 / a0.w));
      st4(o + 4, make_float4(0.5f * Mxx, Mxy, 0.5f * Myy, s0.x));
      st4(o + 8, make_float4(s0.y, s0.z, 0.f, 0.f));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K7 for LONG tiles: four waves per tile, one per 8x8 quadrant
// ---------------------------------------------------------------------------------------------
// k_raster_bwd gives a tile to ONE wave, which walks the list entry by entry (~750 - 1000 cycles each when the wave is
// alone on its SIMD).  In an object-centric scene -- the reference's kind: one object on a table -- a few hundred tiles
// carry walks of 700 - 1600 entries while the frame's whole work would fit 100 - 200 entries per wave slot: the launch
// lasts as long as its longest tile (tools/tile_load.py: critical path 4.5 - 16x the balanced load at 300 k Gaussians,
// 720p) and most of the GPU idles.  Here a workgroup of four waves owns the tile: wave k keeps ONE pixel per lane (its
// quadrant), walks only the entries whose quadrant bit is set and that its own pixels still reach, and reduces its 64
// lanes with the same weighted transposed LDS read; the four quadrant totals of an entry meet in LDS at the end of the
// batch.  Per entry a wave executes a third of the instructions, so the chain is ~2.5 - 3x shorter; the sum of the
// work is ~25 % larger (one reduction per quadrant instead of one per tile), which is why only chain-bound frames
// (frame_is_chain_bound) take this path.  Same decisions (the forward's compare on the same bits), same per-pixel arithmetic;
// the quadrant totals are added in the order k = 0, 1, 2, 3 (deterministic; differs from the one-wave kernel's order
// by rounding).  Losses (tile_loss) are k_raster_bwd's, which visits every tile first.
#ifdef TGS_QUAD_TIMING   // developer build (tools/k7_quad_timing.py): where a tile's time goes, per schedule slot and wave
__device__ unsigned long long g_quad_dbg[8192 * 16];
#define QT_NOW() __builtin_amdgcn_s_memtime()
#else
#define QT_NOW() 0ull
#endif
#ifndef TGS_QUAD_WAVES
#define TGS_QUAD_WAVES 4      // waves per SIMD asked of the register allocator: 109 VGPRs; 5 (96 VGPRs, 42 spilled) measures 2 % slower on the step
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TGS_QUAD_WAVES, 8))) void k_raster_bwd_quad(
    CamK cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const int32_t* __restrict__ sorted_gid, const int32_t* __restrict__ tile_start,
    const float* __restrict__ out_rgb, const float* __restrict__ out_depth,
    const float* __restrict__ final_T, const int32_t* __restrict__ stop_pos, const float* __restrict__ v_rgb,
    const float* __restrict__ v_depth, const float* __restrict__ v_alpha, LossK loss,
    float* __restrict__ partials, const int32_t* __restrict__ tile_order, QuadRule quad, int n_slots) {
  if (!frame_is_chain_bound(tile_start, T_total, quad)) return;   // the usual case: nothing to do (every wave alike)
  const int tid = threadIdx.x, lane = tid & 63, k = tid >> 6;   // k = this wave's quadrant
  __shared__ int s_qlim[4];
  __shared__ int s_clamp[2], s_stager[2], s_arrive;
  __shared__ float4 recs2[2 * 64 * 3];  // two batches of staged records: the next one is staged while this one is walked
  __shared__ float4 sums[4 * 64 * 4];   // [quadrant][Gaussian j][16 slots]
  constexpr int RED_RS = 68;
  constexpr int RED_ROWS = 8;
  __shared__ float4 red4[4 * RED_ROWS * RED_RS / 4];
  // A resident grid takes the schedule's slots (block b of the one-wave kernel = entry b of tile_order) as workgroups
  // become free: slot 8 i + x belongs to XCD x, whose workgroups draw i from the XCD's counter (zeroed by k_raster_bwd,
  // the launch before this one) -- longest lists first, every tile on the XCD whose L2 holds its neighbours' records.
  __shared__ int s_slot;
  const int xcc = blockIdx.x & (TGS_XCDS - 1);    // workgroup b runs on XCD b % 8
  int32_t* ctr = frame_scratch(tile_start);
  int turn = 0;     // own XCD first, then the others' leftovers (correct wherever the workgroups were placed)
  for (;;) {
  __syncthreads();                          // (the previous tile's s_slot / s_qlim have been read)
  if (tid == 0) {
    const int x = (xcc + turn) & (TGS_XCDS - 1);
    s_slot = TGS_XCDS * atomicAdd(ctr + TGS_SLOTCTR_AT(T_total, x), 1) + x;
  }
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(s_slot);   // wave-uniform for the compiler too (scalar loads, scalar loops)
  if (slot >= n_slots) {
    if (++turn == TGS_XCDS) break;
    continue;
  }
  const int tile = tile_order ? tile_order[slot] : xcd_tile(slot, T_total);
  if (tile >= T_total) continue;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  const int start = tile_start[tile], end = tile_start[tile + 1];
  const int n = end - start;
  if (n <= quad.min_walk) continue;
  const int px = tx * TGS_BLOCK + 8 * (k & 1) + (lane & 7);
  const int py = ty * TGS_BLOCK + 8 * (k >> 1) + (lane >> 3);
  const bool inb = px < cam.W && py < cam.H;
  const size_t p = (size_t)py * cam.W + px;
  const int lim = inb ? min(stop_pos[p], n) : 0;   // number of leading list positions that may contribute to this pixel
  const int qlim = wave_minmax_i<true>(lim);
  if (lane == 0) s_qlim[k] = qlim;
  __syncthreads();
  const int tmax = max(max(s_qlim[0], s_qlim[1]), max(s_qlim[2], s_qlim[3]));
  if (tmax <= quad.min_walk) continue;   // k_raster_bwd's tile
  if (quad.scan_min > 0 && slot < quad.scan_heads && tmax > quad.scan_min) continue;   // k_raster_bwd_scan's tile

  // per-pixel state, the arithmetic of k_raster_bwd's prologue for pixel slot k of the lane
  float T = 1.f, vCr = 0.f, vCg = 0.f, vCb = 0.f, vD = 0.f, SX = 0.f;
  if (inb) {
    const float Tf = final_T[p];
    T = Tf;
    float vA = v_alpha ? v_alpha[p] : 0.f;
    if (v_rgb) { vCr = v_rgb[3 * p]; vCg = v_rgb[3 * p + 1]; vCb = v_rgb[3 * p + 2]; }
    if (v_depth) vD = v_depth[p];
    if (loss.on) {
      if (loss.gt_rgb) {
        const float d0 = out_rgb[3 * p] - loss.gt_rgb[3 * p];
        const float d1 = out_rgb[3 * p + 1] - loss.gt_rgb[3 * p + 1];
        const float d2 = out_rgb[3 * p + 2] - loss.gt_rgb[3 * p + 2];
        vCr += loss.l1w * ((d0 > 0.f) - (d0 < 0.f));
        vCg += loss.l1w * ((d1 > 0.f) - (d1 < 0.f));
        vCb += loss.l1w * ((d2 > 0.f) - (d2 < 0.f));
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
          const float gdh = 2.f * wgt * r;
          vD += gdh * ia;
          if (1.f - Tf > 1e-10f) vA += -gdh * dhat * ia;
        }
      }
    }
    const float bgdot = cam.bg[0] * vCr + cam.bg[1] * vCg + cam.bg[2] * vCb;
    SX = -Tf * (vA - bgdot);
  }
  // pixel constants of the quadrant (the operands eval_s takes for slot k)
  const PixConst pc = make_pix_const(lane);
  const float pu = (k & 1) ? pc.u[1] : pc.u[0], pv = (k >> 1) ? pc.v[1] : pc.v[0];
  const float puu = (k & 1) ? pc.uu[1] : pc.uu[0], pvv = (k >> 1) ? pc.vv[1] : pc.vv[0];
  const float puv = k == 0 ? pc.uv[0] : (k == 1 ? pc.uv[1] : (k == 2 ? pc.uv[2] : pc.uv[3]));
  const float q5 = (k & 1) ? 1.f : 0.f, q6 = (k >> 1) ? 1.f : 0.f;   // Sx / Sy rows: the quadrant's 8-pixel offset

  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = start + tmax + tid; i < end; i += 256) {   // behind the last stop position nothing received a gradient
    const int gid = sorted_gid[i];
    const size_t P = pair_index(group_base, gid, ld4(splats + (size_t)gid * TGS_SPLAT_FLOATS + 8), tx, ty);
    float* o = partials + P * TGS_PARTIAL_FLOATS;
    st4(o, z4); st4(o + 4, z4); st4(o + 8, z4);
  }

  float* red = reinterpret_cast<float*>(red4) + k * (RED_ROWS * RED_RS);
  float4* sums_k = sums + k * (64 * 4);
  const int red_c = lane & 15;
  const int red_part = (lane >> 4) ^ ((red_c >= 4 && red_c < 12) ? 1 : 0);
  float wt[16];
  const float* red_rd;
  {
    const unsigned long long ROWS = 0x0076454645443210ull;
    red_rd = red + (int)((ROWS >> (4 * red_c)) & 15ull) * RED_RS + red_part * 16;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const float u0 = (float)(i & 7) - 7.5f;
      const float v0 = (float)(2 * red_part + (i >> 3)) - 7.5f;
      float w = 1.f;                                   // c 0..4, 13
      w = (red_c == 5) ? u0 : w;
      w = (red_c == 7) ? v0 : w;
      w = (red_c == 6 || red_c == 8) ? 8.f : w;
      w = (red_c == 9) ? u0 * u0 : w;
      w = (red_c == 11) ? v0 * v0 : w;
      w = (red_c == 10) ? 16.f * u0 + 64.f : w;
      w = (red_c == 12) ? 16.f * v0 + 64.f : w;
      w = (red_c >= 14) ? 0.f : w;
      wt[i] = w;
    }
  }
  // Staging (64 records: two dependent loads, the tile-centred quadratic, the quadrant mask: ~10 % of a tile's life when
  // wave 0 did it between two barriers with the other three waves idle) is done by whichever wave finishes its walk of
  // the CURRENT batch first -- a batch lasts as long as its slowest quadrant, so somebody has the time (tools/
  // k7_quad_timing.py: 21 % of all wave time, 46 % of the deepest tile's, was spent waiting for it) -- into the other
  // half of recs2; the stager keeps the batch's pair indices and conics and converts / writes its records after the walk.
  struct Staged { size_t P; float4 a0, a1; };
  auto stage = [&](int base, int half) -> Staged {
    const int cnt = min(64, tmax - (base - start));
    Staged g; g.P = 0; g.a0 = z4; g.a1 = z4;
    if (lane < cnt) {
      const int gid = sorted_gid[base + lane];
      const float* r = splats + (size_t)gid * TGS_SPLAT_FLOATS;
      g.a0 = ld4(r); g.a1 = ld4(r + 4);
      const float4 a2 = ld4(r + 8);
      float gx, gy;
      centre_rel(g.a0, a2, tx, ty, cam.pix_center, gx, gy);
      g.a0.x = gx; g.a0.y = gy;                       // kept for the conversion of the moments below
      const TileRec t = make_tile_rec(g.a0, g.a1, a2, gx, gy);
      float4* rc = recs2 + half * (64 * 3);
      rc[lane * 3] = t.a; rc[lane * 3 + 1] = t.b; rc[lane * 3 + 2] = t.c;
      g.P = pair_index(group_base, gid, a2, tx, ty);
    }
    const unsigned long long hot = __ballot(lane < cnt && g.a0.w > CLAMP_FREE_OPACITY);
    if (lane == 0) { s_clamp[half] = hot != 0ull; s_stager[half] = k; }
    return g;
  };
  const int base0 = start + ((tmax - 1) & ~63);
  Staged cur, nxt;
  cur.P = 0; cur.a0 = z4; cur.a1 = z4; nxt = cur;
  if (k == 0) cur = stage(base0, 0);
  if (tid == 0) s_arrive = 0;
  [[maybe_unused]] unsigned long long qt0 = QT_NOW();
  __syncthreads();
  int half = 0;
  for (int base = base0; base >= start; base -= 64, half ^= 1) {
    const int rel = base - start;
    const int cnt = min(64, tmax - rel);
    const float4* recs = recs2 + half * (64 * 3);
    [[maybe_unused]] const unsigned long long qt1 = QT_NOW();
    sums_k[lane * 4] = z4; sums_k[lane * 4 + 1] = z4; sums_k[lane * 4 + 2] = z4; sums_k[lane * 4 + 3] = z4;   // (only this wave writes them)

    // this quadrant's entries: mask bit k, below the quadrant's own furthest stop position
    const unsigned my_mask = lane < cnt ? __float_as_uint(recs[lane * 3 + 2].w) : 0u;
    const int r = qlim - rel;
    const unsigned long long keep = r >= 64 ? ~0ull : (r <= 0 ? 0ull : ((1ull << r) - 1ull));
    const unsigned long long rem0 = __ballot((my_mask >> k) & 1u) & keep;
    auto walk = [&](auto mayclamp) {
    constexpr bool MAYCLAMP = decltype(mayclamp)::value;
    unsigned long long rem = rem0;
    while (rem) {
      const int j = 63 - __builtin_clzll(rem);
      rem &= ~(1ull << j);
      const float4 qa = recs[j * 3], qb = recs[j * 3 + 1], qc = recs[j * 3 + 2];
      const int pos = rel + j;
      float s = fmaf(qa.y, pu, qa.x);          // eval_s for slot k
      s = fmaf(qa.z, pv, s);
      s = fmaf(qa.w, puu, s);
      s = fmaf(qb.x, puv, s);
      s = fmaf(qb.y, pvv, s);
      const float e = __builtin_amdgcn_exp2f(-s);
      const float al0 = MAYCLAMP ? fminf(ALPHA_MAX, e) : e;
      const bool ok = s <= LOG2_255, in = pos < lim;
      const bool go = ok & in;
      const unsigned long long gob = __builtin_amdgcn_ballot_w64(ok) & __builtin_amdgcn_ballot_w64(in);
      const float al = go ? al0 : 0.f;
      const float ra = __builtin_amdgcn_rcpf(1.0f - al);
      const float Tp = T * ra;
      const float w = al * Tp;
      float cv = qb.w * vCr;
      cv = fmaf(qc.x, vCg, cv); cv = fmaf(qc.y, vCb, cv); cv = fmaf(qb.z, vD, cv);
      const float z = w * cv;
      float q = fmaf(-(al * ra), SX, z);
      if constexpr (MAYCLAMP) q *= fmaxf(e * (1.0f / ALPHA_MAX), 1.0f);
      SX += z;
      T = Tp;
      if (gob != 0ull) {
        red[0 * RED_RS + lane] = w * vCr; red[1 * RED_RS + lane] = w * vCg;
        red[2 * RED_RS + lane] = w * vCb; red[3 * RED_RS + lane] = w * vD;
        red[4 * RED_RS + lane] = q; red[5 * RED_RS + lane] = q * q5;
        red[6 * RED_RS + lane] = q * q6; red[7 * RED_RS + lane] = q * puv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4 r0 = ld4(red_rd), r1 = ld4(red_rd + 4), r2 = ld4(red_rd + 8), r3 = ld4(red_rd + 12);
        float o0 = r0.x * wt[0], o1 = r1.x * wt[4], o2 = r2.x * wt[8], o3 = r3.x * wt[12];
        o0 = fmaf(r0.y, wt[1], o0); o1 = fmaf(r1.y, wt[5], o1); o2 = fmaf(r2.y, wt[9], o2); o3 = fmaf(r3.y, wt[13], o3);
        o0 = fmaf(r0.z, wt[2], o0); o1 = fmaf(r1.z, wt[6], o1); o2 = fmaf(r2.z, wt[10], o2); o3 = fmaf(r3.z, wt[14], o3);
        o0 = fmaf(r0.w, wt[3], o0); o1 = fmaf(r1.w, wt[7], o1); o2 = fmaf(r2.w, wt[11], o2); o3 = fmaf(r3.w, wt[15], o3);
        float O = (o0 + o1) + (o2 + o3);
        O = fold_xor16(O);
        O = fold_xor32(O);
        if (lane < 16) reinterpret_cast<float*>(sums_k)[j * 16 + lane] = O;
        __builtin_amdgcn_wave_barrier();
      }
    }
    };
    if (s_clamp[half]) walk(std::true_type{});
    else walk(std::false_type{});
    [[maybe_unused]] const unsigned long long qt2 = QT_NOW();
    if (base - 64 >= start) {                 // first to finish: stage the next batch
      int t = 0;
      if (lane == 0) t = atomicAdd(&s_arrive, 1);
      if (__builtin_amdgcn_readfirstlane(t) == 0) nxt = stage(base - 64, half ^ 1);
    }
    [[maybe_unused]] const unsigned long long qt3 = QT_NOW();
    __syncthreads();                          // every quadrant's totals of this batch are in LDS, the next batch is staged
    [[maybe_unused]] const unsigned long long qt4 = QT_NOW();
    const bool mine = s_stager[half] == k;
    if (mine) {
      if (lane == 0) s_arrive = 0;
      if (lane < cnt) {
      float4 s0 = sums[lane * 4], s1 = sums[lane * 4 + 1], s2 = sums[lane * 4 + 2], s3 = sums[lane * 4 + 3];
#pragma unroll
      for (int w = 1; w < 4; w++) {        // quadrant totals in the order 0, 1, 2, 3
        const float4* sw = sums + w * (64 * 4) + lane * 4;
        const float4 t0 = sw[0], t1 = sw[1], t2 = sw[2], t3 = sw[3];
        s0.x += t0.x; s0.y += t0.y; s0.z += t0.z; s0.w += t0.w;
        s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
        s2.x += t2.x; s2.y += t2.y; s2.z += t2.z; s2.w += t2.w;
        s3.x += t3.x; s3.y += t3.y; s3.z += t3.z; s3.w += t3.w;
      }
      const float4 a0 = cur.a0, a1 = cur.a1;
      const float gx = a0.x, gy = a0.y;           // centre relative to the tile centre (set while staging)
      const float A = a1.x, B = a1.y, Cc = a1.z;
      const float Q0 = s1.x, Qu = s1.y + s1.z, Qv = s1.w + s2.x, Quu = s2.y + s2.z, Qvv = s2.w + s3.x, Quv = s3.y;
      const float Mx = -(gx * Q0 - Qu), My = -(gy * Q0 - Qv);
      const float Mxx = -(gx * gx * Q0 - 2.f * gx * Qu + Quu);
      const float Mxy = -(gx * gy * Q0 - gx * Qv - gy * Qu + Quv);
      const float Myy = -(gy * gy * Q0 - 2.f * gy * Qv + Qvv);
      float* o = partials + cur.P * TGS_PARTIAL_FLOATS;
      st4(o, make_float4(A * Mx + B * My, B * Mx + Cc * My, s0.w, Q0 / a0.w));
      st4(o + 4, make_float4(0.5f * Mxx, Mxy, 0.5f * Myy, s0.x));
      st4(o + 8, make_float4(s0.y, s0.z, 0.f, 0.f));
      }
    }
    cur = nxt;                                // (meaningful in the wave that staged the next batch: the only one that will use it)
    [[maybe_unused]] const unsigned long long qt5 = QT_NOW();
    __syncthreads();                          // the totals have been consumed (the next walk overwrites them), s_arrive is 0
#ifdef TGS_QUAD_TIMING
    if (lane == 0 && slot < 8192) {
      unsigned long long* d = g_quad_dbg + slot * 16 + k * 4;
      d[0] += (qt1 - qt0) + (QT_NOW() - qt5);   // barriers around the conversion (one wave converts, three wait)
      d[1] += qt3 - qt2;                        // staging the next batch (the first wave to finish)
      d[2] += qt2 - qt1;                        // own walk
      d[3] += qt4 - qt3;                        // waiting for the slowest quadrant
    }
    qt0 = QT_NOW();
#endif
  }
  }   // slots
}


// ---------------------------------------------------------------------------------------------
// K7 for the LONGEST tiles of a chain-bound frame: a scan over the entries instead of a walk (round 6)
// ---------------------------------------------------------------------------------------------
// The four-wave launch lasts as long as its deepest tile: measured on the saved 720p checkpoints K7 = 110 us + 0.15 us x
// the frame's deepest walk (tools/k7_tail_probe.py) -- a wave alone on its SIMD issues one instruction per ~5.75 cycles
// and needs ~60 of them per (quadrant, entry), one entry after the other, because the entries of a pixel depend on each
// other through T and the sum behind.  But that dependence is a SCAN: T in front of entry j = T behind the batch x
// prod_{i >= j} 1 / (1 - alpha_i), the sum behind entry j = sum behind the batch + sum_{i > j} w_i c_i.v.
// Here a workgroup of 16 waves owns the tile, wave w its 4x4 block w.  Per batch of 64 list entries a wave compacts the
// entries that touch ITS block (the forward's exact 16-bit block mask: ~20 % of them) and goes through them 16 at a time:
// the 16 lanes of a DPP row hold 16 entries (lane 0 the deepest), the four rows four pixels; four iterations cover the
// block's 16 pixels.  Per iteration: alpha of 16 entries x 4 pixels (eval_s on the forward's bits), a 4-step DPP prefix
// product for T and a 4-step prefix sum for the sum behind, and every lane adds its (entry, pixel) terms to ten registers
// -- no cross-lane reduction per entry.  The per-pixel state lives in lane 0 of the pixel's row (T behind folded into the
// first factor, the sum behind into the first term; `row_ror:1` brings the row's totals back to lane 0): no readlanes,
// no broadcasts.  After the four iterations two lane-swap folds add the four rows, the block's sums go to its slab of an
// LDS accumulator; after the batch's one barrier every wave sums four entries over the 16 blocks (one lane per block,
// DPP row sum) and a rotating wave converts the moments and writes the 64 records.
// First build of this idea (commit 781c77f: the 64 lanes = ALL 64 entries of the batch, one pixel at a time) spent 7.8 us
// per batch -- CU-throughput bound, 16 waves x 16 pixels x 85 instructions on four SIMDs -- no better than the four-wave form.
// Same decisions as the forward; sums in scan order: rounding only (test_k7_scan_form_on_the_longest_tiles).
// MEASURED, NOT THE DEFAULT (TGS_K7_SCAN_MIN = 0; profiles/r6_ab_runs.txt): the frame's deepest tile (1233 entries)
// takes 84 us here against ~250 us in the four-wave form -- the chain IS 3x shorter -- but a workgroup saturates its
// CU's four SIMDs (16 waves x ~680 VALU per batch), so per tile-entry the form costs 1.65x the four-wave form's CU
// time (all tiles through it: 342 us against 207), and run beside the four-wave launch on its own stream the frame's
// K7 goes 221 -> ~200 us in the kernel trace while the step does not move (the one-wave launch stretches 14 -> 35 us,
// the fork / join events cost what is left).
constexpr int SCAN_ENT = 12;                       // floats per (block, entry) accumulator row (10 used; 48 B: b128-aligned)
constexpr int SCAN_SLAB = 64 * SCAN_ENT + 4;       // floats per block slab (+4: the 16 blocks' rows of one entry fall into different banks)
constexpr int SCAN_ACC = 16 * SCAN_SLAB;           // floats per buffer
constexpr int SCAN_LDS_BYTES = (2 * SCAN_ACC + 2 * 64 * SCAN_ENT + 65 * 12) * 4 + 16 * 64;

__device__ __forceinline__ float row_scan_mul(float v) {      // inclusive prefix product over the 16 lanes of every DPP row
  asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(v));
  return v;
}
__device__ __forceinline__ float row_scan_add(float v) {
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(v));
  return v;
}
template <int CTRL>
__device__ __forceinline__ float row_dpp_mov(float old, float v) {   // lanes without a source lane keep `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// x, y -> one register: the sum over the two halves (rows 0 + 2, 1 + 3) of x in the lower half, of y in the upper half
__device__ __forceinline__ float fold32(float x, float y) {
  const tgs_u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// x, y -> rows (x0 + x1, y0 + y1, x2 + x3, y2 + y3)
__device__ __forceinline__ float fold16(float x, float y) {
  const tgs_u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__global__ __launch_bounds__(1024) void k_raster_bwd_scan(
    CamK cam, int T_total, const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const int32_t* __restrict__ sorted_gid, const int32_t* __restrict__ tile_start,
    const float* __restrict__ out_rgb, const float* __restrict__ out_depth,
    const float* __restrict__ final_T, const int32_t* __restrict__ stop_pos, const float* __restrict__ v_rgb,
    const float* __restrict__ v_depth, const float* __restrict__ v_alpha, LossK loss,
    float* __restrict__ partials, const int32_t* __restrict__ tile_order, QuadRule quad) {
  if (!frame_is_chain_bound(tile_start, T_total, quad)) return;
  const int slot = blockIdx.x;
  const int tile = tile_order ? tile_order[slot] : xcd_tile(slot, T_total);
  if (tile >= T_total) return;
  const int start = tile_start[tile], end = tile_start[tile + 1];
  const int n = end - start;
  if (n <= quad.scan_min) return;                     // (walk <= list length)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ty = tile / cam.TW, tx = tile - ty * cam.TW;
  const int bx = w & 3, by = w >> 2;                  // this wave's 4x4 block
  const int row = lane >> 4, li = lane & 15;
  const bool head = li == 0;
  extern __shared__ float4 scan_lds4[];
  float* acc = reinterpret_cast<float*>(scan_lds4);                   // [2][16 blocks][64 entries][SCAN_ENT] (+ slab pads)
  float* tot = acc + 2 * SCAN_ACC;                                     // [2][64 entries][SCAN_ENT]
  float4* recs = reinterpret_cast<float4*>(tot + 2 * 64 * SCAN_ENT);  // [64 + null][3]
  unsigned char* idx = reinterpret_cast<unsigned char*>(recs + 65 * 3) + w * 64;   // this wave's compacted entry list
  __shared__ int s_blim[16];

  // pixel (k, row) of the block = (4 bx + row, 4 by + k) of the tile; its state in lane 0 of the row, constants in all 16 lanes
  float Tm[4], SXs[4], vr[4], vg[4], vb[4], vd[4];
  int lim[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int px = tx * TGS_BLOCK + 4 * bx + row, py = ty * TGS_BLOCK + 4 * by + k;
    float T = 1.f, SX = 0.f;
    lim[k] = 0; vr[k] = vg[k] = vb[k] = vd[k] = 0.f;
    if (px < cam.W && py < cam.H) {
      const size_t pi = (size_t)py * cam.W + px;
      const float Tf = final_T[pi];
      T = Tf;
      lim[k] = min(stop_pos[pi], n);
      float vA = v_alpha ? v_alpha[pi] : 0.f;
      float vCr = 0.f, vCg = 0.f, vCb = 0.f, vD = 0.f;
      if (v_rgb) { vCr = v_rgb[3 * pi]; vCg = v_rgb[3 * pi + 1]; vCb = v_rgb[3 * pi + 2]; }
      if (v_depth) vD = v_depth[pi];
      if (loss.on) {
        if (loss.gt_rgb) {
          const float d0 = out_rgb[3 * pi] - loss.gt_rgb[3 * pi];
          const float d1 = out_rgb[3 * pi + 1] - loss.gt_rgb[3 * pi + 1];
          const float d2 = out_rgb[3 * pi + 2] - loss.gt_rgb[3 * pi + 2];
          vCr += loss.l1w * ((d0 > 0.f) - (d0 < 0.f));
          vCg += loss.l1w * ((d1 > 0.f) - (d1 < 0.f));
          vCb += loss.l1w * ((d2 > 0.f) - (d2 < 0.f));
        }
        if (loss.gt_depth) {
          const float gd = loss.gt_depth[pi];
          if (gd > 0.f) {
            const float alpha = fmaxf(1.f - Tf, 1e-10f);
            const float ia = 1.0f / alpha;
            const float dhat = out_depth[pi] * ia;
            const float r = dhat - gd;
            float wgt = loss.dw;
            if (loss.unc) wgt = wgt / (loss.uw * loss.unc[pi] + loss.eps);
            const float gdh = 2.f * wgt * r;
            vD += gdh * ia;
            if (1.f - Tf > 1e-10f) vA += -gdh * dhat * ia;
          }
        }
      }
      const float bgdot = cam.bg[0] * vCr + cam.bg[1] * vCg + cam.bg[2] * vCb;
      SX = -Tf * (vA - bgdot);
      vr[k] = vCr; vg[k] = vCg; vb[k] = vCb; vd[k] = vD;
    }
    Tm[k] = head ? T : 1.f;
    SXs[k] = head ? SX : 0.f;
  }
  int klim[4];                                        // how far into the list the four pixels of iteration k reach
#pragma unroll
  for (int k = 0; k < 4; k++) klim[k] = wave_minmax_i<true>(lim[k]);
  const int blim = max(max(klim[0], klim[1]), max(klim[2], klim[3]));
  if (lane == 0) {
    s_blim[w] = blim;
    recs[64 * 3] = make_float4(3.0e38f, 0.f, 0.f, 0.f);     // the null record: alpha = 0 (every wave stores the same bits)
    recs[64 * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    recs[64 * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  int tmax = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) tmax = max(tmax, s_blim[i]);
  if (tmax <= quad.scan_min) return;                  // the four-wave launch's tile (same predicate there)
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = start + tmax + tid; i < end; i += 1024) {   // behind the last stop position nothing received a gradient
    const int gid = sorted_gid[i];
    const size_t P = pair_index(group_base, gid, ld4(splats + (size_t)gid * TGS_SPLAT_FLOATS + 8), tx, ty);
    float* o = partials + P * TGS_PARTIAL_FLOATS;
    st4(o, z4); st4(o + 4, z4); st4(o + 8, z4);
  }
  const int blk = 4 * by + bx;
  const float u = (float)(4 * bx + row) - 7.5f, uu = u * u;
  const float vb0 = (float)(4 * by) - 7.5f;
  const int nb = (tmax + 63) >> 6;
  // batch bi (0 = the deepest) covers list positions [rel, rel + cnt); lane l <-> entry cnt - 1 - l: lane 0 is the deepest
  auto batch_rel = [&](int bi) { return (nb - 1 - bi) << 6; };
  auto fetch_gid = [&](int bi) -> int {
    if (bi >= nb) return -1;
    const int rel = batch_rel(bi), cnt = min(64, tmax - rel);
    return lane < cnt ? sorted_gid[start + rel + cnt - 1 - lane] : -1;
  };
  int gid_n = fetch_gid(0);
  float4 n0 = z4, n1 = z4, n2 = z4;
  if (gid_n >= 0) { const float* r = splats + (size_t)gid_n * TGS_SPLAT_FLOATS; n0 = ld4(r); n1 = ld4(r + 4); n2 = ld4(r + 8); }
  int gid_nn = fetch_gid(1);
  // the batch before this one, for the wave that writes its records
  size_t pP = 0; float pA = 0.f, pB = 0.f, pC = 0.f, pgx = 0.f, pgy = 0.f, popac = 1.f; bool pvalid = false;
  auto finish = [&](int bi) {                          // lane = entry of batch bi: moments -> the partial record
    if (!pvalid) return;
    const float* t = tot + ((bi & 1) * 64 + lane) * SCAN_ENT;
    const float4 t0 = ld4(t), t1 = ld4(t + 4);
    const float2 t2 = *reinterpret_cast<const float2*>(t + 8);
    const float Q0 = t1.x, Qu = t1.y, Qv = t1.z, Quu = t1.w, Quv = t2.x, Qvv = t2.y;
    const float Mx = -(pgx * Q0 - Qu), My = -(pgy * Q0 - Qv);
    const float Mxx = -(pgx * pgx * Q0 - 2.f * pgx * Qu + Quu);
    const float Mxy = -(pgx * pgy * Q0 - pgx * Qv - pgy * Qu + Quv);
    const float Myy = -(pgy * pgy * Q0 - 2.f * pgy * Qv + Qvv);
    float* o = partials + pP * TGS_PARTIAL_FLOATS;
    st4(o, make_float4(pA * Mx + pB * My, pB * Mx + pC * My, t0.w, Q0 / popac));
    st4(o + 4, make_float4(0.5f * Mxx, Mxy, 0.5f * Myy, t0.x));
    st4(o + 8, make_float4(t0.y, t0.z, 0.f, 0.f));
  };

  for (int bi = 0; bi < nb; bi++) {
    const int rel = batch_rel(bi);
    const int cnt = min(64, tmax - rel);
    const bool valid = lane < cnt;
    const int gid = gid_n;
    const float4 a0 = n0, a1 = n1, a2 = n2;
    // records of the next batch and the indices of the one after it: in flight while this batch is composited
    gid_n = gid_nn;
    if (gid_n >= 0) { const float* r = splats + (size_t)gid_n * TGS_SPLAT_FLOATS; n0 = ld4(r); n1 = ld4(r + 4); n2 = ld4(r + 8); }
    gid_nn = fetch_gid(bi + 2);
    float gx = 0.f, gy = 0.f;
    size_t P = 0;
    bool touch = false;
    if (valid) {
      centre_rel(a0, a2, tx, ty, cam.pix_center, gx, gy);
      const unsigned mask = block_mask16(gx, gy, a1.x, a1.y, a1.z, -__log2f(a0.w));
      const TileRec t = make_tile_rec<false>(a0, a1, a2, gx, gy);
      recs[lane * 3] = t.a; recs[lane * 3 + 1] = t.b; recs[lane * 3 + 2] = t.c;      // (all 16 waves store the same bits)
      touch = ((mask >> blk) & 1u) && (rel + cnt - 1 - lane) < blim;
      if (w == (bi & 15)) P = pair_index(group_base, gid, a2, tx, ty);
    }
    const bool clampy = __ballot(valid && a0.w > CLAMP_FREE_OPACITY) != 0ull;
    float* slab = acc + (bi & 1) * SCAN_ACC + w * SCAN_SLAB;
    st4(slab + lane * SCAN_ENT, z4); st4(slab + lane * SCAN_ENT + 4, z4); st4(slab + lane * SCAN_ENT + 8, z4);
    const unsigned long long bal = __ballot(touch);
    const int cntT = __popcll(bal);
    if (touch) idx[__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (unsigned char)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto chunks = [&](auto mayclamp) {
      constexpr bool MAYCLAMP = decltype(mayclamp)::value;
      for (int c0 = 0; c0 < cntT; c0 += 16) {          // 16 touching entries at a time, deepest first
        const bool has = c0 + li < cntT;
        const int src = has ? (int)idx[c0 + li] : 64;
        const float4 qa = recs[src * 3], qb = recs[src * 3 + 1], qc = recs[src * 3 + 2];
        const int pos = rel + cnt - 1 - src;
        float a[10];
#pragma unroll
        for (int i = 0; i < 10; i++) a[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (klim[k] <= rel) continue;                  // the four pixels stopped in front of this batch (state unchanged)
          const float v = vb0 + (float)k, vv = v * v, uv = u * v;
          float s = fmaf(qa.y, u, qa.x);                 // eval_s: the forward's arithmetic, operand for operand
          s = fmaf(qa.z, v, s); s = fmaf(qa.w, uu, s); s = fmaf(qb.x, uv, s); s = fmaf(qb.y, vv, s);
          const float ex = __builtin_amdgcn_exp2f(-s);
          const float al0 = MAYCLAMP ? fminf(ALPHA_MAX, ex) : ex;
          const bool go = (s <= LOG2_255) & (pos < lim[k]);
          const float al = go ? al0 : 0.f;
          const float ra = __builtin_amdgcn_rcpf(1.0f - al);        // = 1 exactly when al == 0
          const float Pm = row_scan_mul(ra * Tm[k]);     // T in front of this entry (lane 0 carries T behind the chunk)
          const float wgt = al * Pm;
          float cv = qb.w * vr[k];
          cv = fmaf(qc.x, vg[k], cv); cv = fmaf(qc.y, vb[k], cv); cv = fmaf(qb.z, vd[k], cv);
          const float z = wgt * cv;
          const float Zi = row_scan_add(z + SXs[k]);     // (lane 0 carries the sum behind the chunk)
          const float SXj = row_dpp_mov<0x111>(SXs[k], Zi);   // row_shr:1: the sum behind this entry; lane 0 keeps its own
          float q = fmaf(-(al * ra), SXj, z);
          if constexpr (MAYCLAMP) q *= fmaxf(ex * (1.0f / ALPHA_MAX), 1.0f);
          a[0] = fmaf(wgt, vr[k], a[0]); a[1] = fmaf(wgt, vg[k], a[1]); a[2] = fmaf(wgt, vb[k], a[2]); a[3] = fmaf(wgt, vd[k], a[3]);
          a[4] += q; a[5] = fmaf(q, u, a[5]); a[6] = fmaf(q, v, a[6]);
          a[7] = fmaf(q, uu, a[7]); a[8] = fmaf(q, uv, a[8]); a[9] = fmaf(q, vv, a[9]);
          // the row's totals (lane 15) back to lane 0: the state in front of this chunk
          const float Pt = row_dpp_mov<0x121>(1.f, Pm), Zt = row_dpp_mov<0x121>(0.f, Zi);    // row_ror:1
          Tm[k] = head ? Pt : 1.f;
          SXs[k] = head ? Zt : 0.f;
        }
        // the four rows (pixels) of every entry: rows of f* hold the totals of values (0, 2, 1, 3), (4, 6, 5, 7), (8, 8, 9, 9)
        const float f0 = fold16(fold32(a[0], a[1]), fold32(a[2], a[3]));
        const float f1 = fold16(fold32(a[4], a[5]), fold32(a[6], a[7]));
        const float f2x = fold32(a[8], a[9]);
        const float f2 = fold16(f2x, f2x);
        if (has) {
          float* o = slab + src * SCAN_ENT;
          const int v0 = ((row & 1) << 1) | (row >> 1);
          o[v0] = f0; o[4 + v0] = f1; o[8 + (row >> 1)] = f2;
        }
      }
    };
    if (clampy) chunks(std::true_type{}); else chunks(std::false_type{});
    __syncthreads();
    {   // entry 4 w + row of the batch: sum over the 16 blocks, lane li = block li
      const float* o = acc + (bi & 1) * SCAN_ACC + li * SCAN_SLAB + (4 * w + row) * SCAN_ENT;
      const float4 x0 = ld4(o), x1 = ld4(o + 4);
      const float2 x2 = *reinterpret_cast<const float2*>(o + 8);
      float c[10] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y};
      row_sum16x10(c);
      if (head) {
        float* t = tot + ((bi & 1) * 64 + 4 * w + row) * SCAN_ENT;
        st4(t, make_float4(c[0], c[1], c[2], c[3])); st4(t + 4, make_float4(c[4], c[5], c[6], c[7]));
        *reinterpret_cast<float2*>(t + 8) = make_float2(c[8], c[9]);
      }
    }
    if (bi > 0 && w == ((bi - 1) & 15)) finish(bi - 1);
    pP = P; pA = a1.x; pB = a1.y; pC = a1.z; pgx = gx; pgy = gy; popac = a0.w; pvalid = valid;
  }
  __syncthreads();
  if (w == ((nb - 1) & 15)) finish(nb - 1);
}

}  // namespace

// Process-wide DEFAULTS of the per-call TgsRasterOpts fields: read from the environment once (first use), settable
// through the tgs_set_* calls afterwards (A/B runs, tests).  Atomic words, relaxed: a setter racing a launch on another
// thread gives that launch the old or the new default, never a torn one.  Callers that need re-entrancy pass
// TgsRasterOpts and leave the setters alone (tgs.h).
#include <atomic>
struct RasterDefault {
  const char* env; int builtin; bool as_flag;
  std::atomic<int> v{-1};
  int get() {
    int x = v.load(std::memory_order_relaxed);
    if (x < 0) {
      const char* e = getenv(env);
      x = e ? atoi(e) : builtin;
      x = as_flag ? (x != 0) : (x < 0 ? 0 : x);
      int expect = -1;
      if (!v.compare_exchange_strong(expect, x, std::memory_order_relaxed)) x = expect;   // a setter got there first
    }
    return x;
  }
  void set(int x) { v.store(as_flag ? (x != 0) : x, std::memory_order_relaxed); }
};
static RasterDefault g_k6_blocks{"TGS_K6_BLOCKS", 1, true};
static RasterDefault g_k7_f2b{"TGS_K7_F2B", 0, true};
// TGS_K7_QUAD: frame_is_chain_bound()'s factor (default 8: deepest walk beyond 4x the balanced per-slot load); 0 = one wave per tile always
static RasterDefault g_k7_quad{"TGS_K7_QUAD", 8, false};
// (round 6, same-box sweep on the saved 720p checkpoints, profiles/r6_ab_runs.txt: min_walk 48 -> 16: K7 263 -> 240 / 253 -> 232 us; flat below)
static RasterDefault g_k7_quad_min{"TGS_K7_QUAD_MIN", 16, false};
// TGS_K6_SPLIT: tile_is_split()'s factor (default 2: lists beyond 2x the balanced per-slot load, and 256); 0 = never.  Round 6 sweep
// (profiles/r6_ab_runs.txt): 4 -> 2: K6 147 -> 108 / 134 -> 100 us on the 720p checkpoints, cfg3 unchanged (no tile qualifies),
// 1 M clustered +1 %; factor 1 costs cfg3 9 %
static RasterDefault g_k6_split{"TGS_K6_SPLIT", 2, false};
// TGS_K7_BLOCKS: 1 = the backward in 4x4-block form (k_raster_bwd_blocks); default 0 (measured slower)
static RasterDefault g_k7_blocks{"TGS_K7_BLOCKS", 0, true};
// TGS_K7_SCAN_MIN: tiles of a chain-bound frame that walk more than this many entries (among the schedule's first
// TGS_K7_SCAN_HEADS slots) go to k_raster_bwd_scan on a second stream; 0 = off
static RasterDefault g_k7_scan_min{"TGS_K7_SCAN_MIN", 0, false};
static RasterDefault g_k7_scan_heads{"TGS_K7_SCAN_HEADS", 512, false};
// TGS_K7_SCAN_SIDE: 1 (default) = the scan form's launch on an internal high-priority stream beside the four-wave launch
// (fork / join by events: also legal inside a stream capture); 0 = in line on the caller's stream.  One stream and one
// event pair per process: concurrent backward calls from several host threads must set this to 0.
static RasterDefault g_k7_scan_side{"TGS_K7_SCAN_SIDE", 1, true};
struct ScanSide { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false; };
static ScanSide& scan_side() {
  static ScanSide x = [] {
    ScanSide y;
    int lo = 0, hi = 0;
    y.ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
           hipStreamCreateWithPriority(&y.s, hipStreamNonBlocking, hi) == hipSuccess &&
           hipEventCreateWithFlags(&y.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&y.join, hipEventDisableTiming) == hipSuccess;
    return y;
  }();
  return x;
}
static inline int opt_or(const TgsRasterOpts* o, int32_t TgsRasterOpts::*f, RasterDefault& d) {
  return (o && o->*f >= 0) ? (int)(o->*f) : d.get();
}
extern "C" int tgs_set_raster_variant(int k6_blocks_on, int k7_front_to_back) {
  if (k6_blocks_on >= 0) g_k6_blocks.set(k6_blocks_on);
  if (k7_front_to_back >= 0) g_k7_f2b.set(k7_front_to_back);
  return (g_k6_blocks.get() ? 1 : 0) | (g_k7_f2b.get() ? 2 : 0);
}
// the split's shape: the shortest list it applies to (never below 64: one staged batch) and how many entries of the schedule
// get extra blocks (tgs_set_k6_split_shape; the trainer widens both for object-centric models, model.spatial_sort)
static RasterDefault g_k6_floor{"TGS_K6_FLOOR", 256, false};
static RasterDefault g_k6_heads{"TGS_K6_HEADS", 512, false};
extern "C" int tgs_set_k6_split_shape(int floor, int heads) {
  if (floor >= 0) g_k6_floor.set(floor);
  if (heads >= 0) g_k6_heads.set(heads > 65535 ? 65535 : heads);
  return g_k6_floor.get() | (g_k6_heads.get() << 16);
}
extern "C" int tgs_set_k6_split(int factor) {
  if (factor >= 0) g_k6_split.set(factor);
  return g_k6_split.get();
}
extern "C" int tgs_set_k7_scan(int min_walk, int heads) {
  if (min_walk >= 0) g_k7_scan_min.set(min_walk);
  if (heads >= 0) g_k7_scan_heads.set(heads);
  return g_k7_scan_min.get() | (g_k7_scan_heads.get() << 16);
}
#ifdef TGS_QUAD_TIMING
extern "C" int tgs_debug_quad_timing(unsigned long long* host_dst, int clear) {
  if (host_dst && hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_quad_dbg), sizeof(unsigned long long) * 8192 * 16) != hipSuccess) return -1;
  if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_quad_dbg)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 16) != hipSuccess) return -2; }
  return 0;
}
#endif
extern "C" int tgs_set_k7_quad(int factor, int min_walk) {
  if (factor >= 0) g_k7_quad.set(factor);
  if (min_walk >= 0) g_k7_quad_min.set(min_walk);
  return g_k7_quad.get() | (g_k7_quad_min.get() << 8);
}

extern "C" int tgs_rasterize_fwd(const TgsCamera* cam, const float* splats,
                                 const int32_t* sorted_gid, int32_t* tile_start, int64_t tile_start_len,
                                 const int32_t* tile_order, float* out_rgb, float* out_depth,
                                 float* final_T, int32_t* final_idx, int32_t* stop_pos, uint64_t* slot_ok,
                                 const TgsRasterOpts* opts, void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  TGS_CHECK_ARG(splats && sorted_gid && tile_start && out_rgb && out_depth && final_T,
                "null pointer");
  const CamK k = make_camk(cam);
  const int T = k.TW * k.TH;
  TGS_CHECK_ARG(tile_start_len >= (int64_t)T + 1 + TGS_TILE_START_SCRATCH,
                "tile_start buffer shorter than tgs_tile_start_len(W, H) (the forward writes the scratch behind the starts)");
  const int grid = TGS_XCDS * tgs_xcd_slots(T);
#define TGS_LAUNCH_FWD(IDX, OK)                                                                          \
  hipLaunchKernelGGL((k_raster_fwd<IDX, OK>), dim3(grid), dim3(64), 0, (hipStream_t)stream, k, T, splats, \
                     sorted_gid, tile_start, out_rgb, out_depth, final_T, final_idx, tile_order,          \
                     (unsigned long long*)slot_ok, stop_pos)
  // default: the 4x4-block form (-14.5 % at cfg3, bit-identical images); TGS_K6_BLOCKS=0 selects the quadrant
  // form, which also serves the slot_ok bitmaps.  TgsRasterOpts.k6_blocks / tgs_set_raster_variant switch it (tests, A/B).
  if (opt_or(opts, &TgsRasterOpts::k6_blocks, g_k6_blocks) && !slot_ok) {   // 4x4-block form (bit-identical images)
    // tiles with long lists are split into quadrant blocks (raster_fwd_quadrant): three extra blocks for each of the
    // schedule's first 512 entries (the longest lists of every XCD), which return at once unless their tile is split
    SplitRule sr;
    sr.factor = tile_order ? opt_or(opts, &TgsRasterOpts::k6_split, g_k6_split) : 0;
    sr.n_slots = grid;
    sr.heads = sr.factor > 0 ? min(grid, g_k6_heads.get()) : 0;
    sr.floor = max(64, g_k6_floor.get());
    const int blocks = grid + 3 * sr.heads;
    if (final_idx)
      hipLaunchKernelGGL(k_raster_fwd_blocks<true>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, k, T, splats,
                         sorted_gid, tile_start, out_rgb, out_depth, final_T, final_idx, tile_order, stop_pos, sr);
    else
      hipLaunchKernelGGL(k_raster_fwd_blocks<false>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, k, T, splats,
                         sorted_gid, tile_start, out_rgb, out_depth, final_T, final_idx, tile_order, stop_pos, sr);
    TGS_CHECK_LAUNCH();
    return TGS_OK;
  }
  if (final_idx) { if (slot_ok) TGS_LAUNCH_FWD(true, true); else TGS_LAUNCH_FWD(true, false); }
  else { if (slot_ok) TGS_LAUNCH_FWD(false, true); else TGS_LAUNCH_FWD(false, false); }
#undef TGS_LAUNCH_FWD
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

static int rasterize_bwd_impl(const TgsCamera* cam, const float* splats,
                                 const int32_t* group_base, const int32_t* sorted_gid,
                                 int32_t* tile_start, int64_t tile_start_len, const int32_t* tile_order,
                                 const float* out_rgb, const float* out_depth,
                                 const float* final_T, const int32_t* stop_pos, const float* v_rgb,
                                 const float* v_depth, const float* v_alpha,
                                 const TgsLossSpec* loss, float* partials, float* tile_loss,
                                 int band, const uint64_t* slot_ok, const TgsRasterOpts* opts, void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  TGS_CHECK_ARG(splats && group_base && sorted_gid && tile_start && out_rgb && out_depth &&
                final_T && partials, "null pointer");
  LossK lk;
  lk.on = 0; lk.gt_rgb = nullptr; lk.gt_depth = nullptr; lk.unc = nullptr;
  lk.l1w = lk.dw = lk.uw = 0.f; lk.eps = 1e-6f;
  if (loss) {
    lk.on = 1;
    lk.gt_rgb = (loss->l1_weight != 0.f) ? loss->gt_rgb : nullptr;
    lk.gt_depth = (loss->depth_weight != 0.f) ? loss->gt_depth : nullptr;
    lk.unc = loss->uncertainty;
    lk.l1w = loss->l1_weight; lk.dw = loss->depth_weight;
    lk.uw = loss->uncertainty_weight; lk.eps = loss->eps;
  }
  const CamK k = make_camk(cam);
  const int T = k.TW * k.TH;
  TGS_CHECK_ARG(tile_start_len >= (int64_t)T + 1 + TGS_TILE_START_SCRATCH,
                "tile_start buffer shorter than tgs_tile_start_len(W, H) (the backward keeps slot counters behind the starts)");
  int grid = TGS_XCDS * tgs_xcd_slots(T);
  if (band >= 0) {   // one image band: the blocks of chunk `band` of every XCD's slots
    TGS_CHECK_ARG(tile_order, "a band launch needs the tile_order of tgs_bin_sort");
    TGS_CHECK_ARG(band < tgs_band_count(T), "band out of range");
    const int c = tgs_band_slots(T), s0 = band * c, s1 = min(s0 + c, tgs_xcd_slots(T));
    tile_order += (size_t)s0 * TGS_XCDS;
    grid = (s1 - s0) * TGS_XCDS;
  }
  if (opt_or(opts, &TgsRasterOpts::k7_front_to_back, g_k7_f2b) || slot_ok) {   // the front-to-back form (round 1-3; kept for A/B and for the slot_ok bitmaps)
    hipLaunchKernelGGL(k_raster_bwd_f2b, dim3(grid), dim3(64), 0, (hipStream_t)stream, k, T, splats,
                       group_base, sorted_gid, tile_start, out_rgb, out_depth, final_T,
                       v_rgb, v_depth, v_alpha, lk, partials, tile_loss, tile_order,
                       (const unsigned long long*)slot_ok);
  } else if (opt_or(opts, &TgsRasterOpts::k7_blocks, g_k7_blocks)) {   // 4x4-block form (round 6: measured, not the default)
    TGS_CHECK_ARG(stop_pos, "stop_pos (written by tgs_rasterize_fwd) is required");
    hipLaunchKernelGGL(k_raster_bwd_blocks, dim3(grid), dim3(64), 0, (hipStream_t)stream, k, T, splats,
                       group_base, sorted_gid, tile_start, out_rgb, out_depth, final_T, stop_pos,
                       v_rgb, v_depth, v_alpha, lk, partials, tile_loss, tile_order);
  } else {
    TGS_CHECK_ARG(stop_pos, "stop_pos (written by tgs_rasterize_fwd) is required");
    QuadRule qf;
    qf.factor = band >= 0 ? 0 : opt_or(opts, &TgsRasterOpts::k7_quad, g_k7_quad);   // (a band launch keeps one wave per tile)
    qf.min_walk = opt_or(opts, &TgsRasterOpts::k7_quad_min_walk, g_k7_quad_min);
    qf.scan_min = qf.factor > 0 ? g_k7_scan_min.get() : 0;
    qf.scan_heads = min(grid, g_k7_scan_heads.get());
    if (qf.scan_min > 0 && qf.scan_min < qf.min_walk) qf.scan_min = qf.min_walk;
    // The scan form's launch runs BESIDE the other two (all three only read the frame and write disjoint records; the
    // scan form needs nothing of k_raster_bwd's): forked FIRST, so that its workgroups (108 KB of LDS, 16 waves: a CU
    // of their own, more or less) are resident before the four-wave form's persistent grid fills every CU -- launched
    // behind it they would only find room when that grid retires, i.e. run in series.  Joined before returning.
    ScanSide* side = nullptr;
    if (qf.scan_min > 0 && qf.scan_heads > 0) {
      static const hipError_t scan_attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_raster_bwd_scan),
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_LDS_BYTES);
      TGS_CHECK_ARG(scan_attr == hipSuccess, "k_raster_bwd_scan: dynamic LDS size refused");
      hipStream_t ss = (hipStream_t)stream;
      if (g_k7_scan_side.get()) {
        side = &scan_side();
        TGS_CHECK_ARG(side->ok, "k_raster_bwd_scan: no side stream");
        TGS_HIP(hipEventRecord(side->fork, (hipStream_t)stream));
        TGS_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
        ss = side->s;
      }
      hipLaunchKernelGGL(k_raster_bwd_scan, dim3(qf.scan_heads), dim3(1024), SCAN_LDS_BYTES, ss, k, T, splats,
                         group_base, sorted_gid, tile_start, out_rgb, out_depth, final_T, stop_pos,
                         v_rgb, v_depth, v_alpha, lk, partials, tile_order, qf);
      if (side) TGS_HIP(hipEventRecord(side->join, side->s));
    }
    hipLaunchKernelGGL(k_raster_bwd, dim3(grid), dim3(64), 0, (hipStream_t)stream, k, T, splats,
                       group_base, sorted_gid, tile_start, out_rgb, out_depth, final_T, stop_pos,
                       v_rgb, v_depth, v_alpha, lk, partials, tile_loss, tile_order, qf);
    if (qf.factor > 0) {
      TGS_CHECK_LAUNCH();
      // 5 workgroups of 28 KB LDS fit a CU: 1280 resident workgroups walk the schedule
      hipLaunchKernelGGL(k_raster_bwd_quad, dim3(min(grid, 1280)), dim3(256), 0, (hipStream_t)stream, k, T, splats,
                         group_base, sorted_gid, tile_start, out_rgb, out_depth, final_T, stop_pos,
                         v_rgb, v_depth, v_alpha, lk, partials, tile_order, qf, grid);
    }
    if (side) TGS_HIP(hipStreamWaitEvent((hipStream_t)stream, side->join, 0));
  }
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_rasterize_bwd(const TgsCamera* cam, const float* splats,
                                 const int32_t* group_base, const int32_t* sorted_gid,
                                 int32_t* tile_start, int64_t tile_start_len, const int32_t* tile_order,
                                 const float* out_rgb, const float* out_depth,
                                 const float* final_T, const int32_t* stop_pos, const float* v_rgb,
                                 const float* v_depth, const float* v_alpha,
                                 const TgsLossSpec* loss, float* partials, float* tile_loss,
                                 const uint64_t* slot_ok, const TgsRasterOpts* opts, void* stream) {
  return rasterize_bwd_impl(cam, splats, group_base, sorted_gid, tile_start, tile_start_len, tile_order, out_rgb, out_depth, final_T, stop_pos, v_rgb, v_depth, v_alpha, loss, partials, tile_loss, -1, slot_ok, opts, stream);
}

extern "C" int tgs_rasterize_bwd_band(const TgsCamera* cam, const float* splats,
                                 const int32_t* group_base, const int32_t* sorted_gid,
                                 int32_t* tile_start, int64_t tile_start_len, const int32_t* tile_order,
                                 const float* out_rgb, const float* out_depth,
                                 const float* final_T, const int32_t* stop_pos, const float* v_rgb,
                                 const float* v_depth, const float* v_alpha,
                                 const TgsLossSpec* loss, float* partials, float* tile_loss,
                                 int band, const uint64_t* slot_ok, const TgsRasterOpts* opts, void* stream) {
  TGS_CHECK_ARG(band >= 0, "band < 0");
  return rasterize_bwd_impl(cam, splats, group_base, sorted_gid, tile_start, tile_start_len, tile_order, out_rgb, out_depth, final_T, stop_pos, v_rgb, v_depth, v_alpha, loss, partials, tile_loss, band, slot_ok, opts, stream);
}

extern "C" size_t tgs_slot_ok_len(int W, int H, int64_t capacity) {
  if (capacity < 0) capacity = 0;
  const size_t T = (size_t)((W + TGS_BLOCK - 1) / TGS_BLOCK) * (size_t)((H + TGS_BLOCK - 1) / TGS_BLOCK);
  return (size_t)(capacity >> 6) + T + 2;
}
