// tgs_common.h -- shared host/device helpers for libtgs_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tgs.h"

#define TGS_WAVE 64

// K6 / K7 schedule: block b of the launch runs on XCD b % 8 (observed dispatch order).  The tiles (row
// major) are dealt to the XCDs in granules of TGS_XCD_GRANULE consecutive tiles: granule q belongs to
// XCD q % 8.  Slot i of XCD x (i = b / 8) is tile ((i / G) * 8 + x) * G + i % G; slots past the last
// tile hold none.  Fine interleaving balances the XCDs for any view -- one contiguous image band per
// XCD left the XCDs of the emptier bands idle (K7 -12 %, K6 -12 % averaged over the 8 orbit views of
// cfg3, tools/order_probe.py) -- and granules keep horizontal neighbours (which share most of their
// Gaussians) in one L2; whole tile rows per XCD (68 rows over 8 XCDs: 9 against 8) were 6 % off
// balance.  Granule sweep at cfg3 (same box, ms per step): 4: 1.066, 8: 1.061, 16: 1.074, 32: 1.075, 64: 1.075.
#define TGS_XCDS 8
#ifndef TGS_XCD_GRANULE
#define TGS_XCD_GRANULE 8
#endif
// The rasterizer's scratch behind the T + 1 tile starts (tgs_tile_start_len = T + 1 + 512 ints): per XCD x the frame's
// deepest walk and the sum of its walks (K6 publish_walk -> K7 frame_is_chain_bound) and a slot counter of K7's
// four-wave launch.  One XCD's words sit 256 B from the next one's, walk words and slot counter 128 B apart -- eight
// L2s updating neighbouring words of ONE line pass the line around for every atomic.  Zeroed by k_scan_tiles.
#define TGS_WALK_WORDS 8
#define TGS_WALK_AT(T, x) ((T) + 1 + 64 * (x))
#define TGS_WALKSUM_AT(T, x) ((T) + 1 + 64 * (x) + 1)
#define TGS_SLOTCTR_AT(T, x) ((T) + 1 + 64 * (x) + 32)
#define TGS_TILE_START_SCRATCH 512
static inline __host__ __device__ int tgs_xcd_slots(int T) {
  const int q = (T + TGS_XCD_GRANULE - 1) / TGS_XCD_GRANULE;
  return ((q + TGS_XCDS - 1) / TGS_XCDS) * TGS_XCD_GRANULE;
}
// An XCD's slots are cut into chunks of <= 1024 slots, a multiple of the granule (one register sort
// builds a chunk's visiting order, longest list first).  Chunk c of all XCDs together covers the
// row-major tile range [8 c chunk, 8 (c+1) chunk), an "image band" that tgs_rasterize_bwd_band can
// launch on its own.  TGS_BANDS = minimum number of chunks: 1 -- finer chunks (4 per XCD at 1080p)
// cost K6 / K7 6 % (longest-first only inside 256 slots).
#ifndef TGS_BANDS
#define TGS_BANDS 1
#endif
static inline __host__ __device__ int tgs_band_slots(int T) {
  const int per = tgs_xcd_slots(T);
  int n = (per + 1023) / 1024;
  if (n < TGS_BANDS) n = TGS_BANDS;
  const int c = (per + n - 1) / n;
  return ((c + TGS_XCD_GRANULE - 1) / TGS_XCD_GRANULE) * TGS_XCD_GRANULE;
}
static inline __host__ __device__ int tgs_band_count(int T) {
  const int c = tgs_band_slots(T);
  return (tgs_xcd_slots(T) + c - 1) / c;
}
static inline __host__ __device__ int tgs_xcd_slot_tile(int T, int x, int i) {
  const int k = i / TGS_XCD_GRANULE;
  const int tile = (k * TGS_XCDS + x) * TGS_XCD_GRANULE + (i - k * TGS_XCD_GRANULE);
  return tile < T ? tile : T;      // T = no tile
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void tgs_set_error(const char* fmt, ...);

#define TGS_CHECK_ARG(cond, msg)                       \
  do {                                                 \
    if (!(cond)) {                                     \
      tgs_set_error("%s: %s", __func__, msg);          \
      return TGS_E_ARG;                                \
    }                                                  \
  } while (0)

#define TGS_CHECK_LAUNCH()                                                     \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      tgs_set_error("%s: HIP launch failed: %s", __func__, hipGetErrorString(e_)); \
      return TGS_E_HIP;                                                        \
    }                                                                          \
  } while (0)

#define TGS_HIP(call)                                                          \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      tgs_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_)); \
      return TGS_E_HIP;                                                        \
    }                                                                          \
  } while (0)

// Kernel-side camera block (passed by value in kernarg -> lives in SGPRs).
struct CamK {
  float R[9];      // world->cam rotation, row-major
  float t[3];      // world->cam translation
  float campos[3]; // -R^T t
  float fx, fy, cx, cy;
  float limx, limy; // 1.3 * tan(fov/2)
  float near_plane, pix_center, glob_scale;
  float bg[3];
  int W, H, TW, TH;
  int long_run;    // binning: runs of more tiles than this are "long" (tgs_set_long_run; not a camera property -- it rides here because every kernel of the front half and K8 takes the block)
};

extern "C" int tgs_set_long_run(int tiles);
static inline CamK make_camk(const TgsCamera* c) {
  CamK k;
  k.long_run = tgs_set_long_run(-1);
  const float* V = c->viewmat;
  for (int r = 0; r < 3; r++) {
    for (int j = 0; j < 3; j++) k.R[3 * r + j] = V[4 * r + j];
    k.t[r] = V[4 * r + 3];
  }
  for (int j = 0; j < 3; j++)
    k.campos[j] = -(V[0 + j] * V[3] + V[4 + j] * V[7] + V[8 + j] * V[11]);
  k.fx = c->fx; k.fy = c->fy; k.cx = c->cx; k.cy = c->cy;
  k.limx = 1.3f * (0.5f * (float)c->W) / c->fx;
  k.limy = 1.3f * (0.5f * (float)c->H) / c->fy;
  k.near_plane = c->near_plane; k.pix_center = c->pix_center; k.glob_scale = c->glob_scale;
  for (int j = 0; j < 3; j++) k.bg[j] = c->bg[j];
  k.W = c->W; k.H = c->H;
  k.TW = (c->W + TGS_BLOCK - 1) / TGS_BLOCK;
  k.TH = (c->H + TGS_BLOCK - 1) / TGS_BLOCK;
  return k;
}

static inline bool camera_ok(const TgsCamera* c) {
  return c && c->W > 0 && c->H > 0 && c->fx > 0.f && c->fy > 0.f;
}

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

// App. B.4 tile rectangle (the normative 3-sigma rule).
__device__ __forceinline__ void tile_rect(float u, float v, int radius, int TW, int TH,
                                          int& x0, int& y0, int& x1, int& y1) {
  const float r = (float)radius;
  const float inv = 1.0f / (float)TGS_BLOCK;  // exact (power of two)
  x0 = min(max((int)((u - r) * inv), 0), TW);
  x1 = min(max((int)((u + r) * inv) + 1, 0), TW);
  y0 = min(max((int)((v - r) * inv), 0), TH);
  y1 = min(max((int)((v + r) * inv) + 1, 0), TH);
}

// Splat slot 10 holds the Gaussian's tile rectangle packed as x0 | y0<<8 | w<<16 | h<<24
// (image sides are limited to 255 tiles = 4080 px).  It is the B.4 rect intersected with the
// tiles that contain at least one pixel where alpha can reach 1/255 (output preserving).
__device__ __forceinline__ unsigned pack_rect(int x0, int y0, int w, int h) {
  return (unsigned)x0 | ((unsigned)y0 << 8) | ((unsigned)w << 16) | ((unsigned)h << 24);
}
__device__ __forceinline__ void unpack_rect(unsigned r, int& x0, int& y0, int& w, int& h) {
  x0 = r & 255u; y0 = (r >> 8) & 255u; w = (r >> 16) & 255u; h = r >> 24;
}

// wave64 sum via DPP (no LDS, no ds_bpermute).  Total lands in lane 63; the helper returns it
// wave-uniformly (SGPR) through v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  int s = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, s);
}

__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror      -> every lane holds its row's sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = total
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = wave_sum_to_lane63(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// wave64 integer min / max through DPP (lanes a DPP step does not reach combine with their own value,
// which is neutral); the result is wave-uniform.
template <int CTRL, int ROW_MASK, bool MAX>
__device__ __forceinline__ int dpp_minmax(int v) {
  const int s = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
  return MAX ? max(v, s) : min(v, s);
}
template <bool MAX>
__device__ __forceinline__ int wave_minmax_i(int v) {
  v = dpp_minmax<0xB1, 0xf, MAX>(v);   // quad_perm [1,0,3,2]
  v = dpp_minmax<0x4E, 0xf, MAX>(v);   // quad_perm [2,3,0,1]
  v = dpp_minmax<0x141, 0xf, MAX>(v);  // row_half_mirror
  v = dpp_minmax<0x140, 0xf, MAX>(v);  // row_mirror      -> every lane holds its row's result
  v = dpp_minmax<0x142, 0xa, MAX>(v);  // row_bcast:15 into rows 1,3
  v = dpp_minmax<0x143, 0xc, MAX>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = result
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming (non-temporal) 16-byte accesses for data touched exactly once per launch
typedef float tgs_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  const tgs_f4v v = __builtin_nontemporal_load(reinterpret_cast<const tgs_f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  tgs_f4v t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<tgs_f4v*>(p));
}

#endif  // __HIPCC__
