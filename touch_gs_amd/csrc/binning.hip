// binning.hip -- K2-K5: tile intersection counting, tile-offset scan, bin fill and per-tile
// depth sort.  gfx950, wave64.
//
// Spec: SURVEY.md App. B.4 (tile rect) and B.6 (order = tile, then depth bits, ties by Gaussian id).
// The CUDA implementations behind the reference's `rasterize_gaussians` (gsplat
// map_gaussian_to_intersects + cub::DeviceRadixSort + get_tile_bin_edges, App. A.2) globally
// radix-sort 64-bit (tile|depth) keys: ~6 passes x 24 B per intersection.  Here the tile part of
// the key is resolved by a counting sort (one integer atomic per intersection, which also yields
// the tile bin edges for free) and the depth part by a per-tile bitonic sort held entirely in the
// 160 KB LDS of one CU, so an intersection costs 4 B (rank) + 8 B (pair) written once, read once,
// + 4 B sorted id.  Integer atomics only: the output is bit-reproducible.
//
// Work decomposition: a "group" is 256 consecutive Gaussians.  Its intersections occupy one
// contiguous range [group_base, group_base+total) of the pair index space, so every pass is
// balanced (one thread per intersection, found by binary search in the group's LDS-resident scan)
// no matter how many tiles a single Gaussian covers.
#include "tgs_binning.h"

namespace {

// Loads the group's rects from the splat records and builds the in-group scan.
__device__ __forceinline__ int group_load_scan(const CamK& cam, int N,
                                               const float* __restrict__ splats, GroupScan& S,
                                               int& my_off) {
  const int g = blockIdx.x * TGS_GROUP + threadIdx.x;
  int hits = 0, x0 = 0, y0 = 0, w = 0;
  unsigned dbits = 0;
  if (g < N) {
    const float* rec = splats + (size_t)g * TGS_SPLAT_FLOATS;
    int h;
    unpack_rect(__float_as_uint(rec[10]), x0, y0, w, h);
    hits = w * h;
    dbits = __float_as_uint(rec[2]);
  }
  return group_scan_store(S, hits, x0, y0, w, dbits, my_off);
}

// K3a: count intersections per tile; allocate the group's pair range; remember every pair's
// arrival rank inside its tile.
__global__ __launch_bounds__(TGS_GROUP) void k_tile_count(
    CamK cam, int N, float* __restrict__ splats, int32_t* __restrict__ group_base,
    int32_t* __restrict__ tile_count, int32_t* __restrict__ rank, int32_t* __restrict__ status,
    long long capacity, int32_t* __restrict__ sticky) {
  __shared__ GroupScan S;
  int my_off;
  const int total = group_load_scan(cam, N, splats, S, my_off);
  const int g = blockIdx.x * TGS_GROUP + threadIdx.x;
  if (g < N) splats[(size_t)g * TGS_SPLAT_FLOATS + 11] = __int_as_float(my_off);
  group_count_tiles(S, cam.TW, total, group_base, tile_count, rank, status, capacity, sticky);
}

// K4: exclusive scan of tile counts -> tile_start[T+1] (single workgroup; T is a few 10^4).
// Thread t owns the contiguous chunk [t*C, (t+1)*C), C = ceil(T/1024): one serial pass over its
// chunk, one workgroup scan of the 1024 chunk totals, one pass to write -- two barriers in all.
// On capacity overflow every list is made empty so that downstream kernels touch nothing.
//
// Blocks 1..8 (launched only when a tile_order buffer is given) build the visiting order of the
// compositing kernels: block b of K6/K7 runs on XCD b % 8 and owns tile tile_order[b]; every XCD keeps
// its contiguous band of tiles (so the splat records neighbouring tiles share stay in its L2) but
// visits it longest list first.  With ~2 tiles per resident wave slot the spatial order leaves the
// last slots running alone for a whole tile; longest-first shortens that tail (K6 -3 %, K7 -4 %,
// alternating same-box runs after clock warm-up).
__global__ __launch_bounds__(1024) void k_scan_tiles(int T, const int32_t* __restrict__ tile_count,
                                                     int32_t* __restrict__ tile_start,
                                                     const int32_t* __restrict__ status,
                                                     int32_t* __restrict__ tile_order) {
  __shared__ int wave_tot[16];
  __shared__ unsigned long long okeys[8192];   // one XCD band: <= ceil(255*255/8) = 8129 tiles
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {
    const int x = blockIdx.x - 1;
    const int per = (T + 7) >> 3;
    const int t0 = x * per, len = max(0, min(per, T - t0));
    int np2 = 2;
    while (np2 < per) np2 <<= 1;
    // ascending on (~n, tile) = descending list length, ties by tile id; pads sort to the end
    for (int i = tid; i < np2; i += 1024)
      okeys[i] = i < len ? ((unsigned long long)(~(unsigned)tile_count[t0 + i]) << 32) | (unsigned)(t0 + i)
                         : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < (np2 >> 1); i += 1024) {
          const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
          const int hi = lo | j;
          const bool up = (lo & k) == 0;
          const unsigned long long a = okeys[lo], b = okeys[hi];
          if ((a > b) == up) { okeys[lo] = b; okeys[hi] = a; }
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < per; i += 1024)
      tile_order[i * 8 + x] = i < len ? (int)(okeys[i] & 0xffffffffull) : T;   // T = no tile
    return;
  }
  const bool overflow = status[1] != 0;
  const int C = (T + 1023) / 1024;
  const int lo = min(tid * C, T), hi = min(lo + C, T);
  int sum = 0;
  if (!overflow)
    for (int i = lo; i < hi; i++) sum += tile_count[i];
  int incl = sum;
#pragma unroll
  for (int o = 1; o < TGS_WAVE; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if ((tid & 63) >= o) incl += t;
  }
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  int wbase = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    if (w < (tid >> 6)) wbase += wave_tot[w];
    tot += wave_tot[w];
  }
  int run = wbase + incl - sum;
  for (int i = lo; i < hi; i++) {
    tile_start[i] = run;
    run += overflow ? 0 : tile_count[i];
  }
  if (tid == 0) tile_start[T] = tot;
}

// K3b: scatter (gid, depth bits) into the tile bins.  slot = tile_start + arrival rank.
__global__ __launch_bounds__(TGS_GROUP) void k_fill_bins(
    CamK cam, int N, const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const int32_t* __restrict__ tile_start, const int32_t* __restrict__ rank,
    uint2* __restrict__ pairs, const int32_t* __restrict__ status) {
  if (status[1]) return;
  __shared__ GroupScan S;
  int my_off;
  const int total = group_load_scan(cam, N, splats, S, my_off);
  __syncthreads();
  const long long base = group_base[blockIdx.x];
  const int g0 = blockIdx.x * TGS_GROUP;
  for (int i = threadIdx.x; i < total; i += TGS_GROUP) {
    int j, tile;
    group_pair(S, cam.TW, i, j, tile);
    const int slot = tile_start[tile] + rank[base + i];
    pairs[slot] = make_uint2((unsigned)(g0 + j), S.depth_bits[j]);  // u64 = depth<<32 | gid
  }
}

__device__ __forceinline__ int next_pow2(int n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}

// K5: per-tile bitonic sort of (depth bits, gid) in LDS; writes the sorted Gaussian ids.
// One workgroup per tile; tiles whose list length is outside (LO, CAP] belong to another class.
template <int CAP, int THREADS, int LO>
__global__ __launch_bounds__(THREADS) void k_sort_tiles_lds(
    int T, const int32_t* __restrict__ tile_start, const unsigned long long* __restrict__ pairs,
    int32_t* __restrict__ sorted_gid) {
  __shared__ unsigned long long keys[CAP];
  const int tid = threadIdx.x;
  // grid-stride over tiles: the rare long-list classes are launched with a small grid so that an
  // empty class costs ~2 us instead of a full grid of 128-KB-LDS workgroups
  for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
  const int s = tile_start[tile];
  const int n = tile_start[tile + 1] - s;
  if (n <= LO || n > CAP) continue;
  if (n == 1) {
    if (tid == 0) sorted_gid[s] = (int)(pairs[s] & 0xffffffffull);
    continue;
  }
  __syncthreads();  // keys[] reuse across iterations
  const int np2 = next_pow2(n);
  for (int i = tid; i < np2; i += THREADS) keys[i] = (i < n) ? pairs[s + i] : ~0ull;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < (np2 >> 1); i += THREADS) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n; i += THREADS) sorted_gid[s + i] = (int)(keys[i] & 0xffffffffull);
  }
}

// Rare long lists: one launch covers both remaining classes.  (LO, CAP] sorts in 128 KB of LDS like
// the common class; anything longer runs the same network in global memory on a power-of-two
// padded copy at fb[2*s ...) (next_pow2(n) < 2n, so per-tile regions never overlap).
template <int CAP, int LO>
__global__ __launch_bounds__(1024) void k_sort_tiles_long(
    int T, const int32_t* __restrict__ tile_start, const unsigned long long* __restrict__ pairs,
    unsigned long long* __restrict__ fb, int32_t* __restrict__ sorted_gid) {
  __shared__ unsigned long long lds_keys[CAP];
  const int tid = threadIdx.x;
  for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const int s = tile_start[tile];
    const int n = tile_start[tile + 1] - s;
    if (n <= LO) continue;
    const int np2 = next_pow2(n);
    const bool in_lds = n <= CAP;
    unsigned long long* keys = in_lds ? lds_keys : fb + 2 * (size_t)s;
    __syncthreads();  // lds_keys reuse across iterations
    for (int i = tid; i < np2; i += 1024) keys[i] = (i < n) ? pairs[s + i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < (np2 >> 1); i += 1024) {
          const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
          const int hi = lo | j;
          const bool up = (lo & k) == 0;
          const unsigned long long a = keys[lo], b = keys[hi];
          if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
        }
        if (!in_lds) __threadfence_block();
        __syncthreads();
      }
    }
    for (int i = tid; i < n; i += 1024) sorted_gid[s + i] = (int)(keys[i] & 0xffffffffull);
  }
}

constexpr int SORT_CAP_A = 2048;   // 16 KB LDS, 256 threads
constexpr int SORT_CAP_B = 16384;  // 128 KB LDS, 1024 threads; longer lists sort in global memory

}  // namespace

extern "C" int tgs_num_groups(int N) { return (N + TGS_GROUP - 1) / TGS_GROUP; }
extern "C" int tgs_num_tiles(int W, int H) {
  return ((W + TGS_BLOCK - 1) / TGS_BLOCK) * ((H + TGS_BLOCK - 1) / TGS_BLOCK);
}
extern "C" int tgs_tile_order_len(int W, int H) { return ((tgs_num_tiles(W, H) + 7) / 8) * 8; }
// scratch layout: pairs u64[cap] | fallback u64[2*cap] | rank i32[cap]
extern "C" size_t tgs_sort_scratch_bytes(int64_t capacity) {
  if (capacity < 0) capacity = 0;
  return (size_t)capacity * (8 + 16 + 4) + 64;
}

int tgs_bin_finish(const CamK& k, int N, const float* splats, const int32_t* group_base,
                   int32_t* tile_start, int32_t* tile_cursor, int32_t* sorted_gid,
                   int32_t* tile_order, int64_t capacity, void* scratch, int32_t* status,
                   hipStream_t s) {
  const int T = k.TW * k.TH;
  const int G = tgs_num_groups(N);
  const BinScratch sc = carve_scratch(scratch, capacity);
  hipLaunchKernelGGL(k_scan_tiles, dim3(tile_order ? 9 : 1), dim3(1024), 0, s, T, tile_cursor, tile_start,
                     status, tile_order);
  TGS_CHECK_LAUNCH();
  if (G > 0) {
    hipLaunchKernelGGL(k_fill_bins, dim3(G), dim3(TGS_GROUP), 0, s, k, N, splats, group_base,
                       tile_start, sc.rank, (uint2*)sc.pairs, status);
    TGS_CHECK_LAUNCH();
    const int small_grid = T < 256 ? T : 256;
    hipLaunchKernelGGL((k_sort_tiles_lds<SORT_CAP_A, 256, 0>), dim3(T), dim3(256), 0, s, T,
                       tile_start, sc.pairs, sorted_gid);
    TGS_CHECK_LAUNCH();
    hipLaunchKernelGGL((k_sort_tiles_long<SORT_CAP_B, SORT_CAP_A>), dim3(small_grid), dim3(1024), 0,
                       s, T, tile_start, sc.pairs, sc.fb, sorted_gid);
    TGS_CHECK_LAUNCH();
  }
  return TGS_OK;
}

extern "C" int tgs_bin_sort(const TgsCamera* cam, int N, float* splats, int32_t* group_base,
                            int32_t* tile_start, int32_t* tile_cursor, int32_t* sorted_gid,
                            int32_t* tile_order, int64_t capacity, void* scratch, int32_t* status,
                            int32_t* sticky_overflow, void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  TGS_CHECK_ARG(N >= 0 && capacity >= 0, "negative size");
  TGS_CHECK_ARG(capacity < (1ll << 31), "capacity must be < 2^31");
  TGS_CHECK_ARG(group_base && tile_start && tile_cursor && sorted_gid && scratch && status,
                "null pointer");
  TGS_CHECK_ARG(N == 0 || splats, "null splats");
  const CamK k = make_camk(cam);
  const int T = k.TW * k.TH;
  const int G = tgs_num_groups(N);
  hipStream_t s = (hipStream_t)stream;
  const BinScratch sc = carve_scratch(scratch, capacity);
  hipLaunchKernelGGL(k_clear_counters, dim3((max(T, 2) + 255) / 256), dim3(256), 0, s, tile_cursor, T, status,
                     sticky_overflow);
  TGS_CHECK_LAUNCH();
  if (G > 0) {
    hipLaunchKernelGGL(k_tile_count, dim3(G), dim3(TGS_GROUP), 0, s, k, N, splats, group_base,
                       tile_cursor, sc.rank, status, (long long)capacity, sticky_overflow);
    TGS_CHECK_LAUNCH();
  }
  return tgs_bin_finish(k, N, splats, group_base, tile_start, tile_cursor, sorted_gid, tile_order,
                        capacity, scratch, status, s);
}
