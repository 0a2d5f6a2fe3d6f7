// tgs_binning.h -- pieces of the tile-binning pass shared by binning.hip and the fused
// projection+count kernel in project.hip.
#pragma once
#include "tgs_common.h"

// scratch layout of tgs_bin_sort: pairs u64[cap] | fallback u64[2*cap] | rank i32[cap]
struct BinScratch {
  unsigned long long* pairs;
  unsigned long long* fb;
  int32_t* rank;
};
static inline BinScratch carve_scratch(void* scratch, int64_t capacity) {
  BinScratch b;
  b.pairs = (unsigned long long*)scratch;
  b.fb = b.pairs + capacity;
  b.rank = (int32_t*)(b.fb + 2 * capacity);
  return b;
}

// scan + fill + sort (everything after the per-tile counts exist); defined in binning.hip.
// `front` (tgs_project_bin_sort_front only): the tag word of the fused optimizer kernel that ran this frame's K1 -- a
// mismatch voids the frame -- and the counters / status word of the NEXT frame, cleared on the side by the scan launch.
struct BinFront {
  const int32_t* tag_word;
  int32_t tag_expect;
  int32_t* next_tile_cursor;   // may be NULL
  int32_t* next_status;
  int next_T;
};
int tgs_bin_finish(const CamK& k, int N, const float* splats, const int32_t* group_base,
                   int32_t* tile_start, int64_t tile_start_len, int32_t* tile_cursor, int32_t* sorted_gid,
                   int32_t* tile_order, int64_t capacity, void* scratch, int32_t* status,
                   int32_t* sticky_overflow, int32_t max_list_hint, const BinFront* front, hipStream_t s);

// Per-tile intersection counters are kept once per XCD (row x of an [8][T] array, x = the XCC the
// counting workgroup runs on): the 8 L2s of an MI355X are kept coherent by ownership migration, so a
// counter line that all XCDs increment bounces between them -- with private rows every atomic stays
// in its own L2 (and the hot tiles of an object-centric scene are spread over 8 lines instead of 1).
// A tile's list is the concatenation of its 8 sub-lists (the per-tile sort orders it anyway).
// counter buffer layout: count[8][T] | sub_start[8][T] | scan aggregates[64] | longest list[1] (+pad)
#define TGS_XCC 8
#define TGS_SCAN_WGS 64
// ... | pair allocators: one 128-B line per XCD holding {allocated pairs + overshoot, pairs that found no room, overshoot of failed attempts}.  The pair index space [0, capacity) is cut into one region per XCD and a
// group takes its contiguous pair range from the region of the XCD it runs on; if that region is
// full (which XCD runs which workgroup is not deterministic, and with few groups the shares are very
// uneven) it takes the range from the first other region with room.  An allocation can therefore
// only fail when  #pairs + 8 x (largest group total) > capacity  -- the bound status[2] reports
// (k_scan_tiles reads the group totals off the records of each group's last Gaussian: a per-group
// atomicMax on the allocator lines cost the front half 25 us at cfg3, memory-side atomics on one line
// serialise at ~13 ns each).
#define TGS_ALLOC_STRIDE 32
#define TGS_ALLOC_OFF(T) ((2 * TGS_XCC * (T) + TGS_SCAN_WGS + 4 + 31) / 32 * 32)
// ... | largest group total: one word per XCD, each in a 128-B line of its own behind the allocator lines (round 5).  Every
// K1 / count workgroup folds its group's pair total in with a fire-and-forget atomicMax; k_scan_tiles takes the maximum of
// the 8 words for status[2] = #pairs + 8 x the largest group.  (The scan's status workgroup used to read the total off the
// last record of each of the N / 256 groups: four dependent scattered loads per thread on the scan launch's critical path.
// On lines of their own the 3906 atomics of a cfg3 frame do not sit behind the allocators' returning atomics.)
#ifndef TGS_MG_ATOMIC
#define TGS_MG_ATOMIC 1
#endif
#define TGS_MAXG_OFF(T) (TGS_ALLOC_OFF(T) + TGS_XCC * TGS_ALLOC_STRIDE)
static inline int tgs_counter_len(int T) { return TGS_MAXG_OFF(T) + TGS_XCC * TGS_ALLOC_STRIDE; }

// Aggregated counting (spatially ordered parameter buffers): when the tile bounding box of a group's
// 256 Gaussians holds at most TGS_AGG_TILES tiles, its pairs are first counted per tile in an LDS
// histogram (integer LDS atomics: 5.6 cycles per wave instruction, tools/ubench/lds_atomic.hip) and
// the group then issues ONE returning global atomic per touched tile instead of one per pair.  With
// Morton-ordered Gaussians a group touches ~50 tiles with ~20 pairs each; a group of a randomly
// ordered buffer spans the whole image and takes the direct path.
#define TGS_AGG_TILES 1024
#define TGS_AGG_U 4            // pairs per thread kept in registers by the aggregated path (256 * 4 per group)
#ifndef TGS_LONG_RUN
#define TGS_LONG_RUN 32        // DEFAULT of CamK.long_run (tgs_set_long_run): runs of more tiles than this stay outside the group's box (here) and are summed by the workgroup (K8)
#endif
#ifndef TGS_DIRECT_U
#define TGS_DIRECT_U 4         // pairs per thread and round of the direct counting path and of k_fill_bins
#endif

#ifdef __HIPCC__
// binning group of this workgroup in K1 / k_tile_count
__device__ __forceinline__ int tgs_group_id() { return (int)blockIdx.x; }

__device__ __forceinline__ int xcc_id() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4); any value in [0, 8) gives correct results, the true
  // XCC id gives the locality
  return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & (TGS_XCC - 1));
}

struct GroupScan {
  int off[TGS_GROUP + 1];  // exclusive scan of tiles_hit inside the group
  int x0[TGS_GROUP], y0[TGS_GROUP], w[TGS_GROUP];
  unsigned depth_bits[TGS_GROUP];
  int wave_tot[TGS_GROUP / TGS_WAVE];
  int base, fits;
  int wave_bbox[TGS_GROUP / TGS_WAVE][4];   // per wave: min x0, min y0, max x1, max y1 of its rects (tiles)
  int hist[TGS_AGG_TILES];     // aggregated counting: pairs per tile of the bounding box, then their base rank
};

// Builds the in-group exclusive scan from each thread's (rect, depth).  Returns the group total.
// Must be called by all 256 threads; the caller synchronises before reading S.
__device__ __forceinline__ int group_scan_store(GroupScan& S, int hits, int x0, int y0, int w,
                                                unsigned dbits, int& my_off, int long_run) {
  const int tid = threadIdx.x;
  int incl = hits;
#pragma unroll
  for (int o = 1; o < TGS_WAVE; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if ((tid & (TGS_WAVE - 1)) >= o) incl += t;
  }
  if ((tid & (TGS_WAVE - 1)) == TGS_WAVE - 1) S.wave_tot[tid / TGS_WAVE] = incl;
  {   // tile bounding box of the wave's rects (DPP reductions; same-address LDS atomics would serialise)
    // (Gaussians with long runs of tiles stay outside the box: their pairs are counted directly, group_count_tiles)
    const bool any = hits > 0 && hits <= long_run;
    const int bx0 = wave_minmax_i<false>(any ? x0 : (1 << 30)), by0 = wave_minmax_i<false>(any ? y0 : (1 << 30));
    const int bx1 = wave_minmax_i<true>(any ? x0 + w : 0), by1 = wave_minmax_i<true>(any ? y0 + hits / max(w, 1) : 0);
    if ((tid & (TGS_WAVE - 1)) == 0) {
      int* b = S.wave_bbox[tid / TGS_WAVE];
      b[0] = bx0; b[1] = by0; b[2] = bx1; b[3] = by1;
    }
  }
  __syncthreads();
  int wbase = 0, total = 0;
#pragma unroll
  for (int i = 0; i < TGS_GROUP / TGS_WAVE; i++) {
    if (i < tid / TGS_WAVE) wbase += S.wave_tot[i];
    total += S.wave_tot[i];
  }
  my_off = wbase + incl - hits;
  S.off[tid] = my_off;
  S.x0[tid] = x0; S.y0[tid] = y0; S.w[tid] = max(w, 1);
  S.depth_bits[tid] = dbits;
  if (tid == 0) S.off[TGS_GROUP] = total;
  return total;
}

// pair i of the group -> (local Gaussian j, tile x, tile y)
__device__ __forceinline__ void group_pair_xy(const GroupScan& S, int i, int& j, int& tx, int& ty) {
  int lo = 0, hi = TGS_GROUP;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int mid = (lo + hi) >> 1;
    if (S.off[mid] <= i) lo = mid; else hi = mid;
  }
  j = lo;
  const int k = i - S.off[lo];
  const int w = S.w[lo];
  const int ky = k / w;
  tx = S.x0[lo] + (k - ky * w);
  ty = S.y0[lo] + ky;
}

// Clears the per-tile counters and the status word.  A kernel rather than hipMemsetAsync: the call
// then is an ordinary node when the step is captured into a hipGraph (a memset root node was seen to
// start before the previous graph launch on the same stream had drained), and it is a little
// cheaper than the runtime's fill.
// `sticky` (may be NULL): a persistent overflow word.  Once an overflowing frame has set it, every
// later frame starts with status[1] = 1 -- empty lists, and the guarded optimizer kernels do nothing
// -- until the host, which reads the status words late and without blocking, clears it and replays.
// element i of the clearing job (i = 0 .. max(TGS_XCC * T, 2) - 1); also run by other kernels' idle threads
__device__ __forceinline__ void clear_counters_elem(int i, int32_t* __restrict__ tile_cursor, int T,
                                                    int32_t* __restrict__ status, const int32_t* __restrict__ sticky) {
  if (i < TGS_XCC * T) tile_cursor[i] = 0;
  if (i < TGS_SCAN_WGS + 4) tile_cursor[2 * TGS_XCC * T + i] = 0;   // look-back flags of k_scan_tiles, longest list
  if (i < 4 * TGS_XCC) tile_cursor[TGS_ALLOC_OFF(T) + (i >> 2) * TGS_ALLOC_STRIDE + (i & 3)] = 0;   // pair allocator lines
  if (i < TGS_XCC) tile_cursor[TGS_MAXG_OFF(T) + i * TGS_ALLOC_STRIDE] = 0;                           // largest group total
  if (i == 0) status[0] = 0;
  if (i == 1) status[1] = sticky ? (*sticky != 0) : 0;
}
static __global__ __launch_bounds__(256) void k_clear_counters(int32_t* __restrict__ tile_cursor, int T,
                                                        int32_t* __restrict__ status,
                                                        const int32_t* __restrict__ sticky) {
  clear_counters_elem(blockIdx.x * 256 + threadIdx.x, tile_cursor, T, status, sticky);
}

// pair i of the group -> (local Gaussian j, tile id)
__device__ __forceinline__ void group_pair(const GroupScan& S, int TW, int i, int& j, int& tile) {
  int lo = 0, hi = TGS_GROUP;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int mid = (lo + hi) >> 1;
    if (S.off[mid] <= i) lo = mid; else hi = mid;
  }
  j = lo;
  const int k = i - S.off[lo];
  const int w = S.w[lo];
  const int ky = k / w;
  tile = (S.y0[lo] + ky) * TW + S.x0[lo] + (k - ky * w);
}

// U pairs of the group -> (local Gaussian, tile x, tile y), their binary searches in LOCK STEP: eight rounds of U
// independent LDS reads instead of 8 U dependent ones (a pair per branch made every read wait for the one before:
// ~1.6 us per round of four pairs on a CU that runs one workgroup -- the cost of k_fill_bins and of the counting loop
// on a group of thousands of pairs, round 6).  Indices beyond `total` are clamped (the caller masks them).
template <int U>
__device__ __forceinline__ void group_pairs_xy(const GroupScan& S, int i0, int stride, int total, int (&j)[U], int (&tx)[U],
                                               int (&ty)[U], bool (&valid)[U]) {
  int lo[U], hi[U], ii[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    valid[u] = i0 + u * stride < total;
    ii[u] = min(i0 + u * stride, total - 1);
    lo[u] = 0; hi[u] = TGS_GROUP;
  }
#pragma unroll
  for (int it = 0; it < 8; it++) {
    int o[U];
#pragma unroll
    for (int u = 0; u < U; u++) o[u] = S.off[(lo[u] + hi[u]) >> 1];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int mid = (lo[u] + hi[u]) >> 1;
      if (o[u] <= ii[u]) lo[u] = mid; else hi[u] = mid;
    }
  }
  int of[U], w[U], x0[U], y0[U];
#pragma unroll
  for (int u = 0; u < U; u++) { of[u] = S.off[lo[u]]; w[u] = S.w[lo[u]]; x0[u] = S.x0[lo[u]]; y0[u] = S.y0[lo[u]]; }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int k = ii[u] - of[u];
    const int ky = k / w[u];
    j[u] = lo[u];
    tx[u] = x0[u] + (k - ky * w[u]);
    ty[u] = y0[u] + ky;
  }
}

// Allocates the group's contiguous pair range and counts its intersections per tile; every pair
// remembers its arrival rank inside its tile.  Call after group_scan_store + __syncthreads-free
// (this function synchronises internally).
__device__ __forceinline__ void group_count_tiles(GroupScan& S, int TW, int T, int total,
                                                  int32_t* __restrict__ group_base,
                                                  int32_t* __restrict__ tile_count,
                                                  int32_t* __restrict__ rank,
                                                  int32_t* __restrict__ status, long long capacity,
                                                  int32_t* __restrict__ sticky, int long_run) {
  const int tid = threadIdx.x;
  const int x = xcc_id();
  // pair range of the group: one returning atomic on the XCD's allocator.  Issued first and consumed
  // last (publication below), so its round trip to the memory side overlaps the counting.
  int local = 0;
  int32_t* __restrict__ alloc = tile_count + TGS_ALLOC_OFF(T);
  if (tid == 0 && total > 0) {
    local = atomicAdd(&alloc[x * TGS_ALLOC_STRIDE], total);
#if TGS_MG_ATOMIC
    atomicMax(&tile_count[TGS_MAXG_OFF(T) + x * TGS_ALLOC_STRIDE], total);   // result unused: no return, no wait
#endif
  }
  int32_t* __restrict__ my_count = tile_count + (size_t)x * T;
  int bx0 = 1 << 30, by0 = 1 << 30, bx1 = 0, by1 = 0;   // tile bounding box of the group's rects
#pragma unroll
  for (int wv = 0; wv < TGS_GROUP / TGS_WAVE; wv++) {
    bx0 = min(bx0, S.wave_bbox[wv][0]); by0 = min(by0, S.wave_bbox[wv][1]);
    bx1 = max(bx1, S.wave_bbox[wv][2]); by1 = max(by1, S.wave_bbox[wv][3]);
  }
  const int bw = max(bx1 - bx0, 0), bh = max(by1 - by0, 0);
  const int area = bw * bh;
  // The box holds the group's SHORT runs only (<= TGS_LONG_RUN tiles, round 6).  A Gaussian that covers hundreds of tiles
  // would stretch it over the image and send all of the group's pairs down the direct path; its own pairs go there
  // anyway (every tile once: nothing to aggregate), the others keep the histogram.  The trainer's row order deals such
  // Gaussians evenly over the groups (optim.balanced_order), so nearly every group of an object-centric scene holds some.
  const bool agg = total > 0 && area > 0 && area <= TGS_AGG_TILES;   // workgroup-uniform
  if (agg)
    for (int b = tid; b < area; b += TGS_GROUP) S.hist[b] = 0;
  __syncthreads();                                            // the group scan (and the cleared histogram) is visible
  int lr[TGS_AGG_U], lb[TGS_AGG_U];
  if (agg) {
    {
      int j[TGS_AGG_U], tx[TGS_AGG_U], ty[TGS_AGG_U];
      bool valid[TGS_AGG_U], lng[TGS_AGG_U];
      group_pairs_xy<TGS_AGG_U>(S, tid, TGS_GROUP, total, j, tx, ty, valid);
#pragma unroll
      for (int u = 0; u < TGS_AGG_U; u++) lng[u] = S.off[j[u] + 1] - S.off[j[u]] > long_run;
#pragma unroll
      for (int u = 0; u < TGS_AGG_U; u++) {
        lb[u] = -1; lr[u] = 0;
        if (valid[u]) {
          if (lng[u]) {                                         // a long run's pair: counted directly (outside the box)
            lb[u] = -2;
            lr[u] = atomicAdd(&my_count[ty[u] * TW + tx[u]], 1);
          } else {
            lb[u] = (ty[u] - by0) * bw + (tx[u] - bx0);
            lr[u] = atomicAdd(&S.hist[lb[u]], 1);               // LDS: rank among the group's pairs of this tile
          }
        }
      }
    }
    __syncthreads();
    for (int b = tid; b < area; b += TGS_GROUP) {             // one global atomic per touched tile
      const int c = S.hist[b];
      if (c > 0) {
        const int by = b / bw;
        S.hist[b] = atomicAdd(&my_count[(by0 + by) * TW + bx0 + (b - by * bw)], c);
      }
    }
  }
  if (tid == 0) {                                             // publication of the pair range
    const long long region = capacity / TGS_XCC;              // pairs per XCD region
    int xr = x;
    long long loc = local;
    bool fits = loc + total <= region;
    if (!fits) {   // rare: this XCD's region is full -- take the range from another region with room
      // A failed attempt is NOT given back (round 5).  Rounds 3-4 subtracted the total again, which lets a third
      // group's add return an offset INSIDE a range a second group took in between (A: add t1 -> doesn't fit; B: add
      // t2 -> [A + t1, A + t1 + t2) fits; A: sub t1; C: add t3 -> [A + t2, ..) overlaps B's range): two groups then
      // write the same pair slots, the tile counters count both, and the lost pairs leave unwritten entries in the
      // sorted lists -- garbage Gaussian ids, a memory fault in K6 (seen on a held-out view of an 8-view model at
      // 89 % of the capacity).  Instead the overshoot stays on the counter, which CLOSES the region (every later add
      // returns a value beyond it), and word 2 of the line remembers it so that the frame's pair count stays exact.
      // The bound holds as before: a region is closed or refuses a group only when its real fill exceeds
      // region - (largest group total), so every allocation succeeds while #pairs + 8 x largest <= capacity.
      atomicAdd(&alloc[x * TGS_ALLOC_STRIDE + 2], total);
      for (int t = 1; t < TGS_XCC && !fits; t++) {
        xr = (x + t) & (TGS_XCC - 1);
        // closed or too full already (a racy look: only saves the add; the add's own result decides)
        if ((long long)__hip_atomic_load(&alloc[xr * TGS_ALLOC_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + total > region)
          continue;
        loc = atomicAdd(&alloc[xr * TGS_ALLOC_STRIDE], total);
        fits = loc + total <= region;
        if (!fits) atomicAdd(&alloc[xr * TGS_ALLOC_STRIDE + 2], total);
      }
      if (!fits) atomicAdd(&alloc[x * TGS_ALLOC_STRIDE + 1], total);   // still part of the frame's pair count
    }
    const int base = (int)(xr * region + loc);
    S.base = base;
    S.fits = fits;
    group_base[tgs_group_id()] = base;
    if (!fits) {
      status[1] = 1;
      if (sticky) *sticky = 1;
    }
  }
  __syncthreads();
  const long long base = S.base;
  const bool fits = S.fits != 0;
  int first_direct = 0;
  if (agg) {
#pragma unroll
    for (int u = 0; u < TGS_AGG_U; u++) {
      const int i = tid + u * TGS_GROUP;
      if (lb[u] != -1 && fits) rank[base + i] = (int32_t)((unsigned)((lb[u] >= 0 ? S.hist[lb[u]] : 0) + lr[u]) | ((unsigned)x << 29));
    }
    first_direct = TGS_AGG_U * TGS_GROUP;                     // pairs beyond 1024 per group (huge footprints)
  }
  // Direct path, TGS_DIRECT_U pairs per thread and round with their returning atomics in flight together (round 6): a
  // group of table / background Gaussians that cover a thousand tiles each holds 20 - 30 k pairs (mean of an
  // object-centric 720p frame: 2.6 k), and one returning atomic per round made its ~110 dependent round trips the
  // duration of the whole launch (K1 + count 104 us for 79 k Gaussians, profiles/r6_before_*).
  for (int i0 = tid + first_direct; i0 < total; i0 += TGS_DIRECT_U * TGS_GROUP) {
    int tile[TGS_DIRECT_U], r[TGS_DIRECT_U];
    {
      int j[TGS_DIRECT_U], tx[TGS_DIRECT_U], ty[TGS_DIRECT_U];
      bool valid[TGS_DIRECT_U];
      group_pairs_xy<TGS_DIRECT_U>(S, i0, TGS_GROUP, total, j, tx, ty, valid);
#pragma unroll
      for (int u = 0; u < TGS_DIRECT_U; u++) tile[u] = valid[u] ? ty[u] * TW + tx[u] : -1;
    }
#pragma unroll
    for (int u = 0; u < TGS_DIRECT_U; u++) r[u] = tile[u] >= 0 ? atomicAdd(&my_count[tile[u]], 1) : 0;
#pragma unroll
    for (int u = 0; u < TGS_DIRECT_U; u++)   // rank inside sub-list x of the tile (capacity < 2^29 pairs per sub-list)
      if (tile[u] >= 0 && fits) rank[base + i0 + u * TGS_GROUP] = (int32_t)((unsigned)r[u] | ((unsigned)x << 29));
  }
}
#endif  // __HIPCC__
