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
__device__ __forceinline__ int group_load_scan(const CamK& cam, int N, int group,
                                               const float* __restrict__ splats, GroupScan& S,
                                               int& my_off) {
  const int g = group * TGS_GROUP + threadIdx.x;
  int hits = 0, x0 = 0, y0 = 0, w = 0;
  unsigned dbits = 0;
  if (g < N) {
    const float* rec = splats + (size_t)g * TGS_SPLAT_FLOATS;
    int h;
    unpack_rect(__float_as_uint(rec[10]), x0, y0, w, h);
    hits = w * h;
    dbits = __float_as_uint(rec[2]);
  }
  return group_scan_store(S, hits, x0, y0, w, dbits, my_off, cam.long_run);
}

// K3a: count intersections per tile; allocate the group's pair range; remember every pair's
// arrival rank inside its tile.
__global__ __launch_bounds__(TGS_GROUP) void k_tile_count(
    CamK cam, int N, float* __restrict__ splats, int32_t* __restrict__ group_base,
    int32_t* __restrict__ tile_count, int32_t* __restrict__ rank, int32_t* __restrict__ status,
    long long capacity, int32_t* __restrict__ sticky) {
  __shared__ GroupScan S;
  int my_off;
  const int total = group_load_scan(cam, N, tgs_group_id(), splats, S, my_off);
  const int g = tgs_group_id() * TGS_GROUP + threadIdx.x;
  if (g < N) splats[(size_t)g * TGS_SPLAT_FLOATS + 11] = __int_as_float(my_off);
  group_count_tiles(S, cam.TW, cam.TW * cam.TH, total, group_base, tile_count, rank, status, capacity, sticky, cam.long_run);
}

// K4: exclusive scan of the [8][T] per-XCD tile counters in tile-major order -> tile_start[T+1] and
// sub_start[8][T] (start of XCD x's sub-list inside tile t's list).  Workgroup b of the first
// NB = ceil(T/1024) owns tiles [1024 b, 1024 b + 1024): thread = tile, 8 coalesced counter loads, a
// workgroup scan of the tile totals, then the workgroup publishes its AGGREGATE (not its prefix) and
// sums the aggregates of the lower workgroups itself -- a one-step look-back, no chain, ~2 global
// round trips in all (the single-workgroup version took 14 us for 8160 counters, 54 us for 8 x 8160).
// On capacity overflow every list is made empty so that downstream kernels touch nothing.
//
// Workgroups NB..NB+7 (launched only when a tile_order buffer is given) build the visiting order of
// the compositing kernels: block b of K6/K7 runs on XCD b % 8 and owns tile tile_order[b]; every XCD
// keeps its contiguous band of tiles (so the splat records neighbouring tiles share stay in its L2)
// but visits it longest list first.  With ~2 tiles per resident wave slot the spatial order leaves
// the last slots running alone for a whole tile; longest-first shortens that tail (K6 -3 %, K7 -4 %,
// alternating same-box runs after clock warm-up).
//
// Front prefetch (tgs_project_bin_sort_front): this frame's K1 ran inside the previous step's fused optimizer kernel,
// which stores `tag_expect` into *front_tag as its last act.  A kernel that was voided by its overflow guard leaves the
// old tag -- and the sticky overflow word raised, which empties this frame anyway: a mismatch therefore simply VOIDS
// the frame (empty lists, status[1] = 1, sticky raised; every workgroup takes the decision from the same two words).
// Rounds 3-4 spent a launch of N idle threads on it (k_project_fwd_unless_done: 4.9 us at cfg3) that re-ran K1 for a
// frame nothing would ever look at.  The n_clear workgroups behind the order blocks clear the counters and the status
// word of the frame AFTER this one (whose K1 this step's optimizer kernel will run) -- that launch's other job.
struct ScanFront {
  const int32_t* tag;      // NULL: no front prefetch involved
  int32_t tag_expect;
  int32_t* sticky;         // raised on a mismatch (may be NULL)
  int32_t* next_cursor;    // NULL: nothing to clear
  int32_t* next_status;
  int next_T, n_clear, n_order8;
};
__global__ __launch_bounds__(1024) void k_scan_tiles(int T, int NB, int32_t* __restrict__ tile_count,
                                                     int32_t* __restrict__ tile_start,
                                                     int32_t* __restrict__ status,
                                                     int32_t* __restrict__ tile_order,
                                                     const float* __restrict__ splats, int N, ScanFront fr) {
  __shared__ int wave_tot[16];
  __shared__ int wave_mg[16];
  __shared__ int s_base;
  const int tid = threadIdx.x;
  const bool voided = fr.tag && *fr.tag != fr.tag_expect;
  if ((int)blockIdx.x >= NB + fr.n_order8 && (int)blockIdx.x < NB + fr.n_order8 + fr.n_clear) {
    clear_counters_elem(((int)blockIdx.x - NB - fr.n_order8) * 1024 + tid, fr.next_cursor, fr.next_T, fr.next_status, fr.sticky);
    return;
  }
  if ((int)blockIdx.x == (int)gridDim.x - 1) {
    // The LAST workgroup writes the status word, beside the scan (it needs nothing from it):
    // status[0]: the frame's intersection count = the 8 per-XCD allocators + the pairs that found no
    // room; status[2]: the capacity that is SUFFICIENT for this frame whatever XCD its workgroups run
    // on, #pairs + 8 x the largest group total (tgs_binning.h).  A group's total = in-group offset of its
    // last Gaussian (record slot 11) + that Gaussian's tile count (rect in slot 10).  On overflow
    // status[0] = status[2], so that a caller that grows to status[0] x growth converges in one retry.
    int mg = 0;
    // (a frame voided by its front tag has no records: the K1 that would have written them never ran -- the buffers hold
    // whatever the allocator left, and a garbage "sufficient capacity" would make the caller grow to it)
#if TGS_MG_ATOMIC
    if (!voided && tid < TGS_XCC) mg = tile_count[TGS_MAXG_OFF(T) + tid * TGS_ALLOC_STRIDE];   // folded in by every K1 group
    const int G = 0;
#else
    const int G = voided ? 0 : (N + TGS_GROUP - 1) / TGS_GROUP;
#endif
    for (int g = tid; g < G; g += 1024) {
      const float* rec = splats + (size_t)min(g * TGS_GROUP + TGS_GROUP - 1, N - 1) * TGS_SPLAT_FLOATS;
      int x0, y0, w, h;
      unpack_rect(__float_as_uint(rec[10]), x0, y0, w, h);
      mg = max(mg, __float_as_int(rec[11]) + w * h);
    }
    mg = wave_max_i(mg);
    if ((tid & 63) == 0) wave_mg[tid >> 6] = mg;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 16; w++) mg = max(mg, wave_mg[w]);
      const int32_t* xa = tile_count + TGS_ALLOC_OFF(T);
      long long n = 0;
      for (int x = 0; x < TGS_XCC; x++)   // allocated (+ overshoot of failed attempts, word 2) + pairs that found no room
        n += (long long)xa[x * TGS_ALLOC_STRIDE] - xa[x * TGS_ALLOC_STRIDE + 2] + xa[x * TGS_ALLOC_STRIDE + 1];
      const long long need = voided ? 0 : n + (long long)TGS_XCC * mg;
      if (voided) {                 // (the scan workgroups read the tag themselves: this store is for the later kernels)
        status[1] = 1;
        if (fr.sticky) *fr.sticky = 1;
      }
      status[0] = (int32_t)min((status[1] != 0 || voided) ? need : n, 0x7fffffffll);
      status[2] = (int32_t)min(need, 0x7fffffffll);
      status[3] = 0;                // longest list of the frame: k_fill_bins fills it in (no Gaussians: no lists)
    }
    return;
  }
  if ((int)blockIdx.x >= NB) {
    // K6 / K7 schedule when no sort launch follows (no Gaussians: every list is empty): the XCD's
    // slots in spatial order
    const int x = blockIdx.x - NB;
    const int per = tgs_xcd_slots(T);
    for (int i = tid; i < per; i += 1024) tile_order[i * TGS_XCDS + x] = tgs_xcd_slot_tile(T, x, i);
    return;
  }
  const bool overflow = status[1] != 0 || voided;
  int32_t* __restrict__ sub_start = tile_count + TGS_XCC * T;
  int32_t* __restrict__ agg = sub_start + TGS_XCC * T;     // [NB] aggregate + 1 of every scan workgroup, 0 = not yet
  const int b = blockIdx.x, t = b * 1024 + tid;
  int c[TGS_XCC];
  int sum = 0;
#pragma unroll
  for (int x = 0; x < TGS_XCC; x++) {
    c[x] = (t < T && !overflow) ? tile_count[x * T + t] : 0;
    sum += c[x];
  }
  int incl = sum;
#pragma unroll
  for (int o = 1; o < TGS_WAVE; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if ((tid & 63) >= o) incl += v;
  }
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  int wbase = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    if (w < (tid >> 6)) wbase += wave_tot[w];
    tot += wave_tot[w];
  }
  if (tid == 0) __hip_atomic_store(&agg[b], tot + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  {   // longest list of the frame: lets the long-list sort launches return immediately when unused
    const int wmax = wave_max_i(sum);
    if ((tid & 63) == 0 && wmax > 0) atomicMax(agg + TGS_SCAN_WGS, wmax);
  }
  // look-back: sum the aggregates of workgroups 0 .. b-1 (NB <= 64: one wave)
  if (tid < TGS_WAVE) {
    int v = 0;
    if (tid < b) {
      int a;
      do { a = __hip_atomic_load(&agg[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while (a == 0);
      v = a - 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (tid == 0) s_base = v;
  }
  __syncthreads();
  int run = s_base + wbase + incl - sum;
  if (t < T) {
    tile_start[t] = run;
#pragma unroll
    for (int x = 0; x < TGS_XCC; x++) {
      sub_start[x * T + t] = run;
      run += c[x];
    }
  }
  if (b == NB - 1 && tid == 0) tile_start[T] = s_base + tot;
  // 16 of the 512 scratch ints behind the tile starts collect the frame's deepest walk and the sum of its walks (K6,
  // raster.hip publish_walk: a pair of words per XCD, 256 B apart) for K7
  if (b == NB - 1 && tid >= 1 && tid <= 16) tile_start[((tid - 1) & 1) ? TGS_WALKSUM_AT(T, (tid - 1) >> 1) : TGS_WALK_AT(T, (tid - 1) >> 1)] = 0;
  // ... and the 8 slot counters of K7's four-wave launch (TGS_SLOTCTR_AT): k_raster_bwd clears them again before every
  // backward, but a buffer that comes out of torch.empty must not depend on that launch having reached the store
  if (b == NB - 1 && tid >= 17 && tid <= 24) tile_start[TGS_SLOTCTR_AT(T, tid - 17)] = 0;
}

// K3b: scatter (gid, depth bits) into the tile bins.  slot = tile_start + arrival rank.
__global__ __launch_bounds__(TGS_GROUP) void k_fill_bins(
    CamK cam, int N, const float* __restrict__ splats, const int32_t* __restrict__ group_base,
    const int32_t* __restrict__ sub_start, const int32_t* __restrict__ rank,
    uint2* __restrict__ pairs, int32_t* __restrict__ status, const int32_t* __restrict__ max_list) {
  // status[3] = the frame's longest list (complete: the scan launch has ended): what a caller sizes `max_list_hint` from
  if (blockIdx.x == 0 && threadIdx.x == 0) status[3] = *max_list;
  if (status[1]) return;
  const int T = cam.TW * cam.TH;
  __shared__ GroupScan S;
  int my_off;
  const int total = group_load_scan(cam, N, blockIdx.x, splats, S, my_off);
  __syncthreads();
  const long long base = group_base[blockIdx.x];
  const int g0 = blockIdx.x * TGS_GROUP;
  // TGS_DIRECT_U pairs per thread and round: the rank loads, then the sub-list starts they select, in flight together
  // (round 6; one pair per round was a chain of two dependent global loads per pair -- 85 us at 0.8 M pairs when one
  // group of huge Gaussians holds 28 k of them, 26 us for cfg3's 3.3 M evenly spread ones)
  for (int i0 = threadIdx.x; i0 < total; i0 += TGS_DIRECT_U * TGS_GROUP) {
    int j[TGS_DIRECT_U], tile[TGS_DIRECT_U], slot[TGS_DIRECT_U];
    unsigned r[TGS_DIRECT_U];
    {
      int tx[TGS_DIRECT_U], ty[TGS_DIRECT_U];
      bool valid[TGS_DIRECT_U];
      group_pairs_xy<TGS_DIRECT_U>(S, i0, TGS_GROUP, total, j, tx, ty, valid);
#pragma unroll
      // (indices beyond `total` were clamped to the group's last pair: the loads below stay unconditional -- branches
      // around them made every load wait for the previous one -- and only the store is masked)
      for (int u = 0; u < TGS_DIRECT_U; u++) tile[u] = ty[u] * cam.TW + tx[u];
#pragma unroll
      for (int u = 0; u < TGS_DIRECT_U; u++)   // sub-list (XCC) id << 29 | rank inside it
        r[u] = (unsigned)rank[base + min(i0 + u * TGS_GROUP, total - 1)];
#pragma unroll
      for (int u = 0; u < TGS_DIRECT_U; u++)
        slot[u] = sub_start[(r[u] >> 29) * T + tile[u]] + (int)(r[u] & 0x1fffffffu);
#pragma unroll
      for (int u = 0; u < TGS_DIRECT_U; u++)
        if (valid[u]) pairs[slot[u]] = make_uint2((unsigned)(g0 + j[u]), S.depth_bits[j[u]]);  // u64 = depth<<32 | gid
    }
  }
}

__device__ __forceinline__ int next_pow2(int n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}

// ---------------------------------------------------------------------------------------------
// K5: per-tile sort of (depth bits, gid), register resident: ONE wave64 sorts one tile's list.
// ---------------------------------------------------------------------------------------------
// The first version of this kernel ran the bitonic network of a 256-thread workgroup in LDS: 45
// stages x 8 KB of ds_read_b64 / ds_write_b64 per 512-element list -- ~3 GB of LDS traffic per cfg3
// frame, i.e. LDS-bandwidth bound (63 us, 38 % bank-conflict cycles).  Here every lane keeps E keys
// (virtual index e = lane * E + r) in VGPRs: the stages with partner distance j < E are
// register-to-register, the others exchange with lane ^ (j / E) through DPP (quad_perm, row_ror:8,
// bank-masked row shifts) or the gfx950 lane-swap instructions -- no LDS, no barriers, and the
// global loads stay coalesced.  E = 8 / 16 covers lists <= 512 / 1024 (the common class),
// E = 32 / 64 (lists <= 2048 / 4096: object-centric scenes) run from a second, small launch.
typedef unsigned long long u64;
typedef unsigned tgs_u2 __attribute__((ext_vector_type(2)));

template <int CTRL, int BANK>
__device__ __forceinline__ unsigned dpp_upd(unsigned old, unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, BANK, false);
}

// value of lane (l ^ D) for D = 1, 2, 4, 8, 16, 32
template <int D>
__device__ __forceinline__ unsigned lane_xor(unsigned v, int lane) {
  if constexpr (D == 1) return dpp_upd<0xB1, 0xf>(v, v);          // quad_perm [1,0,3,2]
  else if constexpr (D == 2) return dpp_upd<0x4E, 0xf>(v, v);     // quad_perm [2,3,0,1]
  else if constexpr (D == 4) {                                    // banks 0,2 <- lane+4; banks 1,3 <- lane-4
    const unsigned t = dpp_upd<0x104, 0x5>(v, v);                 // row_shl:4
    return dpp_upd<0x114, 0xA>(t, v);                             // row_shr:4
  } else if constexpr (D == 8) return dpp_upd<0x128, 0xf>(v, v);  // row_ror:8
  else if constexpr (D == 16) {
    const tgs_u2 r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // x = rows {0,0,2,2}, y = rows {1,1,3,3}
    return (lane & 16) ? r.x : r.y;
  } else {
    const tgs_u2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // x = {lo,lo}, y = {hi,hi}
    return (lane & 32) ? r.x : r.y;
  }
}

template <int D>
__device__ __forceinline__ u64 lane_xor64(u64 v, int lane) {
  return ((u64)lane_xor<D>((unsigned)(v >> 32), lane) << 32) | lane_xor<D>((unsigned)v, lane);
}

// one cross-lane stage: every key meets the same register of lane ^ D; the lower lane of a pair
// keeps the minimum when the pair sorts upwards
template <int E, int D>
__device__ __forceinline__ void cross_stage(u64 (&k)[E], int lane, bool up) {
  const bool keep_min = (((lane & D) == 0) == up);
#pragma unroll
  for (int r = 0; r < E; r++) {
    const u64 p = lane_xor64<D>(k[r], lane);
    k[r] = ((p < k[r]) == keep_min) ? p : k[r];
  }
}

// register-to-register stage of distance J inside every lane; UPMASK < 0: direction `up` for all
// registers, else the direction of register r is ((r & UPMASK) == 0)
template <int E, int J, int UPMASK>
__device__ __forceinline__ void local_stage(u64 (&k)[E], bool up) {
#pragma unroll
  for (int r = 0; r < E; r++) {
    if ((r & J) == 0) {
      const bool u = UPMASK < 0 ? up : ((r & UPMASK) == 0);
      const u64 a = k[r], b = k[r | J];
      const bool sw = (a > b) == u;
      k[r] = sw ? b : a;
      k[r | J] = sw ? a : b;
    }
  }
}

template <int E, int J, int UPMASK>
__device__ __forceinline__ void local_tail(u64 (&k)[E], bool up) {   // stages J, J/2, ..., 1
  if constexpr (J >= 1) {
    local_stage<E, J, UPMASK>(k, up);
    local_tail<E, J / 2, UPMASK>(k, up);
  }
}

// levels KK = 2 .. E: the direction bit of a pair is a bit of the register index (KK < E) or the
// lowest lane bit (KK == E)
template <int E, int KK>
__device__ __forceinline__ void local_levels(u64 (&k)[E], int lane) {
  if constexpr (KK <= E) {
    if constexpr (KK < E) local_tail<E, KK / 2, KK>(k, true);
    else local_tail<E, E / 2, -1>(k, (lane & 1) == 0);
    local_levels<E, KK * 2>(k, lane);
  }
}

// One cross-WAVE stage (partner thread = tid ^ d, d >= 64) of a multi-wave workgroup: the keys
// travel through LDS in register-major order (conflict-free ds_write_b64 / ds_read_b64).
template <int E>
__device__ __forceinline__ void cross_wave_stage(u64 (&k)[E], int tid, int d, bool up, u64* __restrict__ lds, int NT) {
  const bool keep_min = (((tid & d) == 0) == up);
#pragma unroll
  for (int r = 0; r < E; r++) lds[r * NT + tid] = k[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < E; r++) {
    const u64 p = lds[r * NT + (tid ^ d)];
    k[r] = ((p < k[r]) == keep_min) ? p : k[r];
  }
  __syncthreads();
}

// Sorts one tile's list with W waves x 64 lanes x E keys (n <= 64 W E).  W == 1 needs no LDS and no
// barriers; for W > 1 only the stages whose partner lives in another wave (log2 W (log2 W + 1) / 2
// of them) go through LDS.
template <int E, int W>
__device__ __forceinline__ void sort_regs(u64 (&k)[E], int tid, u64* __restrict__ lds) {
  constexpr int NT = TGS_WAVE * W;
  const int lane = tid & (TGS_WAVE - 1);
  local_levels<E, 2>(k, lane);
  constexpr int LV = W == 1 ? 6 : (W == 4 ? 8 : 10);   // log2(64 W)
#pragma unroll 1
  for (int lv = 1; lv <= LV; lv++) {   // levels KK = 2E .. 64WE: direction = bit lv of the thread id
    const bool up = ((tid >> lv) & 1) == 0;
    if constexpr (W > 1)
      for (int d = 1 << (lv - 1); d >= TGS_WAVE; d >>= 1) cross_wave_stage<E>(k, tid, d, up, lds, NT);
    if (lv >= 6) cross_stage<E, 32>(k, lane, up);
    if (lv >= 5) cross_stage<E, 16>(k, lane, up);
    if (lv >= 4) cross_stage<E, 8>(k, lane, up);
    if (lv >= 3) cross_stage<E, 4>(k, lane, up);
    if (lv >= 2) cross_stage<E, 2>(k, lane, up);
    cross_stage<E, 1>(k, lane, up);
    local_tail<E, E / 2, -1>(k, up);
  }
}

template <int E, int W>
__device__ __forceinline__ void sort_tile_regs(const u64* __restrict__ pairs, int32_t* __restrict__ sorted_gid,
                                               int s, int n, int tid, u64* __restrict__ lds) {
  constexpr int NT = TGS_WAVE * W;
  u64 k[E];
#pragma unroll
  for (int r = 0; r < E; r++) {        // coalesced load; the initial order is irrelevant
    const int i = r * NT + tid;
    k[r] = i < n ? pairs[s + i] : ~0ull;
  }
  sort_regs<E, W>(k, tid, lds);
  // sorted order: e = tid * E + r
#pragma unroll
  for (int r = 0; r < E; r++) {
    const int e = tid * E + r;
    if (e < n) sorted_gid[s + e] = (int)(k[r] & 0xffffffffull);
  }
}

// Visiting order of the compositing kernels for XCD x (tgs_common.h: every 8th granule of 8 tiles): its
// slots are cut into chunks of <= 1024 consecutive slots and every chunk is put in descending
// list-length order, ties by tile id, by one wave -- the same register sort on (~length, tile) keys.
// The XCD therefore visits chunk 0 longest first, then chunk 1, ...: the tail of the launch is made
// of the last chunk's shortest lists.
template <int E>
__device__ __forceinline__ void xcd_order_regs_e(int T, int x, int i0, int len, const int32_t* __restrict__ tile_start,
                                                 int32_t* __restrict__ tile_order, int lane) {
  u64 k[E];
#pragma unroll
  for (int r = 0; r < E; r++) {
    const int i = r * TGS_WAVE + lane;
    const int tile = i < len ? tgs_xcd_slot_tile(T, x, i0 + i) : T;
    k[r] = tile < T ? ((u64)(~(unsigned)(tile_start[tile + 1] - tile_start[tile])) << 32) | (unsigned)tile : ~0ull;
  }
  sort_regs<E, 1>(k, lane, nullptr);
#pragma unroll
  for (int r = 0; r < E; r++) {
    const int e = lane * E + r;
    if (e < len) tile_order[(i0 + e) * TGS_XCDS + x] = k[r] != ~0ull ? (int)(k[r] & 0xffffffffull) : T;   // T = no tile
  }
}
__device__ __forceinline__ void xcd_order_regs(int T, int x, int c, int chunk,
                                               const int32_t* __restrict__ tile_start,
                                               int32_t* __restrict__ tile_order, int lane) {
  const int per = tgs_xcd_slots(T);
  const int i0 = c * chunk, len = max(0, min(chunk, per - i0));
  // keys per lane by chunk length, like the list classes: a 720p frame has 450 slots per XCD (8 keys per lane: 360 stage x key
  // units instead of 864) -- these blocks are the longest of the wave-sort launch when the lists are short (round 6)
  if (len <= 128) xcd_order_regs_e<2>(T, x, i0, len, tile_start, tile_order, lane);
  else if (len <= 256) xcd_order_regs_e<4>(T, x, i0, len, tile_start, tile_order, lane);
  else if (len <= 512) xcd_order_regs_e<8>(T, x, i0, len, tile_start, tile_order, lane);
  else xcd_order_regs_e<16>(T, x, i0, len, tile_start, tile_order, lane);
}

// common classes: lists <= 64 / 128 / 256 / 512 / 1024 entries with 1 / 2 / 4 / 8 / 16 keys per lane; one wave per tile
// `sorted_up_to`: the longest list some launch of this frame sorts (1024 / 4096 / INT_MAX by the caller's
// max_list_hint).  A longer list means the caller's hint was wrong: the wave copies the ids UNSORTED -- valid indices,
// so the compositing kernels run on without faulting -- and raises the overflow flag and the sticky word: the frame
// is void exactly like one that overflowed its capacity, status[3] tells the caller what the hint should have been.
__global__ __launch_bounds__(TGS_WAVE) void k_sort_tiles_wave(
    int T, const int32_t* __restrict__ tile_start, const u64* __restrict__ pairs,
    int32_t* __restrict__ sorted_gid, int32_t* __restrict__ tile_order, int n_order, int chunk,
    int sorted_up_to, int32_t* __restrict__ status, int32_t* __restrict__ sticky, int wave_up_to) {
  // the first n_order (0 or 8 x chunks per XCD) blocks build the K6 / K7 schedule of one chunk of one
  // XCD's slots each; they are the longest blocks of the launch, so they are dispatched first
  if ((int)blockIdx.x < n_order) {
    xcd_order_regs(T, blockIdx.x % TGS_XCDS, blockIdx.x / TGS_XCDS, chunk, tile_start, tile_order, threadIdx.x);
    return;
  }
  const int tile = blockIdx.x - n_order;
  const int s = tile_start[tile];
  const int n = tile_start[tile + 1] - s;
  const int lane = threadIdx.x;
  if (n > sorted_up_to) {
    for (int i = lane; i < n; i += TGS_WAVE) sorted_gid[s + i] = (int)(pairs[s + i] & 0xffffffffull);
    if (lane == 0) {
      status[1] = 1;
      if (sticky) *sticky = 1;
    }
    return;
  }
  // lists beyond wave_up_to (512 when the four-wave launch follows, else 1024) are that launch's: a 1024-key list costs one
  // wave 864 stage x key units against 360 for 512 keys, and in an object-centric frame this launch lasts as long as its
  // longest list (round 6: k_sort_tiles_wave 24.5 -> 13 us on the 720p checkpoints)
  if (n <= 0 || n > wave_up_to) return;
  if (n == 1) {
    if (lane == 0) sorted_gid[s] = (int)(pairs[s] & 0xffffffffull);
    return;
  }
  // keys per lane by list length: the network's work is (stages x keys per lane) = 21 / 56 / 144 / 360 / 864 for
  // 1 / 2 / 4 / 8 / 16 keys, so a list of 200 entries costs 0.4 of what it cost padded to 512 (round 5; the short lists are
  // the periphery of every frame and most of an object-centric one)
  if (n <= 64) sort_tile_regs<1, 1>(pairs, sorted_gid, s, n, lane, nullptr);
  else if (n <= 128) sort_tile_regs<2, 1>(pairs, sorted_gid, s, n, lane, nullptr);
  else if (n <= 256) sort_tile_regs<4, 1>(pairs, sorted_gid, s, n, lane, nullptr);
  else if (n <= 512) sort_tile_regs<8, 1>(pairs, sorted_gid, s, n, lane, nullptr);
  else sort_tile_regs<16, 1>(pairs, sorted_gid, s, n, lane, nullptr);
}

// Rare list classes (`max_list`, written by k_scan_tiles, lets both launches return at once when the
// frame has no list of their class; grid-stride over the tiles):
//   k_sort_tiles_wg4   (1024, 4096]   4 waves x 16 keys per lane, 32 KB of LDS: several workgroups
//                      per CU overlap each other's barriers (object-centric scenes live here);
//   k_sort_tiles_huge  (4096, 16384]  16 waves x 16 keys per lane (128 KB of LDS), and beyond that a
//                      bitonic network in global memory on a power-of-two padded copy at
//                      fb[2*s ...) (next_pow2(n) < 2n, so per-tile regions never overlap).
// (One merged 1024-thread launch was 26 us slower on the clustered scene: one workgroup per CU.)
__global__ __launch_bounds__(TGS_WAVE * 4) void k_sort_tiles_wg4(
    int T, const int32_t* __restrict__ tile_start, const u64* __restrict__ pairs,
    int32_t* __restrict__ sorted_gid, const int32_t* __restrict__ max_list, int lo) {
  __shared__ u64 wg_keys[TGS_WAVE * 4 * 16];   // 16 keys x 256 threads
  if (*max_list <= lo) return;
  for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const int s = tile_start[tile];
    const int n = tile_start[tile + 1] - s;
    if (n <= lo || n > 4096) continue;
    __syncthreads();
    if (n <= 1024) { sort_tile_regs<4, 4>(pairs, sorted_gid, s, n, threadIdx.x, wg_keys); continue; }
    // (1024, 2048]: 8 keys per lane -- half the network per thread; in an object-centric frame this launch lasts as long
    // as one tile's sort (a few hundred long lists, every workgroup resident at once)
    if (n <= 2048) sort_tile_regs<8, 4>(pairs, sorted_gid, s, n, threadIdx.x, wg_keys);
    else sort_tile_regs<16, 4>(pairs, sorted_gid, s, n, threadIdx.x, wg_keys);
  }
}

__global__ __launch_bounds__(1024) void k_sort_tiles_huge(
    int T, const int32_t* __restrict__ tile_start, const u64* __restrict__ pairs, u64* __restrict__ fb,
    int32_t* __restrict__ sorted_gid, const int32_t* __restrict__ max_list) {
  __shared__ u64 wg_keys[TGS_WAVE * 16 * 16];   // 128 KB: 16 keys x 1024 threads
  if (*max_list <= 4096) return;
  const int tid = threadIdx.x;
  for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const int s = tile_start[tile];
    const int n = tile_start[tile + 1] - s;
    if (n <= 4096) continue;
    if (n <= 16384) {
      __syncthreads();
      sort_tile_regs<16, 16>(pairs, sorted_gid, s, n, tid, wg_keys);
      continue;
    }
    const int np2 = next_pow2(n);
    u64* keys = fb + 2 * (size_t)s;
    __syncthreads();
    for (int i = tid; i < np2; i += 1024) keys[i] = (i < n) ? pairs[s + i] : ~0ull;
    __threadfence_block();
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < (np2 >> 1); i += 1024) {
          const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
          const int hi = lo | j;
          const bool up = (lo & k) == 0;
          const u64 a = keys[lo], b = keys[hi];
          if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
        }
        __threadfence_block();
        __syncthreads();
      }
    }
    for (int i = tid; i < n; i += 1024) sorted_gid[s + i] = (int)(keys[i] & 0xffffffffull);
  }
}

}  // namespace


extern "C" int tgs_num_groups(int N) { return (N + TGS_GROUP - 1) / TGS_GROUP; }
extern "C" int tgs_num_tiles(int W, int H) {
  return ((W + TGS_BLOCK - 1) / TGS_BLOCK) * ((H + TGS_BLOCK - 1) / TGS_BLOCK);
}
extern "C" int tgs_tile_order_len(int W, int H) {
  return TGS_XCDS * tgs_xcd_slots(tgs_num_tiles(W, H));
}
extern "C" int tgs_num_bands(int W, int H) { return tgs_band_count(tgs_num_tiles(W, H)); }
extern "C" int tgs_band_tiles(int W, int H, int band, int* tile0, int* tile1) {
  const int T = tgs_num_tiles(W, H), c = tgs_band_slots(T);
  if (band < 0 || band >= tgs_band_count(T)) return TGS_E_ARG;
  const long long a = (long long)band * c * TGS_XCDS, b = a + (long long)c * TGS_XCDS;
  if (tile0) *tile0 = (int)(a < T ? a : T);
  if (tile1) *tile1 = (int)(b < T ? b : T);
  return TGS_OK;
}
extern "C" int tgs_tile_counter_len(int W, int H) { return tgs_counter_len(tgs_num_tiles(W, H)); }
// tile starts [T + 1] + the rasterizer's scratch (walk words, slot counters: raster.hip); k_scan_tiles zeroes them
extern "C" int64_t tgs_tile_start_len(int W, int H) { return (int64_t)tgs_num_tiles(W, H) + 1 + TGS_TILE_START_SCRATCH; }
// scratch layout: pairs u64[cap] | fallback u64[2*cap] | rank i32[cap]
extern "C" size_t tgs_sort_scratch_bytes(int64_t capacity) {
  if (capacity < 0) capacity = 0;
  return (size_t)capacity * (8 + 16 + 4) + 64;
}

int tgs_bin_finish(const CamK& k, int N, const float* splats, const int32_t* group_base,
                   int32_t* tile_start, int64_t tile_start_len, int32_t* tile_cursor, int32_t* sorted_gid,
                   int32_t* tile_order, int64_t capacity, void* scratch, int32_t* status,
                   int32_t* sticky_overflow, int32_t max_list_hint, const BinFront* front, hipStream_t s) {
  const int T = k.TW * k.TH;
  TGS_CHECK_ARG(tile_start_len >= (int64_t)T + 1 + TGS_TILE_START_SCRATCH,
                "tile_start buffer shorter than tgs_tile_start_len(W, H) (the scan zeroes the rasterizer's scratch behind the starts)");
  const int G = tgs_num_groups(N);
  const BinScratch sc = carve_scratch(scratch, capacity);
  const int NB = (T + 1023) / 1024;   // <= 64 (image sides are limited to 255 tiles)
  // K6 / K7 schedule: ordered by extra blocks of the wave-sort launch (one per chunk of <= 1024 slots
  // of every XCD); without Gaussians no sort is launched and 8 workgroups of the scan write the
  // spatial order
  const bool order_in_sort = tile_order && G > 0;
  const int per = tgs_xcd_slots(T);
  const int chunk = tgs_band_slots(T), n_sub = tgs_band_count(T);
  const int n_order = order_in_sort ? TGS_XCDS * n_sub : 0;
  (void)per;
  ScanFront fr;
  fr.tag = front ? front->tag_word : nullptr;
  fr.tag_expect = front ? front->tag_expect : 0;
  fr.sticky = sticky_overflow;
  fr.next_cursor = front ? front->next_tile_cursor : nullptr;
  fr.next_status = front ? front->next_status : nullptr;
  fr.next_T = front ? front->next_T : 0;
  fr.n_clear = fr.next_cursor ? (max(TGS_XCC * fr.next_T, 2) + 1023) / 1024 : 0;
  fr.n_order8 = (tile_order && !order_in_sort) ? 8 : 0;
  hipLaunchKernelGGL(k_scan_tiles, dim3(NB + fr.n_order8 + fr.n_clear + 1), dim3(1024), 0, s, T, NB,
                     tile_cursor, tile_start, status, tile_order, splats, N, fr);
  TGS_CHECK_LAUNCH();
  if (G > 0) {
    const int32_t* max_list = tile_cursor + 2 * TGS_XCC * T + TGS_SCAN_WGS;
    hipLaunchKernelGGL(k_fill_bins, dim3(G), dim3(TGS_GROUP), 0, s, k, N, splats, group_base,
                       tile_cursor + TGS_XCC * T, sc.rank, (uint2*)sc.pairs, status, max_list);
    TGS_CHECK_LAUNCH();
    // The long-list classes are launched only if the caller's bound on the frame's longest list allows such lists
    // (max_list_hint < 0: no bound, every class): an unused class launch returns at once but still costs ~4.6 us of
    // the stream (profiles/r4_b_reconcile.json: 9.3 us per step at cfg3 for two launches that never had a list).
    const bool want_wg4 = max_list_hint < 0 || max_list_hint > 1024;
    const bool want_huge = max_list_hint < 0 || max_list_hint > 4096;
    const int sorted_up_to = want_huge ? 0x7fffffff : (want_wg4 ? 4096 : 1024);
    const int wave_up_to = want_wg4 ? 512 : 1024;
    hipLaunchKernelGGL(k_sort_tiles_wave, dim3(T + n_order), dim3(TGS_WAVE), 0, s, T, tile_start,
                       sc.pairs, sorted_gid, tile_order, n_order, chunk, sorted_up_to, status, sticky_overflow, wave_up_to);
    TGS_CHECK_LAUNCH();
    if (want_wg4) {
      hipLaunchKernelGGL(k_sort_tiles_wg4, dim3(T < 2048 ? T : 2048), dim3(TGS_WAVE * 4), 0, s, T, tile_start,
                         sc.pairs, sorted_gid, max_list, wave_up_to);
      TGS_CHECK_LAUNCH();
    }
    if (want_huge) {
      hipLaunchKernelGGL(k_sort_tiles_huge, dim3(T < 256 ? T : 256), dim3(1024), 0, s, T, tile_start, sc.pairs,
                         sc.fb, sorted_gid, max_list);
      TGS_CHECK_LAUNCH();
    }
  }
  return TGS_OK;
}

extern "C" int tgs_bin_sort(const TgsCamera* cam, int N, float* splats, int32_t* group_base,
                            int32_t* tile_start, int64_t tile_start_len, int32_t* tile_cursor, int32_t* sorted_gid,
                            int32_t* tile_order, int64_t capacity, void* scratch, int32_t* status,
                            int32_t* sticky_overflow, int32_t max_list_hint, void* stream) {
  TGS_CHECK_ARG(camera_ok(cam), "bad camera");
  TGS_CHECK_ARG(N >= 0 && capacity >= 0, "negative size");
  TGS_CHECK_ARG(capacity < (1ll << 31), "capacity must be < 2^31");
  TGS_CHECK_ARG(group_base && tile_start && tile_cursor && sorted_gid && scratch && status,
                "null pointer");
  TGS_CHECK_ARG(N == 0 || splats, "null splats");
  const CamK k = make_camk(cam);
  const int T = k.TW * k.TH;
  TGS_CHECK_ARG(tile_start_len >= (int64_t)T + 1 + TGS_TILE_START_SCRATCH, "tile_start buffer shorter than tgs_tile_start_len(W, H)");
  const int G = tgs_num_groups(N);
  hipStream_t s = (hipStream_t)stream;
  const BinScratch sc = carve_scratch(scratch, capacity);
  hipLaunchKernelGGL(k_clear_counters, dim3((max(TGS_XCC * T, 2) + 255) / 256), dim3(256), 0, s, tile_cursor, T, status,
                     sticky_overflow);
  TGS_CHECK_LAUNCH();
  if (G > 0) {
    hipLaunchKernelGGL(k_tile_count, dim3(G), dim3(TGS_GROUP), 0, s, k, N, splats, group_base,
                       tile_cursor, sc.rank, status, (long long)capacity, sticky_overflow);
    TGS_CHECK_LAUNCH();
  }
  return tgs_bin_finish(k, N, splats, group_base, tile_start, tile_start_len, tile_cursor, sorted_gid, tile_order,
                        capacity, scratch, status, sticky_overflow, max_list_hint, nullptr, s);
}
