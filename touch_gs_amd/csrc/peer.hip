// peer.hip -- gradient exchange of the data-parallel step by DIRECT WRITES into IPC-mapped peer memory.
//
// The reference trains on one GPU (scripts/train_bunny_real.sh:52); there is no reference call pattern.  The
// default transport of touch_gs_amd.parallel is RCCL (torch.distributed "nccl").  This is the alternative SURVEY
// section 5 describes for the MI355X node: xGMI is point-to-point -- every GPU has its own link to each of the 7
// others -- so an all-gather is 7 independent one-hop copies that can all be in flight at once, and a ring
// collective (per-link bound, 2 (n-1) latency-bound steps at these message sizes: 12 MB colour blocks, a 44 MB
// geometry gradient) leaves most of the fabric idle.  Here every rank owns one receive buffer, allocated
// uncached (a peer's writes land in HBM, the owner's reads must not hit stale L2 lines) and mapped into every
// other process with hipIpcGetMemHandle / hipIpcOpenMemHandle; a push kernel stores a block into all peers'
// slots at once and the last workgroup to finish raises a flag word in each receiver with a system-scope
// release; the receiver's stream waits for the flags before the consuming kernel starts.  No collective launch,
// no ring, no host on the critical path.
//
//   tgs_peer_push         src -> the same slot of up to 8 receivers (all-gather of a colour block; all-gather of a
//                         reduced gradient slice)
//   tgs_peer_scatter      slice q of src -> receiver q (the reduce-scatter's send side)
//   tgs_peer_reduce_push  sum of `world` received slices in RANK ORDER (deterministic; every rank ends up with the
//                         same bits) -> the same slot of every receiver
//   tgs_peer_wait         the stream waits until `n` flag words have reached `seq`
//
// The caller owns all memory (tgs_peer_alloc is a thin wrapper around hipExtMallocWithFlags + the IPC handle so
// that a host language without HIP bindings can use it; nothing else is allocated here).
#include <string.h>
#include "tgs_common.h"

namespace {

constexpr int PEER_MAX = 8;

struct PeerDst {
  float* dst[PEER_MAX];
  int32_t* flag[PEER_MAX];
};

// The last workgroup of a launch to get here (ticket) publishes the flags.  Ordering: the receive buffers are
// UNCACHED memory, so every data store is written through to its destination and its completion (vmcnt) means it has
// arrived; a workgroup waits for the completion of its own stores (workgroup-scope release + barrier: s_waitcnt, no
// cache maintenance) before it takes a ticket, and the one wave that sees the last ticket issues the single
// system-scope release of the launch.  ASSUMPTION (never exercised across GPUs -- all ranks of every run so far shared
// one device): the completion of a store to uncached memory means it is performed at its destination, also across
// xGMI.  parallel.PeerExchange's DEFAULT does not rely on it: it passes no flags to the data kernels and raises them with
// a separate one-wave launch (tgs_peer_signal) behind the kernel boundary; TGS_PEER_SAFE_FLAGS=0 selects this in-kernel form.  (A system- or agent-scope fence per wave -- the textbook form -- writes back
// the XCD's whole L2 every time: 8192 waves doing that took the 12 MB push to 165 us and the 44 MB rank-order sum to
// 525 us; this form runs at copy speed.)
__device__ __forceinline__ void publish(const PeerDst& d, int n, int32_t seq, int32_t* ticket) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = gridDim.x * gridDim.y;
    if (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
      for (int i = 0; i < n; i++)
        if (d.flag[i]) __hip_atomic_store(d.flag[i], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// src[0 .. n) -> dst[i][0 .. n) for every receiver i, in units of T (float4; float for blocks whose size or slot
// offset is not a multiple of 16 bytes -- a chunk of 3 rows + 4 floats with rows % 4 != 0)
template <typename T>
__global__ __launch_bounds__(256) void k_peer_push(PeerDst d, int n, const T* __restrict__ src, size_t nT,
                                                   int32_t seq, int32_t* ticket) {
  // four independent loads in flight per thread (one load per iteration left the copy at 1 TB/s)
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nT; i += 4 * stride) {
    T v[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i + u * stride < nT) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i + u * stride < nT) {
#pragma unroll
        for (int q = 0; q < PEER_MAX; q++)
          if (q < n) reinterpret_cast<T*>(d.dst[q])[i + u * stride] = v[u];
      }
  }
  publish(d, n, seq, ticket);
}

// src[q * slice4 .. ) -> dst[q][0 .. len_q): blockIdx.y = receiver
__global__ __launch_bounds__(256) void k_peer_scatter(PeerDst d, int n, const float4* __restrict__ src, size_t slice4,
                                                      size_t total4, int32_t seq, int32_t* ticket) {
  const int q = blockIdx.y;
  const size_t b = (size_t)q * slice4, e = min(b + slice4, total4);
  float4* __restrict__ o = reinterpret_cast<float4*>(d.dst[q]);
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = b + (size_t)blockIdx.x * 256 + threadIdx.x; i < e; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i + u * stride < e) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i + u * stride < e) o[i + u * stride - b] = v[u];
  }
  publish(d, n, seq, ticket);
}

struct PeerSrc { const float* src[PEER_MAX]; };

// out[i] = ((src0[i] + src1[i]) + src2[i]) + ...   in rank order, written to every receiver
__global__ __launch_bounds__(256) void k_peer_reduce_push(PeerSrc s, int world, PeerDst d, int n, size_t n4, int32_t seq,
                                                          int32_t* ticket) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 v[PEER_MAX];          // all ranks' values in flight at once, then summed in rank order
#pragma unroll
    for (int r = 0; r < PEER_MAX; r++)
      if (r < world) v[r] = reinterpret_cast<const float4*>(s.src[r])[i];
    float4 a = v[0];
#pragma unroll
    for (int r = 1; r < PEER_MAX; r++)
      if (r < world) { a.x += v[r].x; a.y += v[r].y; a.z += v[r].z; a.w += v[r].w; }
#pragma unroll
    for (int q = 0; q < PEER_MAX; q++)
      if (q < n) reinterpret_cast<float4*>(d.dst[q])[i] = a;
  }
  publish(d, n, seq, ticket);
}

// flags[i] = seq with a system-scope release, as a launch of its own: everything earlier launches on the stream
// stored is complete at the kernel boundary whatever the memory type (the conservative form of publish())
__global__ __launch_bounds__(64) void k_peer_signal(PeerDst d, int n, int32_t seq) {
  const int i = threadIdx.x;
  if (i < n && d.flag[i]) __hip_atomic_store(d.flag[i], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct PeerFlags { const int32_t* flag[2 * PEER_MAX]; };

// One wave: lane i polls flag i (system-scope acquire) with a wall-clock bound; a timeout raises *err instead of
// hanging the queue (the caller checks it at its next sync point).
// On a timeout the wait gives up, records who was missing (*err) and POISONS the step: the two optional words (the
// caller's sticky overflow word and the agreed verdict word the optimizer kernels are guarded by) are set to 1, so the
// kernels queued behind this one do not consume the stale receive slots (a timed-out wait used to let the SH
// and geometry Adam run on them until the host looked at *err).
__global__ __launch_bounds__(64) void k_peer_wait(PeerFlags f, int n, int32_t seq, int32_t* err, long long timeout_ticks,
                                                  int32_t* poison_a, int32_t* poison_b) {
  const int i = threadIdx.x;
  if (i < n && f.flag[i]) {
    const long long t0 = (long long)__builtin_readcyclecounter();
    // relaxed polls (the flag word is uncached: every load goes to memory), ONE acquire after the last
    while (__hip_atomic_load(f.flag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq < 0) {
      __builtin_amdgcn_s_sleep(16);
      if ((long long)__builtin_readcyclecounter() - t0 > timeout_ticks) {
        if (err) *err = 1 + i;
        if (poison_a) *poison_a = 1;
        if (poison_b) *poison_b = 1;
        break;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope: the kernels behind this one read the payload
}

int fill_dst(PeerDst& d, int n, void* const* dsts, int32_t* const* flags) {
  for (int i = 0; i < PEER_MAX; i++) {
    d.dst[i] = i < n ? (float*)dsts[i] : nullptr;
    d.flag[i] = (i < n && flags) ? flags[i] : nullptr;
  }
  return 0;
}

int grid_for(size_t n4) { const size_t g = (n4 + 255) / 256; return (int)(g < 2048 ? g : 2048); }

}  // namespace

extern "C" int tgs_peer_alloc(size_t bytes, int allow_kinds, void** dptr, unsigned char* handle64, int* kind_out) {
  TGS_CHECK_ARG(dptr && handle64 && bytes > 0, "null pointer / zero size");
  TGS_CHECK_ARG((allow_kinds & (TGS_PEER_MEM_UNCACHED | TGS_PEER_MEM_FINEGRAINED | TGS_PEER_MEM_PLAIN)) != 0, "no memory kind allowed");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  void* p = nullptr;
  int kind = 0;
  // uncached: a peer's stores arrive in this GPU's HBM behind its L2 -- what the transport's ordering argument is
  // written for; fine-grained: coherent at system scope, same argument; plain (cached) memory only on request: the
  // owner's L2 may then hold stale lines of a slot a peer has written, the caller must publish behind kernel
  // boundaries AND invalidate before reading.  The kind obtained is reported, never silently downgraded.
  if ((allow_kinds & TGS_PEER_MEM_UNCACHED) && hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) == hipSuccess) kind = TGS_PEER_MEM_UNCACHED;
  if (!kind) (void)hipGetLastError();
  if (!kind && (allow_kinds & TGS_PEER_MEM_FINEGRAINED) && hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess) kind = TGS_PEER_MEM_FINEGRAINED;
  if (!kind) (void)hipGetLastError();
  if (!kind && (allow_kinds & TGS_PEER_MEM_PLAIN) && hipMalloc(&p, bytes) == hipSuccess) kind = TGS_PEER_MEM_PLAIN;
  if (!kind) {
    (void)hipGetLastError();
    tgs_set_error("tgs_peer_alloc: none of the allowed memory kinds (mask %d) could be allocated (%zu bytes)", allow_kinds, bytes);
    return TGS_E_HIP;
  }
  if (kind_out) *kind_out = kind;
  TGS_HIP(hipMemset(p, 0, bytes));
  TGS_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  TGS_HIP(hipIpcGetMemHandle(&h, p));
  memcpy(handle64, &h, 64);
  *dptr = p;
  return TGS_OK;
}

extern "C" int tgs_peer_open(const unsigned char* handle64, void** dptr) {
  TGS_CHECK_ARG(dptr && handle64, "null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  TGS_HIP(hipIpcOpenMemHandle(dptr, h, hipIpcMemLazyEnablePeerAccess));
  return TGS_OK;
}

extern "C" int tgs_peer_close(void* dptr) {
  if (dptr) TGS_HIP(hipIpcCloseMemHandle(dptr));
  return TGS_OK;
}

extern "C" int tgs_peer_free(void* dptr) {
  if (dptr) TGS_HIP(hipFree(dptr));
  return TGS_OK;
}

extern "C" int tgs_peer_push(int n_dst, void* const* dsts, int32_t* const* flags, const void* src, size_t bytes,
                             int32_t seq, int32_t* ticket, void* stream) {
  TGS_CHECK_ARG(n_dst >= 0 && n_dst <= PEER_MAX, "at most 8 receivers");
  TGS_CHECK_ARG(bytes % 4 == 0 && ((uintptr_t)src % 4) == 0, "size and pointers must be multiples of 4 bytes");
  TGS_CHECK_ARG(ticket && (n_dst == 0 || (dsts && src)), "null pointer");
  if (n_dst == 0) return TGS_OK;
  PeerDst d;
  fill_dst(d, n_dst, dsts, flags);
  bool vec = bytes % 16 == 0 && ((uintptr_t)src % 16) == 0;
  for (int i = 0; i < n_dst; i++) vec = vec && ((uintptr_t)dsts[i] % 16) == 0;
  if (vec)
    hipLaunchKernelGGL(k_peer_push<float4>, dim3(max(grid_for(bytes / 16), 1)), dim3(256), 0, (hipStream_t)stream, d, n_dst,
                       (const float4*)src, bytes / 16, seq, ticket);
  else
    hipLaunchKernelGGL(k_peer_push<float>, dim3(max(grid_for(bytes / 4), 1)), dim3(256), 0, (hipStream_t)stream, d, n_dst,
                       (const float*)src, bytes / 4, seq, ticket);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_peer_scatter(int n_dst, void* const* dsts, int32_t* const* flags, const void* src,
                                size_t slice_bytes, size_t total_bytes, int32_t seq, int32_t* ticket, void* stream) {
  TGS_CHECK_ARG(n_dst >= 1 && n_dst <= PEER_MAX, "1..8 receivers");
  TGS_CHECK_ARG(slice_bytes % 16 == 0 && total_bytes % 16 == 0 && ((uintptr_t)src % 16) == 0, "sizes must be multiples of 16 bytes");
  TGS_CHECK_ARG(ticket && dsts && src && slice_bytes * (size_t)n_dst >= total_bytes, "null pointer / slices do not cover the buffer");
  PeerDst d;
  fill_dst(d, n_dst, dsts, flags);
  hipLaunchKernelGGL(k_peer_scatter, dim3(max(grid_for(slice_bytes / 16 / 4), 1), n_dst), dim3(256), 0, (hipStream_t)stream, d,
                     n_dst, (const float4*)src, slice_bytes / 16, total_bytes / 16, seq, ticket);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_peer_reduce_push(int world, const void* const* srcs, int n_dst, void* const* dsts,
                                    int32_t* const* flags, size_t bytes, int32_t seq, int32_t* ticket, void* stream) {
  TGS_CHECK_ARG(world >= 1 && world <= PEER_MAX && n_dst >= 1 && n_dst <= PEER_MAX, "1..8 ranks");
  TGS_CHECK_ARG(bytes % 16 == 0 && ticket && srcs && dsts, "bad size / null pointer");
  PeerDst d;
  fill_dst(d, n_dst, dsts, flags);
  PeerSrc s;
  for (int i = 0; i < PEER_MAX; i++) s.src[i] = i < world ? (const float*)srcs[i] : nullptr;
  const size_t n4 = bytes / 16;
  hipLaunchKernelGGL(k_peer_reduce_push, dim3(max(grid_for(n4), 1)), dim3(256), 0, (hipStream_t)stream, s, world, d, n_dst,
                     n4, seq, ticket);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_peer_signal(int n, int32_t* const* flags, int32_t seq, void* stream) {
  TGS_CHECK_ARG(n >= 0 && n <= PEER_MAX && (n == 0 || flags), "at most 8 flags");
  if (n == 0) return TGS_OK;
  PeerDst d;
  for (int i = 0; i < PEER_MAX; i++) { d.dst[i] = nullptr; d.flag[i] = i < n ? flags[i] : nullptr; }
  hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, (hipStream_t)stream, d, n, seq);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}

extern "C" int tgs_peer_wait(int n, const int32_t* const* flags, int32_t seq, int32_t* err, float timeout_s,
                             int32_t* poison_a, int32_t* poison_b, void* stream) {
  TGS_CHECK_ARG(n >= 0 && n <= 2 * PEER_MAX && (n == 0 || flags), "at most 16 flags");
  if (n == 0) return TGS_OK;
  PeerFlags f;
  for (int i = 0; i < 2 * PEER_MAX; i++) f.flag[i] = i < n ? flags[i] : nullptr;
  // __builtin_readcyclecounter = s_memtime: the shader clock (~2.4 GHz nominal)
  const long long ticks = (long long)((timeout_s > 0 ? timeout_s : 20.0) * 2.0e9);
  hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, f, n, seq, err, ticks, poison_a, poison_b);
  TGS_CHECK_LAUNCH();
  return TGS_OK;
}
