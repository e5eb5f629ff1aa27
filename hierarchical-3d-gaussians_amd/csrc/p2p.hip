// Direct (two-shot) SUM all-reduce over peer pointers (include/hgs.h, hgs_p2p_*): the exchange step of per-view data
// parallelism written against xGMI's point-to-point topology instead of a ring.
//
//   barrier A   every rank's bucket is complete (its producer kernels precede this call on the stream)
//   reduce      rank r: bucket_r[shard r] = sum over k = 0..world-1 of bucket_k[shard r]   (reads all peers at once)
//   barrier B   every shard is reduced
//   gather      rank r: bucket_r[shard k] = bucket_k[shard k] for k != r
//   barrier C   nobody reads a peer's bucket any more: the next step may overwrite it
//
// A barrier = store the call's epoch into the own flag word (release, system scope), then one lane per peer polls the
// peer's word (acquire, system scope, with s_sleep between polls and a WALL-CLOCK bound, HGS_P2P_TIMEOUT_S, default 60 s).
//
// A timeout is FATAL, not advisory: the rank sets its error word (flag[3], sticky) and from then on the reduce and
// gather kernels of that rank write NaN instead of sums, so a peer that fell behind by more than the bound (or died)
// can never produce silently wrong gradients -- every rank that waited for it ends with a NaN bucket, and
// DirectAllReduce.check() (called periodically by the training step) turns the word into an exception.
//
// Visibility across devices (the buckets are ordinary coarse-grained hipMalloc memory unless HGS_P2P_FINEGRAINED=1):
// every transfer of ownership happens at a KERNEL BOUNDARY.  The producers of a bucket (K8, the SH backward) are whole
// kernels that precede barrier A on the stream; HIP's dispatch packets carry system-scope release / acquire fences, so
// when a producer kernel has completed all eight XCD L2s have written its lines back to HBM (the same mechanism that
// makes one kernel's output visible to the next kernel on ANOTHER XCD of the same device: MI355X_MICROARCH.md, per-XCD
// L2s are not coherent with each other), and the reduce / gather kernels, being separate launches behind a barrier
// kernel, start with an L2 invalidate and therefore read the peers' lines from memory, not from a stale cached copy.
// Inside a kernel only the flag words are exchanged: they live in uncached memory and are accessed with system-scope
// atomics.  What this argument cannot replace is a run on real xGMI: HGS_P2P_VERIFY=N (hgs/dp.py) checks the first N
// exchanges of a job against torch.distributed's all-reduce and raises on a mismatch.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace hgs {
namespace {

constexpr int kMaxWorld = HGS_P2P_MAX_WORLD;
constexpr float kNaN = __builtin_nanf("");

struct P2PPtrs {
  float* buf[kMaxWorld];
  uint32_t* flag[kMaxWorld];
};

// timeout_ticks: wall_clock64() ticks (constant 100 MHz counter)
__global__ __launch_bounds__(64) void p2p_barrier_kernel(P2PPtrs p, int rank, int world, int which, uint32_t epoch,
                                                         unsigned long long timeout_ticks) {
  const int t = threadIdx.x;
  if (t == 0) {
    __threadfence_system();
    __hip_atomic_store(p.flag[rank] + which, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (t < world && t != rank) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
      const uint32_t v = __hip_atomic_load(p.flag[t] + which, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((int32_t)(v - epoch) >= 0) break;
      if (wall_clock64() - t0 >= timeout_ticks) {
        __hip_atomic_store(p.flag[rank] + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // sticky: see above
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __syncthreads();
  __threadfence_system();
}

__device__ __forceinline__ bool p2p_failed(const P2PPtrs& p, int rank) {
  return __hip_atomic_load(p.flag[rank] + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
}

// [lo, hi): float4 indices of this rank's shard
__global__ __launch_bounds__(256) void p2p_reduce_kernel(P2PPtrs p, int rank, int world, size_t lo, size_t hi) {
  const size_t stride = (size_t)gridDim.x * 256;
  if (p2p_failed(p, rank)) {      // a barrier timed out: poison instead of summing possibly incomplete buckets
    for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride)
      reinterpret_cast<float4*>(p.buf[rank])[i] = make_float4(kNaN, kNaN, kNaN, kNaN);
    return;
  }
  for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride) {
    float4 s = reinterpret_cast<const float4*>(p.buf[0])[i];
    for (int k = 1; k < world; ++k) {
      const float4 v = reinterpret_cast<const float4*>(p.buf[k])[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(p.buf[rank])[i] = s;
  }
}

// copies the reduced shards of the other ranks; blockIdx.y = shard
__global__ __launch_bounds__(256) void p2p_gather_kernel(P2PPtrs p, int rank, int world, size_t base, size_t shard,
                                                         size_t end) {
  const int k = blockIdx.y;
  if (k == rank) return;
  const size_t lo = base + (size_t)k * shard, hi = min(lo + shard, end);
  const size_t stride = (size_t)gridDim.x * 256;
  const float4* src = reinterpret_cast<const float4*>(p.buf[k]);
  float4* dst = reinterpret_cast<float4*>(p.buf[rank]);
  if (p2p_failed(p, rank)) {
    for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride) dst[i] = make_float4(kNaN, kNaN, kNaN, kNaN);
    return;
  }
  for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride) dst[i] = src[i];
}

}  // namespace
}  // namespace hgs

using namespace hgs;

extern "C" {

int hgs_p2p_alloc(size_t bytes, int32_t flags, void** ptr, int device) {
  if (!ptr || bytes == 0) { set_error("hgs_p2p_alloc: bad argument"); return HGS_ERR_INVALID; }
  HGS_HIP(hipSetDevice(device));
  if (flags & 1) {            // flag words: never cached
    HGS_HIP(hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached));
  } else if (flags & 2) {     // bucket in fine-grained (device-coherent) memory: see the visibility note above
    HGS_HIP(hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained));
  } else {
    HGS_HIP(hipMalloc(ptr, bytes));
  }
  HGS_HIP(hipMemset(*ptr, 0, bytes));
  HGS_HIP(hipDeviceSynchronize());
  return HGS_OK;
}

int hgs_p2p_free(void* ptr, int device) {
  HGS_HIP(hipSetDevice(device));
  if (ptr) HGS_HIP(hipFree(ptr));
  return HGS_OK;
}

int hgs_p2p_export(void* ptr, uint8_t handle[HGS_P2P_HANDLE_BYTES], int device) {
  static_assert(sizeof(hipIpcMemHandle_t) <= HGS_P2P_HANDLE_BYTES, "IPC handle does not fit");
  if (!ptr || !handle) { set_error("hgs_p2p_export: null argument"); return HGS_ERR_INVALID; }
  HGS_HIP(hipSetDevice(device));
  hipIpcMemHandle_t h;
  HGS_HIP(hipIpcGetMemHandle(&h, ptr));
  memset(handle, 0, HGS_P2P_HANDLE_BYTES);
  memcpy(handle, &h, sizeof(h));
  return HGS_OK;
}

int hgs_p2p_open(const uint8_t handle[HGS_P2P_HANDLE_BYTES], void** ptr, int device) {
  if (!ptr || !handle) { set_error("hgs_p2p_open: null argument"); return HGS_ERR_INVALID; }
  HGS_HIP(hipSetDevice(device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  HGS_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
  return HGS_OK;
}

int hgs_p2p_close(void* ptr, int device) {
  HGS_HIP(hipSetDevice(device));
  if (ptr) HGS_HIP(hipIpcCloseMemHandle(ptr));
  return HGS_OK;
}

int hgs_p2p_allreduce_sum(int32_t rank, int32_t world, void* const* bufs, void* const* flag_blocks, size_t offset,
                          size_t n, uint32_t epoch, hgs_stream_t stream, int device) {
  if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world || !bufs || !flag_blocks) {
    set_error("hgs_p2p_allreduce_sum: bad rank / world (%d / %d) or null tables", rank, world);
    return HGS_ERR_INVALID;
  }
  if ((offset & 3) || (n & 3)) { set_error("hgs_p2p_allreduce_sum: offset and n must be multiples of 4 floats"); return HGS_ERR_INVALID; }
  P2PPtrs p;
  for (int k = 0; k < kMaxWorld; ++k) {
    p.buf[k] = k < world ? static_cast<float*>(bufs[k]) : nullptr;
    p.flag[k] = k < world ? static_cast<uint32_t*>(flag_blocks[k]) : nullptr;
    if (k < world && (!p.buf[k] || !p.flag[k])) { set_error("hgs_p2p_allreduce_sum: null pointer for rank %d", k); return HGS_ERR_INVALID; }
    if (k < world && (reinterpret_cast<uintptr_t>(p.buf[k]) & 15)) { set_error("hgs_p2p_allreduce_sum: bucket of rank %d is not 16-byte aligned", k); return HGS_ERR_INVALID; }
  }
  if (n == 0 || world == 1) return HGS_OK;
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t base = offset >> 2, vecs = n >> 2, end = base + vecs;      // float4 units
  const size_t shard = (vecs + world - 1) / world;
  const size_t lo = base + (size_t)rank * shard, hi = lo + shard < end ? lo + shard : end;
  const int blocks = (int)((shard + 255) / 256 < 2048 ? (shard + 255) / 256 : 2048);
  static const unsigned long long timeout_ticks = []() {
    const char* e = getenv("HGS_P2P_TIMEOUT_S");
    double sec = e ? atof(e) : 60.0;
    if (!(sec > 0.0)) sec = 60.0;
    return (unsigned long long)(sec * 1.0e8);
  }();
  hipLaunchKernelGGL(p2p_barrier_kernel, dim3(1), dim3(64), 0, s, p, rank, world, 0, epoch, timeout_ticks);
  if (lo < hi) hipLaunchKernelGGL(p2p_reduce_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, p, rank, world, lo, hi);
  hipLaunchKernelGGL(p2p_barrier_kernel, dim3(1), dim3(64), 0, s, p, rank, world, 1, epoch, timeout_ticks);
  hipLaunchKernelGGL(p2p_gather_kernel, dim3(blocks > 0 ? blocks : 1, world), dim3(256), 0, s, p, rank, world, base, shard, end);
  hipLaunchKernelGGL(p2p_barrier_kernel, dim3(1), dim3(64), 0, s, p, rank, world, 2, epoch, timeout_ticks);
  HGS_LAUNCH_CHECK("p2p_allreduce", s, false);
  return HGS_OK;
}

}  // extern "C"
