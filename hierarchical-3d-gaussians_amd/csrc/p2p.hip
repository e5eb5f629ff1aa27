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
// peer's word (acquire, system scope, with s_sleep between polls and a bound on the number of polls).
#include <string.h>

#include "common.h"

namespace hgs {
namespace {

constexpr int kMaxWorld = HGS_P2P_MAX_WORLD;
constexpr long kMaxPolls = 4000000;      // x ~0.3 us per poll: about a second before the error word is set

struct P2PPtrs {
  float* buf[kMaxWorld];
  uint32_t* flag[kMaxWorld];
};

__global__ __launch_bounds__(64) void p2p_barrier_kernel(P2PPtrs p, int rank, int world, int which, uint32_t epoch) {
  const int t = threadIdx.x;
  if (t == 0) {
    __threadfence_system();
    __hip_atomic_store(p.flag[rank] + which, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (t < world && t != rank) {
    long polls = 0;
    for (;;) {
      const uint32_t v = __hip_atomic_load(p.flag[t] + which, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((int32_t)(v - epoch) >= 0) break;
      if (++polls >= kMaxPolls) {
        __hip_atomic_store(p.flag[rank] + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __syncthreads();
  __threadfence_system();
}

// [lo, hi): float4 indices of this rank's shard
__global__ __launch_bounds__(256) void p2p_reduce_kernel(P2PPtrs p, int rank, int world, size_t lo, size_t hi) {
  const size_t stride = (size_t)gridDim.x * 256;
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
  for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += stride) dst[i] = src[i];
}

}  // namespace
}  // namespace hgs

using namespace hgs;

extern "C" {

int hgs_p2p_alloc(size_t bytes, int32_t flags, void** ptr, int device) {
  if (!ptr || bytes == 0) { set_error("hgs_p2p_alloc: bad argument"); return HGS_ERR_INVALID; }
  HGS_HIP(hipSetDevice(device));
  if (flags) {
    HGS_HIP(hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached));
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
  hipLaunchKernelGGL(p2p_barrier_kernel, dim3(1), dim3(64), 0, s, p, rank, world, 0, epoch);
  if (lo < hi) hipLaunchKernelGGL(p2p_reduce_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, p, rank, world, lo, hi);
  hipLaunchKernelGGL(p2p_barrier_kernel, dim3(1), dim3(64), 0, s, p, rank, world, 1, epoch);
  hipLaunchKernelGGL(p2p_gather_kernel, dim3(blocks > 0 ? blocks : 1, world), dim3(256), 0, s, p, rank, world, base, shard, end);
  hipLaunchKernelGGL(p2p_barrier_kernel, dim3(1), dim3(64), 0, s, p, rank, world, 2, epoch);
  HGS_LAUNCH_CHECK("p2p_allreduce", s, false);
  return HGS_OK;
}

}  // extern "C"
