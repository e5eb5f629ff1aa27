// Budgeted residency of a hierarchy's attribute rows (include/hgs.h: hgs_resid_*, hgs_host_alloc): the "VRAM-budgeted
// streaming LOD" that BASELINE configs[4] names and the reference's hierarchy viewer offers as `--budget <MB>`
// (README.md:233-235; implemented in the un-vendored SIBR viewer -- nothing to restate, the mechanism here is this
// library's own).  The full attribute arrays live in pinned host memory mapped into the device's address space; the GPU
// keeps B rows in slot arrays and an int32 per Gaussian saying where (or that not).  Misses of a view are fetched by a
// kernel that READS THE HOST ROWS ITSELF (one packed 256-byte row per Gaussian, sixteen lanes per row, one float4 each:
// resid_fetch_kernel below) -- PCIe carries exactly the rows that are needed, there is no host-side gather and no
// staging copy.  Slots are recycled by age (frames since the last use) when the free list runs out.
#include "common.h"

namespace hgs {
namespace {

constexpr int kAges = 64;

// `weights` (nullable): interpolation weight of every entry of the cut.  The rasterizer's in-op LOD gather does not read
// the parent row of an entry whose weight is exactly 1 (gaussian_math.h: lod_row_gather -- 82 % of the entries of the
// 50 M-node render loop), so that parent need not be resident for this entry: its slot is reported as the node's own.
__global__ __launch_bounds__(256) void resid_mark_kernel(const int32_t* __restrict__ ri, const int32_t* __restrict__ pi,
                                                         const float* __restrict__ weights, int n, int G,
                                                         int32_t* __restrict__ slot_of, uint32_t* __restrict__ stamp,
                                                         uint32_t frame, int32_t* __restrict__ miss_ids,
                                                         uint32_t* __restrict__ counters, int32_t* __restrict__ ro,
                                                         int32_t* __restrict__ po) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool need_parent = !weights || weights[i] != 1.0f;
  const int32_t ids[2] = {ri[i], pi[i]};
  int32_t out[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int32_t id = ids[k];
    if (id < 0 || id >= G) { counters[2] = 1u; out[k] = -1; continue; }     // (benign race: every writer stores 1)
    if (k == 1 && !need_parent) { out[1] = out[0]; continue; }
    int32_t s = slot_of[id];
    if (s == -1) {                               // absent: exactly one lane of the launch queues it
      const int32_t old = atomicCAS(&slot_of[id], -1, -2);
      if (old == -1) miss_ids[atomicAdd(&counters[0], 1u)] = id;
      s = old >= 0 ? old : -2;
    }
    if (s >= 0) stamp[s] = frame;                // (same value from every writer)
    out[k] = s;
  }
  ro[i] = out[0];
  po[i] = out[1];
}

__global__ __launch_bounds__(256) void resid_remap_kernel(const int32_t* __restrict__ ri, const int32_t* __restrict__ pi,
                                                          const float* __restrict__ weights, int n,
                                                          const int32_t* __restrict__ slot_of,
                                                          int32_t* __restrict__ ro, int32_t* __restrict__ po) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t r = slot_of[ri[i]];
  ro[i] = r;
  po[i] = (!weights || weights[i] != 1.0f) ? slot_of[pi[i]] : r;
}

// occupied slots by age = min(frame - stamp, kAges - 1)
__global__ __launch_bounds__(256) void resid_age_hist_kernel(const uint32_t* __restrict__ stamp,
                                                             const int32_t* __restrict__ id_of_slot, int B, uint32_t frame,
                                                             uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[kAges];
  if (threadIdx.x < kAges) h[threadIdx.x] = 0u;
  __syncthreads();
  for (int s = blockIdx.x * 256 + threadIdx.x; s < B; s += gridDim.x * 256)
    if (id_of_slot[s] >= 0) atomicAdd(&h[min(frame - stamp[s], (uint32_t)(kAges - 1))], 1u);
  __syncthreads();
  if (threadIdx.x < kAges && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// frees every occupied slot of age >= min_age (>= 1: a row used this frame stays); counters[1] = top of the free stack
__global__ __launch_bounds__(256) void resid_evict_kernel(const uint32_t* __restrict__ stamp, int32_t* __restrict__ id_of_slot,
                                                          int32_t* __restrict__ slot_of, int B, uint32_t frame,
                                                          uint32_t min_age, int32_t* __restrict__ free_list,
                                                          uint32_t* __restrict__ counters) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= B) return;
  const int32_t id = id_of_slot[s];
  if (id < 0 || min(frame - stamp[s], (uint32_t)(kAges - 1)) < min_age) return;
  slot_of[id] = -1;
  id_of_slot[s] = -1;
  free_list[atomicAdd(&counters[1], 1u)] = s;
}

// Host layout: ONE PACKED ROW of HGS_RESID_HOST_ROW_FLOATS = 64 floats (256 B, four PCIe read requests of 64 B) per
// Gaussian --
//   [0, 3 M) SH coefficients   [48, 52) rotation   [52, 55) mean   [55, 58) scale   [58] opacity   rest: padding
// -- instead of the five attribute arrays: a row fetched from five arrays costs seven 64-byte requests for 236 useful
// bytes (measured: 21 GB/s of rows over a link that moves twice that in requests).  Sixteen lanes per missing row, four
// rows per wave: every lane loads ONE float4 of the packed row (one fully coalesced 256-byte read per row), lanes 0..11
// store the SH block, lane 12 the quaternion, lanes 13 / 14 mean, scale and opacity; the slot comes from the top of the
// free stack.  The miss list is sorted for bulk fetches (hgs/residency.py), so a wave's four rows are usually
// neighbours in host memory.
template <bool kVec>
__global__ __launch_bounds__(256) void resid_fetch_kernel(const int32_t* __restrict__ miss_ids, uint32_t m,
                                                          const int32_t* __restrict__ free_list, uint32_t free_top,
                                                          int32_t* __restrict__ slot_of, int32_t* __restrict__ id_of_slot,
                                                          uint32_t* __restrict__ stamp, uint32_t frame,
                                                          const float4* __restrict__ src, hgs_resid_rows dst, int nsh) {
  const uint32_t j = blockIdx.x * 16u + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  if (j >= m) return;
  const size_t id = (size_t)miss_ids[j];
  const size_t s = (size_t)free_list[free_top - 1u - j];
  const float4 v = src[id * (HGS_RESID_HOST_ROW_FLOATS / 4) + sub];
  if (sub < 12) {
    if (kVec) {                                    // nsh % 4 == 0: slot rows of the SH array are 16-byte aligned
      if (sub * 4 < nsh) reinterpret_cast<float4*>(dst.shs + s * nsh)[sub] = v;
    } else {
      const float c[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (sub * 4 + t < nsh) dst.shs[s * nsh + sub * 4 + t] = c[t];
    }
  } else if (sub == 12) {
    reinterpret_cast<float4*>(dst.rotations)[s] = v;
  } else if (sub == 13) {                          // floats 52..55: mean, scale.x
    dst.means3D[s * 3 + 0] = v.x; dst.means3D[s * 3 + 1] = v.y; dst.means3D[s * 3 + 2] = v.z;
    dst.scales[s * 3 + 0] = v.w;
  } else if (sub == 14) {                          // floats 56..59: scale.y, scale.z, opacity, pad
    dst.scales[s * 3 + 1] = v.x; dst.scales[s * 3 + 2] = v.y;
    dst.opacities[s] = v.z;
    slot_of[id] = (int32_t)s;
    id_of_slot[s] = (int32_t)id;
    stamp[s] = frame;
  }
}

}  // namespace
}  // namespace hgs

using namespace hgs;

extern "C" {

void* hgs_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void hgs_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

static int read_words(uint32_t* dst_host, const uint32_t* src_dev, int words, hipStream_t s) {
  HGS_HIP(hipMemcpyAsync(dst_host, src_dev, sizeof(uint32_t) * words, hipMemcpyDeviceToHost, s));
  HGS_HIP(wait_stream(s));
  return HGS_OK;
}

int hgs_resid_mark(const int32_t* render_indices, const int32_t* parent_indices, const float* weights, int32_t n, int32_t G,
                   int32_t* slot_of, uint32_t* stamp, uint32_t frame, int32_t* miss_ids, uint32_t* counters,
                   int32_t* ro, int32_t* po, uint32_t* miss_count_host, hgs_stream_t stream, int device) {
  if (!miss_count_host) { set_error("null miss_count_host"); return HGS_ERR_INVALID; }
  *miss_count_host = 0;
  if (n <= 0) return HGS_OK;
  if (!render_indices || !parent_indices || !slot_of || !stamp || !miss_ids || !counters || !ro || !po || G <= 0) {
    set_error("null argument");
    return HGS_ERR_INVALID;
  }
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  HGS_HIP(hipMemsetAsync(counters, 0, 4 * sizeof(uint32_t), s));
  hipLaunchKernelGGL(resid_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, s, render_indices, parent_indices, weights, n, G,
                     slot_of, stamp, frame, miss_ids, counters, ro, po);
  HGS_LAUNCH_CHECK("resid_mark", s, false);
  uint32_t w[4] = {0, 0, 0, 0};
  int rc = read_words(w, counters, 4, s);
  if (rc) return rc;
  // the misses queued so far are reported on the error path too: the caller has to take them back out of the queue
  *miss_count_host = w[0];
  if (w[2]) { set_error("a render / parent index lies outside [0, %d)", G); return HGS_ERR_INVALID; }
  return HGS_OK;
}

int hgs_resid_evict(uint32_t* stamp, int32_t* id_of_slot, int32_t* slot_of, int32_t B, uint32_t frame, uint32_t need,
                    int32_t* free_list, uint32_t* counters, uint32_t* free_top_inout_host, hgs_stream_t stream,
                    int device) {
  if (!stamp || !id_of_slot || !slot_of || !free_list || !counters || !free_top_inout_host || B <= 0) {
    set_error("null argument");
    return HGS_ERR_INVALID;
  }
  if (*free_top_inout_host >= need) return HGS_OK;
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint32_t* hist_dev = counters + 4;              // counters: [0] misses [1] free-stack top [2] error [4 .. 4 + kAges) ages
  HGS_HIP(hipMemsetAsync(hist_dev, 0, kAges * sizeof(uint32_t), s));
  const int grid = B < 256 * 1024 ? (B + 255) / 256 : 1024;
  hipLaunchKernelGGL(resid_age_hist_kernel, dim3(grid), dim3(256), 0, s, stamp, id_of_slot, B, frame, hist_dev);
  HGS_LAUNCH_CHECK("resid_age_hist", s, false);
  uint32_t hist[kAges];
  int rc = read_words(hist, hist_dev, kAges, s);
  if (rc) return rc;
  const uint32_t missing = need - *free_top_inout_host;
  uint64_t acc = 0;
  int min_age = 0;
  for (int a = kAges - 1; a >= 1; --a) {
    acc += hist[a];
    if (acc >= missing) { min_age = a; break; }
  }
  if (min_age == 0) {
    set_error("the view needs %u more rows than the budget of %d rows can free (rows unused this frame: %llu)", missing, B,
              (unsigned long long)acc);
    return HGS_ERR_CAPACITY;
  }
  HGS_HIP(hipMemcpyAsync(counters + 1, free_top_inout_host, sizeof(uint32_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(resid_evict_kernel, dim3((B + 255) / 256), dim3(256), 0, s, stamp, id_of_slot, slot_of, B, frame,
                     (uint32_t)min_age, free_list, counters);
  HGS_LAUNCH_CHECK("resid_evict", s, false);
  uint32_t top = 0;
  rc = read_words(&top, counters + 1, 1, s);
  if (rc) return rc;
  *free_top_inout_host = top;
  return HGS_OK;
}

int hgs_resid_fetch(const int32_t* miss_ids, uint32_t m, const int32_t* free_list, uint32_t free_top, int32_t* slot_of,
                    int32_t* id_of_slot, uint32_t* stamp, uint32_t frame, const float* host_rows_packed,
                    const hgs_resid_rows* slot_rows, int32_t M, hgs_stream_t stream, int device) {
  if (m == 0) return HGS_OK;
  if (!miss_ids || !free_list || !slot_of || !id_of_slot || !stamp || !host_rows_packed || !slot_rows) {
    set_error("null argument");
    return HGS_ERR_INVALID;
  }
  if (M < 1 || M > 16) { set_error("M = %d SH coefficients per channel: 1..16", M); return HGS_ERR_INVALID; }
  if (free_top < m) { set_error("%u free slots for %u missing rows", free_top, m); return HGS_ERR_CAPACITY; }
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the packed host rows by their device-side address
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, const_cast<float*>(host_rows_packed), 0) != hipSuccess) {
    (void)hipGetLastError();
    set_error("the packed host rows must come from hgs_host_alloc (pinned, device-mapped)");
    return HGS_ERR_INVALID;
  }
  if (((uintptr_t)d | (uintptr_t)slot_rows->rotations) & 15u) {
    set_error("packed host rows / slot rotations must be 16-byte aligned");
    return HGS_ERR_INVALID;
  }
  const float4* src = static_cast<const float4*>(d);
  const bool vec = ((M * 3) & 3) == 0 && (((uintptr_t)slot_rows->shs & 15u) == 0);
  if (vec)
    hipLaunchKernelGGL(resid_fetch_kernel<true>, dim3((m + 15) / 16), dim3(256), 0, s, miss_ids, m, free_list, free_top,
                       slot_of, id_of_slot, stamp, frame, src, *slot_rows, M * 3);
  else
    hipLaunchKernelGGL(resid_fetch_kernel<false>, dim3((m + 15) / 16), dim3(256), 0, s, miss_ids, m, free_list, free_top,
                       slot_of, id_of_slot, stamp, frame, src, *slot_rows, M * 3);
  HGS_LAUNCH_CHECK("resid_fetch", s, false);
  return HGS_OK;
}

int hgs_resid_remap(const int32_t* render_indices, const int32_t* parent_indices, const float* weights, int32_t n,
                    const int32_t* slot_of, int32_t* ro, int32_t* po, hgs_stream_t stream, int device) {
  if (n <= 0) return HGS_OK;
  if (!render_indices || !parent_indices || !slot_of || !ro || !po) { set_error("null argument"); return HGS_ERR_INVALID; }
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(resid_remap_kernel, dim3((n + 255) / 256), dim3(256), 0, s, render_indices, parent_indices, weights,
                     n, slot_of, ro, po);
  HGS_LAUNCH_CHECK("resid_remap", s, false);
  return HGS_OK;
}

}  // extern "C"
