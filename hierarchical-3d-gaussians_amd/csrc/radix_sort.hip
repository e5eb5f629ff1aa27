// K4: stable LSD radix sort of (u64 key, u32 value) pairs over bits [0, end_bit).
//
// Hand-written for wave64: the per-digit stable rank inside a wave comes from
// 8 ballots (one per digit bit) + a popcount, no shared-memory atomics, so equal keys
// keep their input order -- the tie rule the (tile|depth) sort is defined with
// (SURVEY.md App. A.7).  Three kernels per 8-bit pass:
//   histogram  : per-workgroup digit counts            -> counts[digit][block]
//   scan       : per digit, exclusive scan over blocks -> counts (in place) + totals[digit]
//   scatter    : recompute ranks, write pairs to their final slot of this pass
// Only the significant key bits are sorted (32 depth bits + ceil(log2 tiles)).
#include "common.h"

namespace hgs {
namespace {

constexpr int kRsThreads = 256;
constexpr int kRsWaves = kRsThreads / 64;
constexpr int kRadix = 256;
// Keys per lane.  The scatter kernel is a chain of latency-bound phases (offset loads, key loads,
// ranking, scattered stores), so what matters below ~10 M keys is how many workgroups overlap on a
// CU, not work per workgroup: 4 keys per lane (1024 per workgroup) keeps ~10 workgroups per CU in
// flight at the 1-3 M keys of a 1080p frame; 16 per lane amortises the per-workgroup offset loads
// once there are enough keys to fill the chip anyway.
constexpr int kRsItemsSmall = 4;
constexpr int kRsItemsLarge = 16;
constexpr uint32_t kRsLargeThreshold = 8u << 20;
inline int rs_items(uint32_t n) { return n >= kRsLargeThreshold ? kRsItemsLarge : kRsItemsSmall; }

template <typename K>
__device__ __forceinline__ uint32_t digit_of(K k, int shift, uint32_t mask) {
  return (uint32_t)(k >> shift) & mask;
}

// n_dev (optional): actual element count in device memory, clamped to the launch-time capacity n.  Lets a
// caller that only knows an upper bound enqueue the sort without a host round trip; workgroups beyond the
// actual count contribute empty histograms.
__device__ __forceinline__ uint32_t actual_n(uint32_t n_cap, const uint32_t* __restrict__ n_dev) {
  return n_dev ? min(*n_dev, n_cap) : n_cap;
}

template <typename K, int ITEMS>
__global__ __launch_bounds__(kRsThreads) void rs_histogram_kernel(const K* __restrict__ keys, uint32_t n_cap,
                                                                  const uint32_t* __restrict__ n_dev,
                                                                  int shift, uint32_t mask, uint32_t nblk,
                                                                  uint32_t* __restrict__ counts) {
  __shared__ uint32_t hist[kRadix];
  const uint32_t n = actual_n(n_cap, n_dev);
  hist[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)(kRsThreads * ITEMS);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = base + i * kRsThreads + threadIdx.x;
    if (idx < n) atomicAdd(&hist[digit_of(keys[idx], shift, mask)], 1u);
  }
  __syncthreads();
  counts[threadIdx.x * nblk + blockIdx.x] = hist[threadIdx.x];
}

// One workgroup per digit: exclusive scan of counts[digit][0..nblk) in place.
__global__ __launch_bounds__(256) void rs_scan_kernel(uint32_t* __restrict__ counts, uint32_t nblk,
                                                      uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t carry_s;
  uint32_t* row = counts + (size_t)blockIdx.x * nblk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblk; base += 256) {
    const uint32_t i = base + tid;
    const uint32_t v = (i < nblk) ? row[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
    const uint32_t carry = carry_s;
    if (i < nblk) row[i] = carry + wbase + inc - v;
    __syncthreads();
    if (tid == 255) carry_s = carry + wbase + inc;
    __syncthreads();
  }
  if (tid == 0) totals[blockIdx.x] = carry_s;
}

// vals_in == nullptr means "values are the input positions" (iota), saving a pass over an index array.
template <typename K, int ITEMS>
__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const K* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,
                                                                K* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, uint32_t n_cap,
                                                                const uint32_t* __restrict__ n_dev,
                                                                int shift, uint32_t mask, uint32_t nblk,
                                                                const uint32_t* __restrict__ counts,
                                                                const uint32_t* __restrict__ totals) {
  const uint32_t n = actual_n(n_cap, n_dev);
  __shared__ uint32_t wave_hist[kRsWaves][kRadix];   // running per-wave digit counts, then offsets
  __shared__ uint32_t digit_base[kRadix];
  __shared__ uint32_t scan_tmp[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // global base of every digit = exclusive scan of totals (256 entries, one per thread)
  {
    const uint32_t v = totals[tid];
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (lane == 63) scan_tmp[wave] = inc;
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) wave_hist[w][tid] = 0;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += scan_tmp[w];
    digit_base[tid] = wbase + inc - v + counts[(size_t)tid * nblk + blockIdx.x];
  }
  __syncthreads();

  // each wave owns a contiguous run of 64*ITEMS keys; item i of lane l sits at run + i*64 + l
  const uint32_t run = blockIdx.x * (uint32_t)(kRsThreads * ITEMS) + wave * (64u * ITEMS);
  K key[ITEMS];
  uint32_t rank[ITEMS];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = run + i * 64 + lane;
    const bool valid = idx < n;
    key[i] = valid ? keys_in[idx] : (K)~(K)0;
    const uint32_t d = digit_of(key[i], shift, mask);
    // lanes holding the same digit (invalid lanes never match a valid one)
    uint64_t same = __ballot(valid);
    if (!valid) same = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bal = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? bal : ~bal;
    }
    volatile uint32_t* wh = &wave_hist[wave][0];
    const uint32_t before = wh[d];                       // all lanes of a group read the same value
    rank[i] = before + (uint32_t)__popcll(same & lt_mask);
    // the lowest lane of each group publishes the new running count; same-wave LDS
    // accesses execute in program order, so the next item's read sees it.
    if (valid && (same & lt_mask) == 0) wh[d] = before + (uint32_t)__popcll(same);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // turn per-wave counts into per-wave output offsets: digit_base + counts of earlier waves
  {
    uint32_t acc = digit_base[tid];
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) {
      const uint32_t c = wave_hist[w][tid];
      wave_hist[w][tid] = acc;
      acc += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const uint32_t idx = run + i * 64 + lane;
    if (idx < n) {
      const uint32_t d = digit_of(key[i], shift, mask);
      const uint32_t pos = wave_hist[wave][d] + rank[i];
      keys_out[pos] = key[i];
      vals_out[pos] = vals_in ? vals_in[idx] : idx;
    }
  }
}

template <typename K>
struct SortTmp {
  K* keys_alt;
  uint32_t* vals_alt;
  uint32_t* counts;
  uint32_t* totals;
};

inline uint32_t rs_blocks(uint32_t n, int items) { return (n + kRsThreads * items - 1) / (kRsThreads * items); }

template <typename K>
inline SortTmp<K> carve_sort_tmp(void* tmp, uint32_t n) {
  char* p = static_cast<char*>(tmp);
  SortTmp<K> t;
  t.keys_alt = carve<K>(p, n);
  t.vals_alt = carve<uint32_t>(p, n);
  t.counts = carve<uint32_t>(p, (size_t)kRadix * rs_blocks(n, rs_items(n)));
  t.totals = carve<uint32_t>(p, kRadix);
  return t;
}

template <typename K, int ITEMS>
int sort_pairs_t(const K* keys_in, const uint32_t* vals_in, K* keys_out, uint32_t* vals_out, void* tmp, uint32_t n,
                 const uint32_t* n_dev, int end_bit, hipStream_t s, bool debug) {
  if (n == 0) return HGS_OK;
  const int maxbit = (int)sizeof(K) * 8;
  if (end_bit < 1) end_bit = 1;
  if (end_bit > maxbit) end_bit = maxbit;
  const int passes = (end_bit + 7) / 8;
  const SortTmp<K> t = carve_sort_tmp<K>(tmp, n);
  const uint32_t nblk = rs_blocks(n, ITEMS);
  // ping-pong between (out) and (alt) such that the LAST pass writes (out); pass 0 reads (in).
  const K* src_k = keys_in;
  const uint32_t* src_v = vals_in;
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) % 2) == 0;
    K* dst_k = to_out ? keys_out : t.keys_alt;
    uint32_t* dst_v = to_out ? vals_out : t.vals_alt;
    const int shift = p * 8;
    const int bits = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
    const uint32_t mask = (1u << bits) - 1u;
    hipLaunchKernelGGL((rs_histogram_kernel<K, ITEMS>), dim3(nblk), dim3(kRsThreads), 0, s, src_k, n, n_dev, shift, mask, nblk,
                       t.counts);
    HGS_LAUNCH_CHECK("rs_histogram", s, debug);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(kRadix), dim3(256), 0, s, t.counts, nblk, t.totals);
    HGS_LAUNCH_CHECK("rs_scan", s, debug);
    hipLaunchKernelGGL((rs_scatter_kernel<K, ITEMS>), dim3(nblk), dim3(kRsThreads), 0, s, src_k, src_v, dst_k, dst_v, n,
                       n_dev, shift, mask, nblk, t.counts, t.totals);
    HGS_LAUNCH_CHECK("rs_scatter", s, debug);
    src_k = dst_k;
    src_v = dst_v;
  }
  return HGS_OK;
}

}  // namespace

size_t sort_tmp_bytes(uint32_t n) {   // sized for 64-bit keys (covers the 32-bit sorts too)
  return align_up((size_t)n * 8) + align_up((size_t)n * 4) +
         align_up((size_t)kRadix * rs_blocks(n, rs_items(n)) * 4) + align_up(kRadix * 4) + kAlign;
}

int sort_pairs(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
               void* tmp, uint32_t n, int end_bit, hipStream_t s, bool debug) {
  return rs_items(n) == kRsItemsLarge
             ? sort_pairs_t<uint64_t, kRsItemsLarge>(keys_in, vals_in, keys_out, vals_out, tmp, n, nullptr, end_bit, s, debug)
             : sort_pairs_t<uint64_t, kRsItemsSmall>(keys_in, vals_in, keys_out, vals_out, tmp, n, nullptr, end_bit, s, debug);
}

int sort_pairs32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                 void* tmp, uint32_t n, const uint32_t* n_dev, int end_bit, hipStream_t s, bool debug) {
  return rs_items(n) == kRsItemsLarge
             ? sort_pairs_t<uint32_t, kRsItemsLarge>(keys_in, vals_in, keys_out, vals_out, tmp, n, n_dev, end_bit, s, debug)
             : sort_pairs_t<uint32_t, kRsItemsSmall>(keys_in, vals_in, keys_out, vals_out, tmp, n, n_dev, end_bit, s, debug);
}

}  // namespace hgs
