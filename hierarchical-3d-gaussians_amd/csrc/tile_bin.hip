// K4a': grouping the (tile id, Gaussian id) instances by tile with ONE counting pass and ONE scatter pass.
//
// The reference sorts 64-bit (tile | depth) keys with a global radix sort; round-1 of this library sorted by the
// 13-bit tile id with two stable 8-bit radix passes (6 launches, 0.13 ms at 1080p / 2.7 M instances).  Stability is
// not needed: the per-tile depth sort that follows orders unique (depth bits, id) keys whatever order it is given.
// So the tile id itself is the bucket:
//   count   : a workgroup histograms its chunk of instances over ALL tiles in LDS (two 16-bit counters per word)
//             and writes the T counts of its chunk                                   table[chunk][tile]
//   colscan : per tile, exclusive scan of the chunk counts inside each of kGroups chunk groups (in place)
//   base    : exclusive scan of the tile totals = tile ranges (one workgroup); group sums -> prefix over the groups
//   scatter : pos = base[tile] + gprefix[group][tile] + table[chunk][tile] + (LDS fetch-and-add); ids only
// Two passes over the instances instead of four, 4 launches instead of 7 (the tile ranges fall out of the scan).
// Used while the 16-bit-per-tile LDS histogram fits (T <= kTileBinMaxTiles); larger grids take the radix path.
#include "common.h"

namespace hgs {
namespace {

constexpr int kTbThreads = 256;
constexpr int kGroups = 16;
constexpr int kTbStage = 8192;   // tiles whose totals the base kernel stages in LDS (1080p: 8160)

__device__ __forceinline__ uint32_t tb_n(uint32_t n_cap, const uint32_t* __restrict__ n_dev) {
  return n_dev ? min(*n_dev, n_cap) : n_cap;
}

__global__ __launch_bounds__(kTbThreads) void tb_count_kernel(const uint32_t* __restrict__ keys, uint32_t n_cap,
                                                              const uint32_t* __restrict__ n_dev, int T, uint32_t chunk,
                                                              uint32_t* __restrict__ table,
                                                              uint32_t* __restrict__ totals) {
  extern __shared__ uint32_t h[];
  const int words = (T + 1) >> 1;
  for (int w = threadIdx.x; w < words; w += kTbThreads) h[w] = 0u;
  if (blockIdx.x == 0)      // per-tile totals are accumulated with atomics by the column scan that follows
    for (int t = threadIdx.x; t < T; t += kTbThreads) totals[t] = 0u;
  __syncthreads();
  const uint32_t n = tb_n(n_cap, n_dev);
  const uint32_t base = blockIdx.x * chunk;
  const uint32_t end = min(base + chunk, n);
  for (uint32_t i = base + threadIdx.x; i < end; i += kTbThreads) {
    const uint32_t t = keys[i];
    atomicAdd(&h[t >> 1], 1u << ((t & 1u) * 16u));
  }
  __syncthreads();
  uint32_t* row = table + (size_t)blockIdx.x * T;
  for (int t = threadIdx.x; t < T; t += kTbThreads) row[t] = (h[t >> 1] >> ((t & 1) * 16)) & 0xffffu;
}

// grid (ceil(T / 256), kGroups): exclusive scan over the chunks of one group, per tile; group sums to gsum[g][t]
__global__ __launch_bounds__(kTbThreads) void tb_colscan_kernel(uint32_t* __restrict__ table, int nchunks, int cpg, int T,
                                                                uint32_t* __restrict__ gsum,
                                                                uint32_t* __restrict__ totals) {
  const int t = blockIdx.x * kTbThreads + threadIdx.x;
  if (t >= T) return;
  const int g = blockIdx.y;
  const int c0 = g * cpg, c1 = min(c0 + cpg, nchunks);
  uint32_t acc = 0;
  for (int cb = c0; cb < c1; cb += 8) {           // 8 independent loads in flight, then the 8 prefix stores
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (cb + k < c1) ? table[(size_t)(cb + k) * T + t] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (cb + k < c1) table[(size_t)(cb + k) * T + t] = acc;
      acc += v[k];
    }
  }
  gsum[(size_t)g * T + t] = acc;
  if (acc) atomicAdd(&totals[t], acc);
}

// Workgroup 0: exclusive scan of the per-tile totals = tile ranges (a lane owns `per` consecutive tiles: one
// workgroup-wide scan); resets the depth-sort class counters.  Workgroups 1..: per tile, exclusive prefix of the
// group sums over the groups, in place (independent of workgroup 0).
__global__ __launch_bounds__(1024) void tb_base_kernel(uint32_t* __restrict__ gsum, const uint32_t* __restrict__ totals,
                                                       int T, int per, uint32_t* __restrict__ base,
                                                       uint32_t* __restrict__ ranges, uint32_t* __restrict__ big) {
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {
    const int t = (blockIdx.x - 1) * 1024 + tid;
    if (t >= T) return;
    uint32_t v[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) v[g] = gsum[(size_t)g * T + t];
    uint32_t acc = 0;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      gsum[(size_t)g * T + t] = acc;
      acc += v[g];
    }
    return;
  }
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t tot_s[kTbStage];     // totals, then range starts (coalesced global access on both sides)
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { big[0] = 0; big[1] = 0; big[2] = 0; }
  const int t0 = tid * per;
  const bool staged = T <= kTbStage;
  if (staged) {
    for (int t = tid; t < T; t += 1024) tot_s[t] = totals[t];
    __syncthreads();
  }
  uint32_t mine = 0;
  for (int i = 0; i < per; ++i)
    if (t0 + i < T) mine += staged ? tot_s[t0 + i] : totals[t0 + i];
  uint32_t inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(inc, off, 64);
    if (lane >= off) inc += v;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t start = inc - mine;
  for (int w = 0; w < wave; ++w) start += wave_tot[w];
  if (staged) {
    for (int i = 0; i < per; ++i) {
      const int t = t0 + i;
      if (t >= T) break;
      const uint32_t tot = tot_s[t];
      tot_s[t] = start;
      start += tot;
    }
    __syncthreads();
    for (int t = tid; t < T; t += 1024) {
      const uint32_t st = tot_s[t], tot = totals[t];
      base[t] = st;
      reinterpret_cast<uint2*>(ranges)[t] = tot ? make_uint2(st, st + tot) : make_uint2(0u, 0u);   // empty tiles read
    }                                                                                              // (0, 0), as after
    return;                                                                                        // identifyTileRanges
  }
  for (int i = 0; i < per; ++i) {
    const int t = t0 + i;
    if (t >= T) break;
    const uint32_t tot = totals[t];
    base[t] = start;
    ranges[t * 2 + 0] = tot ? start : 0u;
    ranges[t * 2 + 1] = tot ? start + tot : 0u;
    start += tot;
  }
}

// CURSOR: LDS holds one absolute output cursor per tile (gbase + table row, loaded coalesced): one LDS fetch-and-add
// per instance and no dependent global reads.  Otherwise (tile grid too large for 4 bytes of LDS per tile): packed
// 16-bit chunk-local counters and two global reads per instance.
template <bool CURSOR>
__global__ __launch_bounds__(kTbThreads) void tb_scatter_kernel(const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ vals, uint32_t n_cap,
                                                                const uint32_t* __restrict__ n_dev, int T, uint32_t chunk,
                                                                int cpg, const uint32_t* __restrict__ table,
                                                                const uint32_t* __restrict__ gbase,
                                                                const uint32_t* __restrict__ tbase,
                                                                uint32_t* __restrict__ vals_out) {
  extern __shared__ uint32_t h[];
  const uint32_t* row = table + (size_t)blockIdx.x * T;
  const uint32_t* grow = gbase + (size_t)(blockIdx.x / cpg) * T;
  if (CURSOR) {
    for (int t = threadIdx.x; t < T; t += kTbThreads) h[t] = tbase[t] + grow[t] + row[t];
  } else {
    const int words = (T + 1) >> 1;
    for (int w = threadIdx.x; w < words; w += kTbThreads) h[w] = 0u;
  }
  __syncthreads();
  const uint32_t n = tb_n(n_cap, n_dev);
  const uint32_t base = blockIdx.x * chunk;
  const uint32_t end = min(base + chunk, n);
  for (uint32_t i = base + threadIdx.x; i < end; i += kTbThreads) {
    const uint32_t t = keys[i];
    const uint32_t gid = vals[i];
    if (CURSOR) {
      vals_out[atomicAdd(&h[t], 1u)] = gid;
    } else {
      const uint32_t sh = (t & 1u) * 16u;
      const uint32_t r = (atomicAdd(&h[t >> 1], 1u << sh) >> sh) & 0xffffu;
      vals_out[tbase[t] + grow[t] + row[t] + r] = gid;
    }
  }
}

inline uint32_t tb_chunk(int T) { return T <= 12288 ? 4096u : 16384u; }

}  // namespace

bool tile_bin_supported(int32_t T) { return T <= 32768; }

size_t tile_bin_tmp_bytes(uint32_t L, int32_t T) {
  if (!tile_bin_supported(T)) return 0;
  const uint32_t chunk = tb_chunk(T);
  const size_t nchunks = ((size_t)(L ? L : 1) + chunk - 1) / chunk;
  return align_up(nchunks * T * 4) + align_up((size_t)kGroups * T * 4) + 2 * align_up((size_t)T * 4) + kAlign;
}

// keys / vals: the emitted (tile id, Gaussian id) instances; vals_out: ids grouped by tile (unordered inside a tile);
// ranges [T,2] and the depth-sort class counters (big[0..2]) are written as well.
int launch_tile_bin(const uint32_t* keys, const uint32_t* vals, uint32_t* vals_out, void* tmp, uint32_t L_cap,
                    const uint32_t* L_dev, int32_t T, uint32_t* ranges, uint32_t* big, hipStream_t s, bool debug) {
  const uint32_t chunk = tb_chunk(T);
  const int nchunks = (int)(((size_t)L_cap + chunk - 1) / chunk);
  const int cpg = (nchunks + kGroups - 1) / kGroups;
  char* c = static_cast<char*>(tmp);
  uint32_t* table = carve<uint32_t>(c, (size_t)nchunks * T);
  uint32_t* gsum = carve<uint32_t>(c, (size_t)kGroups * T);
  uint32_t* totals = carve<uint32_t>(c, (size_t)T);
  uint32_t* tbase = carve<uint32_t>(c, (size_t)T);
  const size_t lds = (size_t)((T + 1) / 2) * 4;
  static bool attr_set = false;
  if (!attr_set) {   // up to 64 KiB of dynamic LDS at 4K
    HGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tb_count_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    HGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tb_scatter_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    HGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tb_scatter_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    attr_set = true;
  }
  hipLaunchKernelGGL(tb_count_kernel, dim3(nchunks), dim3(kTbThreads), lds, s, keys, L_cap, L_dev, T, chunk, table, totals);
  HGS_LAUNCH_CHECK("tile_bin_count", s, debug);
  hipLaunchKernelGGL(tb_colscan_kernel, dim3((T + kTbThreads - 1) / kTbThreads, kGroups), dim3(kTbThreads), 0, s, table,
                     nchunks, cpg, T, gsum, totals);
  HGS_LAUNCH_CHECK("tile_bin_colscan", s, debug);
  hipLaunchKernelGGL(tb_base_kernel, dim3(1 + (T + 1023) / 1024), dim3(1024), 0, s, gsum, totals, T, (T + 1023) / 1024, tbase,
                     ranges, big);
  HGS_LAUNCH_CHECK("tile_bin_base", s, debug);
  if (T <= 16384)
    hipLaunchKernelGGL(tb_scatter_kernel<true>, dim3(nchunks), dim3(kTbThreads), (size_t)T * 4, s, keys, vals, L_cap, L_dev,
                       T, chunk, cpg, table, gsum, tbase, vals_out);
  else
    hipLaunchKernelGGL(tb_scatter_kernel<false>, dim3(nchunks), dim3(kTbThreads), lds, s, keys, vals, L_cap, L_dev, T, chunk,
                       cpg, table, gsum, tbase, vals_out);
  HGS_LAUNCH_CHECK("tile_bin_scatter", s, debug);
  return HGS_OK;
}

}  // namespace hgs
