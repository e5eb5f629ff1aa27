// K4a': grouping the (tile id, Gaussian id) instances by tile with ONE counting pass and ONE scatter pass, PER TILE BAND.
//
// The reference sorts 64-bit (tile | depth) keys with a global radix sort; round 1 of this library sorted by the
// 13-bit tile id with two stable 8-bit radix passes (6 launches, 0.13 ms at 1080p / 2.7 M instances); round 2 replaced the
// sort by a counting partition over ALL tiles (one stream of instances, 0.074 ms) -- stability is not needed: the
// per-tile depth sort that follows orders unique (depth bits, id) keys whatever order it is given.
//
// Round 3: K3 emits one instance stream per tile BAND (common.h: band x = the eighth of the frame that XCD x composites)
// and every kernel here runs band x's workgroups on XCD x (workgroup b -> XCD b % 8, so band = blockIdx.x % 8):
//   count   : a workgroup histograms its chunk of its band's stream over the band's tiles in LDS and writes the counts
//                                                                                       table[band][chunk][tile]
//   colscan : per tile, exclusive scan of the chunk counts inside each of kGroups chunk groups (in place)
//   base    : exclusive scan of the tile totals = tile ranges (one workgroup); group sums -> prefix over the groups
//   scatter : pos = base[tile] + gprefix[group][tile] + table[chunk][tile] + (LDS fetch-and-add); ids only
// What the band split buys: the scatter's four-byte stores to a tile's list all come from ONE XCD, so the list's cache
// lines are assembled in one L2 instead of being written back in pieces by eight (the per-XCD L2s are not coherent:
// 17 vs 35 us for 2.7 M random four-byte stores, profiles/r02_microbench_atomics.txt); the LDS histogram and the count
// table shrink to an eighth (the table was 21 MB written, scanned and read again per frame at 1080p).
// Used while a band's tiles fit a 32-bit LDS histogram (T <= kTileBinMaxTiles); larger grids take the radix path.
#include "common.h"

namespace hgs {
namespace {

constexpr int kTbThreads = 256;
constexpr int kGroups = 16;
constexpr int kTbStage = 8192;   // tiles whose totals the base kernel stages in LDS (1080p: 8160)
constexpr int kMaxTiles = 32768;

struct BandStream {
  uint32_t begin, end;           // the band's slice of the instance arrays, clipped to the capacity
};
// band totals: column b of g.block_band (stride col = nblk + 1), row nblk
__device__ __forceinline__ BandStream band_stream(const uint32_t* __restrict__ band_totals, int col, int band, uint32_t cap) {
  uint32_t begin = 0;
  for (int b = 0; b < band; ++b) begin += band_totals[(size_t)b * col + (col - 1)];
  const uint32_t n = band_totals[(size_t)band * col + (col - 1)];
  BandStream s;
  s.begin = min(begin, cap);
  s.end = min(begin + n, cap);
  return s;
}

// grid = kBands * max_chunks; workgroup b: band b % kBands (= its XCD), chunk b / kBands of that band's stream
__global__ __launch_bounds__(kTbThreads) void tb_count_kernel(const uint32_t* __restrict__ keys, uint32_t cap,
                                                              const uint32_t* __restrict__ band_totals, int col, int per,
                                                              uint32_t chunk, int max_chunks,
                                                              uint32_t* __restrict__ table, uint32_t* __restrict__ totals,
                                                              int Tp) {
  extern __shared__ uint32_t h[];
  if (blockIdx.x == 0)      // per-tile totals are accumulated with atomics by the column scan that follows
    for (int t = threadIdx.x; t < Tp; t += kTbThreads) totals[t] = 0u;
  const int band = blockIdx.x % kBands, c = blockIdx.x / kBands;
  const BandStream st = band_stream(band_totals, col, band, cap);
  const uint32_t base = st.begin + (uint32_t)c * chunk;
  if (base >= st.end) return;                              // beyond the band's last chunk: no table row is read either
  for (int t = threadIdx.x; t < per; t += kTbThreads) h[t] = 0u;
  __syncthreads();
  const uint32_t end = min(base + chunk, st.end);
  for (uint32_t i = base + threadIdx.x; i < end; i += kTbThreads) atomicAdd(&h[keys[i]], 1u);
  __syncthreads();
  uint32_t* row = table + ((size_t)band * max_chunks + c) * per;
  for (int t = threadIdx.x; t < per; t += kTbThreads) row[t] = h[t];
}

// grid (ceil(per / 256), kGroups, kBands): exclusive scan over the chunks of one group, per tile of the band; group sums to
// gsum[g][tile] and (atomically) to totals[tile]  (tile = global tile id = band * per + local id; arrays padded to Tp = 8 per)
__global__ __launch_bounds__(kTbThreads) void tb_colscan_kernel(uint32_t* __restrict__ table,
                                                                const uint32_t* __restrict__ band_totals, int col, int per,
                                                                uint32_t chunk, int max_chunks, uint32_t cap, int Tp,
                                                                uint32_t* __restrict__ gsum, uint32_t* __restrict__ totals) {
  const int t = blockIdx.x * kTbThreads + threadIdx.x;
  if (t >= per) return;
  const int g = blockIdx.y, band = blockIdx.z;
  const BandStream st = band_stream(band_totals, col, band, cap);
  const int nchunks = (int)((st.end - st.begin + chunk - 1) / chunk);
  const int cpg = (nchunks + kGroups - 1) / kGroups;
  const int c0 = g * cpg, c1 = min(c0 + cpg, nchunks);
  uint32_t* tb = table + (size_t)band * max_chunks * per + t;
  uint32_t acc = 0;
  for (int cb = c0; cb < c1; cb += 8) {           // 8 independent loads in flight, then the 8 prefix stores
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (cb + k < c1) ? tb[(size_t)(cb + k) * per] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (cb + k < c1) tb[(size_t)(cb + k) * per] = acc;
      acc += v[k];
    }
  }
  gsum[(size_t)g * Tp + band * per + t] = acc;
  if (acc) atomicAdd(&totals[band * per + t], acc);
}

// Workgroup 0: exclusive scan of the tile totals = tile ranges (a lane owns `each` consecutive tiles: one workgroup-wide
// scan); resets the depth-sort class counters.  Workgroups 1..: per tile, exclusive prefix of the group sums over the
// groups (independent of workgroup 0).
__global__ __launch_bounds__(1024) void tb_base_kernel(const uint32_t* __restrict__ gsum, uint32_t* __restrict__ gpre,
                                                       const uint32_t* __restrict__ totals,
                                                       int T, int Tp, int each, uint32_t* __restrict__ base,
                                                       uint32_t* __restrict__ ranges, uint32_t* __restrict__ big) {
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {
    const int t = (blockIdx.x - 1) * 1024 + tid;
    if (t >= T) return;
    uint32_t acc = 0;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const uint32_t v = gsum[(size_t)g * Tp + t];
      gpre[(size_t)g * Tp + t] = acc;
      acc += v;
    }
    return;
  }
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t tot_s[kTbStage];     // the tile totals (coalesced loads of the group sums)
  const int lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { big[0] = 0; big[1] = 0; big[2] = 0; }
  auto total_of = [&](int t) { return totals[t]; };
  const int t0 = tid * each;
  const bool staged = T <= kTbStage;
  if (staged) {
    for (int t = tid; t < T; t += 1024) tot_s[t] = total_of(t);
    __syncthreads();
  }
  uint32_t mine = 0;
  for (int i = 0; i < each; ++i)
    if (t0 + i < T) mine += staged ? tot_s[t0 + i] : total_of(t0 + i);
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
  for (int i = 0; i < each; ++i) {
    const int t = t0 + i;
    if (t >= T) break;
    const uint32_t tot = staged ? tot_s[t] : total_of(t);
    base[t] = start;
    // empty tiles read (0, 0), as after identifyTileRanges
    reinterpret_cast<uint2*>(ranges)[t] = tot ? make_uint2(start, start + tot) : make_uint2(0u, 0u);
    start += tot;
  }
}

// Launch order of the one-wave-per-tile kernels (K6, K7) for one band: the band's tiles by DESCENDING instance count
// (binning.hip, tile_order_kernel, explains why).  Runs as kBands extra workgroups at the end of the scatter launch --
// it only needs the tile ranges, which the previous launch wrote -- instead of as a launch of its own (~5 us).
// Counting sort over 1024 quantised counts (count / 4, everything above 4092 in the first bucket); 256 lanes.
__device__ __forceinline__ void band_tile_order(const uint32_t* __restrict__ ranges, int T, int per, int band,
                                                uint32_t* __restrict__ order) {
  __shared__ uint32_t hist[1024];
  __shared__ uint32_t wave_tot[kTbThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = band * per, t1 = min(T, t0 + per);
  for (int i = tid; i < 1024; i += kTbThreads) hist[i] = 0;
  __syncthreads();
  for (int t = t0 + tid; t < t1; t += kTbThreads) {
    const uint32_t c = ranges[2 * t + 1] - ranges[2 * t];
    atomicAdd(&hist[1023u - min(c >> 2, 1023u)], 1u);
  }
  __syncthreads();
  // exclusive scan of the 1024 buckets: a lane owns four consecutive ones
  uint32_t v[4], sum = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[k] = hist[tid * 4 + k]; sum += v[k]; }
  uint32_t inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(inc, off, 64);
    if (lane >= off) inc += u;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t run = inc - sum;
  for (int w = 0; w < wave; ++w) run += wave_tot[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) { hist[tid * 4 + k] = run; run += v[k]; }     // first slot of the bucket inside the band
  __syncthreads();
  for (int t = t0 + tid; t < t1; t += kTbThreads) {
    const uint32_t c = ranges[2 * t + 1] - ranges[2 * t];
    const uint32_t k = atomicAdd(&hist[1023u - min(c >> 2, 1023u)], 1u);
    order[k * 8u + (uint32_t)band] = (uint32_t)t;          // workgroup b = k * 8 + band
  }
  for (int k = (t1 > t0 ? t1 - t0 : 0) + tid; k < per; k += kTbThreads) order[(uint32_t)k * 8u + (uint32_t)band] = 0xffffffffu;
}

// LDS holds one absolute output cursor per tile of the band (tile base + group prefix + table row, loaded coalesced): one
// LDS fetch-and-add per instance and no dependent global reads.
__global__ __launch_bounds__(kTbThreads) void tb_scatter_kernel(const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ vals, uint32_t cap,
                                                                const uint32_t* __restrict__ band_totals, int col, int per,
                                                                uint32_t chunk, int max_chunks, int T, int Tp,
                                                                const uint32_t* __restrict__ table,
                                                                const uint32_t* __restrict__ gpre,
                                                                const uint32_t* __restrict__ tbase,
                                                                uint32_t* __restrict__ vals_out,
                                                                const uint32_t* __restrict__ ranges,
                                                                uint32_t* __restrict__ order) {
  extern __shared__ uint32_t h[];
  if ((int)blockIdx.x >= kBands * max_chunks) {             // the kBands workgroups behind the scatter's grid
    band_tile_order(ranges, T, per, (int)blockIdx.x - kBands * max_chunks, order);
    return;
  }
  const int band = blockIdx.x % kBands, c = blockIdx.x / kBands;
  const BandStream st = band_stream(band_totals, col, band, cap);
  const uint32_t base = st.begin + (uint32_t)c * chunk;
  if (base >= st.end) return;
  const int nchunks = (int)((st.end - st.begin + chunk - 1) / chunk);
  const int cpg = (nchunks + kGroups - 1) / kGroups;
  const uint32_t* row = table + ((size_t)band * max_chunks + c) * per;
  const uint32_t* grow = gpre + (size_t)(c / cpg) * Tp + band * per;
  const int tiles = min(per, T - band * per);
  for (int t = threadIdx.x; t < tiles; t += kTbThreads) h[t] = tbase[band * per + t] + grow[t] + row[t];
  __syncthreads();
  const uint32_t end = min(base + chunk, st.end);
  for (uint32_t i = base + threadIdx.x; i < end; i += kTbThreads) {
    const uint32_t t = keys[i];
    const uint32_t gid = vals[i];
    vals_out[atomicAdd(&h[t], 1u)] = gid;
  }
}

inline uint32_t tb_chunk(int T) { return T <= 12288 ? 4096u : 16384u; }

}  // namespace

bool tile_bin_supported(int32_t T) { return T <= kMaxTiles; }

size_t tile_bin_tmp_bytes(uint32_t L, int32_t T) {
  if (!tile_bin_supported(T)) return 0;
  const uint32_t chunk = tb_chunk(T);
  const size_t max_chunks = ((size_t)(L ? L : 1) + chunk - 1) / chunk;
  const size_t per = band_tiles(T), Tp = per * kBands;
  return align_up(kBands * max_chunks * per * 4) + 2 * align_up((size_t)kGroups * Tp * 4) + 2 * align_up(Tp * 4) + kAlign;
}

// keys / vals: the banded instance streams (band-local tile id, Gaussian id); vals_out: ids grouped by tile (unordered
// inside a tile); ranges [T,2], the depth-sort class counters (big[0..2]) and the compositing kernels' launch order
// (tile_order) are written as well.
int launch_tile_bin(const uint32_t* keys, const uint32_t* vals, uint32_t* vals_out, void* tmp, uint32_t L_cap,
                    const uint32_t* band_totals, int32_t nblk, int32_t T, uint32_t* ranges, uint32_t* big,
                    uint32_t* tile_order, hipStream_t s, bool debug) {
  const uint32_t chunk = tb_chunk(T);
  const int max_chunks = (int)(((size_t)L_cap + chunk - 1) / chunk);
  const int per = band_tiles(T), Tp = per * kBands, col = nblk + 1;
  char* c = static_cast<char*>(tmp);
  uint32_t* table = carve<uint32_t>(c, (size_t)kBands * max_chunks * per);
  uint32_t* gsum = carve<uint32_t>(c, (size_t)kGroups * Tp);
  uint32_t* gpre = carve<uint32_t>(c, (size_t)kGroups * Tp);
  uint32_t* tbase = carve<uint32_t>(c, (size_t)Tp);
  uint32_t* totals = carve<uint32_t>(c, (size_t)Tp);
  const size_t lds = (size_t)per * 4;          // <= 16 KiB (T <= 32768)
  hipLaunchKernelGGL(tb_count_kernel, dim3(kBands * max_chunks), dim3(kTbThreads), lds, s, keys, L_cap, band_totals, col, per,
                     chunk, max_chunks, table, totals, Tp);
  HGS_LAUNCH_CHECK("tile_bin_count", s, debug);
  hipLaunchKernelGGL(tb_colscan_kernel, dim3((per + kTbThreads - 1) / kTbThreads, kGroups, kBands), dim3(kTbThreads), 0, s,
                     table, band_totals, col, per, chunk, max_chunks, L_cap, Tp, gsum, totals);
  HGS_LAUNCH_CHECK("tile_bin_colscan", s, debug);
  hipLaunchKernelGGL(tb_base_kernel, dim3(1 + (T + 1023) / 1024), dim3(1024), 0, s, gsum, gpre, totals, T, Tp, (T + 1023) / 1024, tbase,
                     ranges, big);
  HGS_LAUNCH_CHECK("tile_bin_base", s, debug);
  hipLaunchKernelGGL(tb_scatter_kernel, dim3(kBands * max_chunks + kBands), dim3(kTbThreads), lds, s, keys, vals, L_cap,
                     band_totals, col, per, chunk, max_chunks, T, Tp, table, gpre, tbase, vals_out, ranges, tile_order);
  HGS_LAUNCH_CHECK("tile_bin_scatter", s, debug);
  return HGS_OK;
}

}  // namespace hgs
