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
//             and -- same launch since round 5 (tb_count_kernel below) -- the workgroup that completes one of the band's
//             kGroups chunk groups scans that group's counts per tile (exclusive, in place), stores the group sums and
//             adds them to the tile totals
//   scatter : every workgroup scans ITS BAND's tile totals itself (<= 4096 values in LDS: the tile bases are band-local --
//             band begin + the totals of the band's earlier tiles) and adds the earlier groups' sums: pos = base[tile] +
//             sum of gsum[g' < group][tile] + table[chunk][tile] + (LDS fetch-and-add); ids only.  The band's first
//             workgroup also writes the tile ranges.  (Round 3 had a launch of its own for bases and group prefixes:
//             10 us for a few kilobytes of work, most of it launch latency.)
// What the band split buys: the scatter's four-byte stores to a tile's list all come from ONE XCD, so the list's cache
// lines are assembled in one L2 instead of being written back in pieces by eight (the per-XCD L2s are not coherent:
// 17 vs 35 us for 2.7 M random four-byte stores, profiles/r02_microbench_atomics.txt); the LDS histogram and the count
// table shrink to an eighth (the table was 21 MB written, scanned and read again per frame at 1080p).
// Used while a band's tiles fit a 32-bit LDS histogram (T <= kTileBinMaxTiles); larger grids take the radix path.
#include "common.h"

namespace hgs {
namespace {

constexpr int kTbThreads = 256;
constexpr int kGroups = 8;
constexpr int kTbBatch = 16;               // instances per lane whose loads are issued together (count and scatter)
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

// The binning scratch (BinWs::sort_tmp): the count table, the group sums, then -- adjacent, cleared together by K3
// (binning.hip: zero_words) -- the tile totals and the arrival counters of the chunk groups.
struct TbScratch {
  uint32_t *table, *gsum, *totals, *arrive;
};
__host__ inline TbScratch tb_carve(void* tmp, uint32_t L_cap, int32_t T, uint32_t chunk) {
  const size_t max_chunks = ((size_t)(L_cap ? L_cap : 1) + chunk - 1) / chunk;
  const size_t per = band_tiles(T), Tp = per * kBands;
  char* c = static_cast<char*>(tmp);
  TbScratch t;
  t.table = carve<uint32_t>(c, (size_t)kBands * max_chunks * per);
  t.gsum = carve<uint32_t>(c, (size_t)kGroups * Tp);
  t.totals = carve<uint32_t>(c, Tp + (size_t)kBands * kGroups);
  t.arrive = t.totals + Tp;
  return t;
}

__device__ __forceinline__ void store_sc1(uint32_t* p, uint32_t v) {      // write-through (global_store_dword ... sc1)
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid = kBands * max_chunks; workgroup b: band b % kBands (= its XCD), chunk b / kBands of that band's stream.
// COUNT AND COLUMN SCAN IN ONE LAUNCH (rounds 1-4: two, the second 5.5 us + a kernel boundary for 2.6 MB): the chunks of
// a band form kGroups groups; the workgroup that completes a group -- the last of the group's chunks to count itself in
// on the group's arrival counter -- scans the group's rows per tile (exclusive, in place), stores the group sums and
// adds them to the tile totals.  Nobody waits for anybody.  Hand-over = recipe R1 of cdna_hip_programming.md Guideline
// 16: the rows are stored WRITE-THROUGH (sc1), every storing wave drains its stores, then one lane arrives (agent-scope
// atomic); every wave of the finishing workgroup runs an agent-scope acquire (its vector L1; its L2 never held these lines: sc1
// stores do not leave them there) before it reads the rows with plain loads.  Placement only decides speed (a band's
// workgroups share an XCD), never the result.  `totals` and `arrive` are zero on entry (cleared by K3).
// `super` (may be null): K1's superblock totals, consumed by K3 -- zeroed here for the next frame on this stream.
__global__ __launch_bounds__(kTbThreads) void tb_count_kernel(const uint32_t* __restrict__ keys, uint32_t cap,
                                                              const uint32_t* __restrict__ band_totals, int col, int per,
                                                              uint32_t chunk, int max_chunks,
                                                              uint32_t* __restrict__ table, uint32_t* __restrict__ gsum,
                                                              uint32_t* __restrict__ totals, uint32_t* __restrict__ arrive,
                                                              int Tp, uint32_t* __restrict__ super, int n_super) {
  extern __shared__ uint32_t h[];
  __shared__ uint32_t completes_group;
  for (int r = blockIdx.x * kTbThreads + threadIdx.x; r < n_super; r += gridDim.x * kTbThreads) super[r] = 0u;
  const int band = blockIdx.x % kBands, c = blockIdx.x / kBands;
  const BandStream st = band_stream(band_totals, col, band, cap);
  const uint32_t base = st.begin + (uint32_t)c * chunk;
  if (base >= st.end) return;                              // beyond the band's last chunk: not part of any group
  for (int t = threadIdx.x; t < per; t += kTbThreads) h[t] = 0u;
  __syncthreads();
  const uint32_t end = min(base + chunk, st.end);
  // kTbBatch loads in flight per lane, then their LDS adds (a load -> wait -> add loop pays the load latency once per
  // 256 instances: a workgroup's 4096 took sixteen round trips, most of this kernel's time)
  uint32_t i0 = base;
  for (; i0 + kTbThreads * kTbBatch <= end; i0 += kTbThreads * kTbBatch) {       // whole batches (workgroup-uniform)
    const uint32_t* kp = keys + i0 + threadIdx.x;
    uint32_t k[kTbBatch];
#pragma unroll
    for (int j = 0; j < kTbBatch; ++j) k[j] = kp[j * kTbThreads];
#pragma unroll
    for (int j = 0; j < kTbBatch; ++j) atomicAdd(&h[k[j]], 1u);
  }
  for (uint32_t i = i0 + threadIdx.x; i < end; i += kTbThreads) atomicAdd(&h[keys[i]], 1u);   // the band's last chunk
  __syncthreads();
  uint32_t* row = table + ((size_t)band * max_chunks + c) * per;
  for (int t = threadIdx.x; t < per; t += kTbThreads) store_sc1(&row[t], h[t]);
  // ---- arrive on the chunk group; the workgroup that completes it scans it ---------------------------------------------
  const int nchunks = (int)((st.end - st.begin + chunk - 1) / chunk);
  const int cpg = (nchunks + kGroups - 1) / kGroups;
  const int g = c / cpg, c0 = g * cpg, c1 = min(c0 + cpg, nchunks);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its row stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    // (relaxed: the rows were stored write-through and drained above, which is what the hardware needs.  The arrival as
    // an ACQ_REL in the language's memory model is an L2 write-back in front of the atomic on gfx950: +7 us on this
    // 14 us kernel, measured in round 6 -- profiles/r06_notes.md)
    const uint32_t before = __hip_atomic_fetch_add(&arrive[band * kGroups + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    completes_group = before == (uint32_t)(c1 - c0 - 1);
  }
  __syncthreads();
  if (!completes_group) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // EVERY wave of the scanning workgroup, before its first load
  uint32_t* tb0 = table + (size_t)band * max_chunks * per;
  // two tile columns x kRowBatch rows in flight per lane (a column at a time, eight rows at a time, was eight dependent
  // round trips for a group of ten chunks at 1080p: the tail of this launch)
  constexpr int kRowBatch = 16, kCols = 2;
  for (int tb = threadIdx.x; tb - (int)threadIdx.x < per; tb += kCols * kTbThreads) {
    uint32_t acc[kCols];
#pragma unroll
    for (int u = 0; u < kCols; ++u) acc[u] = 0u;
    for (int cb = c0; cb < c1; cb += kRowBatch) {
      uint32_t v[kCols][kRowBatch];
#pragma unroll
      for (int u = 0; u < kCols; ++u)
#pragma unroll
        for (int k = 0; k < kRowBatch; ++k)       // clamped addresses: loads without branches, masked on use
          v[u][k] = tb0[(size_t)min(cb + k, c1 - 1) * per + min(tb + u * kTbThreads, per - 1)];
#pragma unroll
      for (int u = 0; u < kCols; ++u) {
        const int t = tb + u * kTbThreads;
        if (t < per) {
#pragma unroll
          for (int k = 0; k < kRowBatch; ++k) {
            if (cb + k < c1) {
              tb0[(size_t)(cb + k) * per + t] = acc[u];
              acc[u] += v[u][k];
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kCols; ++u) {
      const int t = tb + u * kTbThreads;
      if (t < per) {
        gsum[(size_t)g * Tp + band * per + t] = acc[u];
        if (acc[u]) atomicAdd(&totals[band * per + t], acc[u]);
      }
    }
  }
}

// Launch order of the one-wave-per-tile kernels (K6, K7) for one band: the band's tiles by DESCENDING instance count
// (binning.hip, tile_order_kernel, explains why).  Runs as kBands extra workgroups at the end of the scatter launch --
// it only needs the tile totals, which the previous launch wrote -- instead of as a launch of its own (~5 us).
// Counting sort over 1024 quantised counts (count / 4, everything above 4092 in the first bucket); 256 lanes.
__device__ __forceinline__ void band_tile_order(const uint32_t* __restrict__ totals, int T, int per, int band,
                                                uint32_t* __restrict__ order) {
  __shared__ uint32_t hist[1024];
  __shared__ uint32_t wave_tot[kTbThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = band * per, t1 = min(T, t0 + per);
  for (int i = tid; i < 1024; i += kTbThreads) hist[i] = 0;
  __syncthreads();
  for (int t = t0 + tid; t < t1; t += kTbThreads) {
    const uint32_t c = totals[t];
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
    const uint32_t c = totals[t];
    const uint32_t k = atomicAdd(&hist[1023u - min(c >> 2, 1023u)], 1u);
    order[k * 8u + (uint32_t)band] = (uint32_t)t;          // workgroup b = k * 8 + band
  }
  for (int k = (t1 > t0 ? t1 - t0 : 0) + tid; k < per; k += kTbThreads) order[(uint32_t)k * 8u + (uint32_t)band] = 0xffffffffu;
}

// LDS holds one absolute output cursor per tile of the band: one LDS fetch-and-add per instance and no dependent global
// reads.  The cursor = (band begin + exclusive scan of the band's tile totals, done here by every workgroup: a lane owns
// `each` consecutive tiles) + the sums of the chunk groups before this chunk's + the chunk's row of the scanned table.
// STAGED (chunk = kTbThreads * kTbBatch: one batch per workgroup; T <= 12288): the chunk's instances are first grouped
// by tile in LDS -- a second LDS array counts the chunk's instances per tile (the fetch-and-add returns the rank inside
// (chunk, tile)), its exclusive scan gives every tile's place inside the chunk, (output position, id) pairs are parked
// there -- and then written out in chunk order: neighbouring lanes store to neighbouring words of a tile's list (runs of
// ~4 at 1080p) instead of 64 unrelated words per store.  2.7 M XCD-local four-byte stores: 17.5 us one by one, 8.9 us in
// runs of four (scripts/microbench/atomics.hip).  The order inside a tile's list differs from the unstaged kernel's; the
// depth sort that follows makes both the same list.
template <bool STAGED>
__global__ __launch_bounds__(kTbThreads) void tb_scatter_kernel(const uint32_t* __restrict__ keys,
                                                                const uint32_t* __restrict__ vals, uint32_t cap,
                                                                const uint32_t* __restrict__ band_totals, int col, int per,
                                                                uint32_t chunk, int max_chunks, int T, int Tp,
                                                                const uint32_t* __restrict__ table,
                                                                const uint32_t* __restrict__ gsum,
                                                                const uint32_t* __restrict__ totals,
                                                                uint32_t* __restrict__ vals_out,
                                                                uint32_t* __restrict__ ranges,
                                                                uint32_t* __restrict__ big,
                                                                uint32_t* __restrict__ order) {
  extern __shared__ uint32_t h[];
  __shared__ uint32_t wave_tot[kTbThreads / 64];
  if ((int)blockIdx.x >= kBands * max_chunks) {             // the kBands workgroups behind the scatter's grid
    band_tile_order(totals, T, per, (int)blockIdx.x - kBands * max_chunks, order);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int band = blockIdx.x % kBands, c = blockIdx.x / kBands;
  if (blockIdx.x == 0 && tid == 0) { big[0] = 0; big[1] = 0; big[2] = 0; }   // the depth sort's class counters
  const BandStream st = band_stream(band_totals, col, band, cap);
  const uint32_t base = st.begin + (uint32_t)c * chunk;
  const bool first = c == 0;                                // writes the band's tile ranges (also of an empty band)
  if (base >= st.end && !first) return;
  const int tiles = max(0, min(per, T - band * per));
  // STAGED: the chunk's instances, requested before anything else (consumed after the cursors are built)
  uint32_t s_tile[STAGED ? kTbBatch : 1], s_gid[STAGED ? kTbBatch : 1];
  uint32_t* cnt = h + per;
  if constexpr (STAGED) {
    if (base < st.end) {
#pragma unroll
      for (int j = 0; j < kTbBatch; ++j) {
        const uint32_t i = min(base + (uint32_t)(j * kTbThreads + tid), st.end - 1u);
        s_tile[j] = keys[i];
        s_gid[j] = vals[i];
      }
    }
    for (int t = tid; t < per; t += kTbThreads) cnt[t] = 0u;
  }
  // ---- tile bases of the band: exclusive scan of its totals ---------------------------------------------------
  const int each = (per + kTbThreads - 1) / kTbThreads;
  const uint32_t* tot = totals + band * per;
  for (int tb = tid; tb < tiles + tid; tb += kTbThreads * 4) {                 // coalesced, four loads in flight
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = tot[min(tb + j * kTbThreads, tiles - 1)];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (tb + j * kTbThreads < tiles) h[tb + j * kTbThreads] = v[j];
  }
  __syncthreads();
  const int t0 = tid * each;
  uint32_t mine = 0;
  for (int i = 0; i < each; ++i)
    if (t0 + i < tiles) mine += h[t0 + i];
  uint32_t inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(inc, off, 64);
    if (lane >= off) inc += v;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t start = st.begin + inc - mine;
  for (int w = 0; w < wave; ++w) start += wave_tot[w];
  for (int i = 0; i < each; ++i) {
    const int t = t0 + i;
    if (t >= tiles) break;
    const uint32_t n = h[t];
    h[t] = start;
    // empty tiles read (0, 0), as after identifyTileRanges
    if (first) reinterpret_cast<uint2*>(ranges)[band * per + t] = n ? make_uint2(start, start + n) : make_uint2(0u, 0u);
    start += n;
  }
  if (base >= st.end) return;
  __syncthreads();
  // ---- + the earlier chunk groups + this chunk's row -------------------------------------------------------------
  const int nchunks = (int)((st.end - st.begin + chunk - 1) / chunk);
  const int cpg = (nchunks + kGroups - 1) / kGroups;
  const int g = c / cpg;
  const uint32_t* row = table + ((size_t)band * max_chunks + c) * per;
  const uint32_t* gs = gsum + band * per;
  for (int tb = tid; tb - tid < tiles; tb += 2 * kTbThreads) {       // two tile columns' row + group sums in flight
    uint32_t add[2][kGroups];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = min(tb + u * kTbThreads, tiles - 1);
      add[u][kGroups - 1] = row[t];
#pragma unroll
      for (int k = 0; k < kGroups - 1; ++k) add[u][k] = gs[(size_t)min(k, max(g - 1, 0)) * Tp + t];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = tb + u * kTbThreads;
      if (t < tiles) {
        uint32_t acc = h[t] + add[u][kGroups - 1];
#pragma unroll
        for (int k = 0; k < kGroups - 1; ++k) acc += (k < g) ? add[u][k] : 0u;
        h[t] = acc;
      }
    }
  }
  __syncthreads();
  const uint32_t end = min(base + chunk, st.end);
  if constexpr (STAGED) {
    const uint32_t n = end - base;                          // 1 .. kTbThreads * kTbBatch
    uint2* stage = reinterpret_cast<uint2*>(cnt + per);
    uint32_t rank[kTbBatch];
#pragma unroll
    for (int j = 0; j < kTbBatch; ++j) {
      rank[j] = 0u;
      if ((uint32_t)(j * kTbThreads + tid) < n) rank[j] = atomicAdd(&cnt[s_tile[j]], 1u);
    }
    __syncthreads();
    // exclusive scan of the chunk's counts over the band's tiles (in place): a lane owns `each` consecutive tiles
    uint32_t own = 0;
    for (int i = 0; i < each; ++i)
      if (t0 + i < tiles) own += cnt[t0 + i];
    uint32_t inc2 = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(inc2, off, 64);
      if (lane >= off) inc2 += v;
    }
    if (lane == 63) wave_tot[wave] = inc2;                  // (its first use was read two barriers ago)
    __syncthreads();
    uint32_t at = inc2 - own;
    for (int w = 0; w < wave; ++w) at += wave_tot[w];
    for (int i = 0; i < each; ++i) {
      if (t0 + i >= tiles) break;
      const uint32_t c_t = cnt[t0 + i];
      cnt[t0 + i] = at;
      at += c_t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kTbBatch; ++j)
      if ((uint32_t)(j * kTbThreads + tid) < n)
        stage[cnt[s_tile[j]] + rank[j]] = make_uint2(h[s_tile[j]] + rank[j], s_gid[j]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kTbBatch; ++j) {
      const uint32_t i = (uint32_t)(j * kTbThreads + tid);
      if (i < n) {
        const uint2 e = stage[i];
        vals_out[e.x] = e.y;
      }
    }
    return;
  }
  // batched as in the count kernel: the loads of kTbBatch instances per lane, their LDS fetch-and-adds, their stores
  uint32_t i0 = base;
  for (; i0 + kTbThreads * kTbBatch <= end; i0 += kTbThreads * kTbBatch) {
    const uint32_t* kp = keys + i0 + tid;
    const uint32_t* vp = vals + i0 + tid;
    uint32_t t[kTbBatch], gid[kTbBatch], pos[kTbBatch];
#pragma unroll
    for (int j = 0; j < kTbBatch; ++j) {
      t[j] = kp[j * kTbThreads];
      gid[j] = vp[j * kTbThreads];
    }
#pragma unroll
    for (int j = 0; j < kTbBatch; ++j) pos[j] = atomicAdd(&h[t[j]], 1u);
#pragma unroll
    for (int j = 0; j < kTbBatch; ++j) vals_out[pos[j]] = gid[j];
  }
  for (uint32_t i = i0 + tid; i < end; i += kTbThreads) {                  // the band's last chunk
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
  return align_up(kBands * max_chunks * per * 4) + align_up((size_t)kGroups * Tp * 4) +
         align_up((Tp + (size_t)kBands * kGroups) * 4) + kAlign;
}

uint32_t* tile_bin_zero_words(void* tmp, uint32_t L_cap, int32_t T) {
  return tile_bin_supported(T) ? tb_carve(tmp, L_cap, T, tb_chunk(T)).totals : nullptr;
}
int tile_bin_zero_count(int32_t T) { return tile_bin_supported(T) ? band_tiles(T) * kBands + kBands * kGroups : 0; }

// keys / vals: the banded instance streams (band-local tile id, Gaussian id); vals_out: ids grouped by tile (unordered
// inside a tile); ranges [T,2], the depth-sort class counters (big[0..2]) and the compositing kernels' launch order
// (tile_order) are written as well.  The scratch's tile totals and arrival counters must be zero (the banded K3 clears
// them).  super (optional): K1's superblock totals, cleared here for the next frame.
int launch_tile_bin(const uint32_t* keys, const uint32_t* vals, uint32_t* vals_out, void* tmp, uint32_t L_cap,
                    const uint32_t* band_totals, int32_t nblk, int32_t T, uint32_t* ranges, uint32_t* big,
                    uint32_t* tile_order, uint32_t* super, hipStream_t s, bool debug) {
  const uint32_t chunk = tb_chunk(T);
  const int max_chunks = (int)(((size_t)L_cap + chunk - 1) / chunk);
  const int per = band_tiles(T), Tp = per * kBands, col = nblk + 1;
  const TbScratch t = tb_carve(tmp, L_cap, T, chunk);
  const size_t lds = (size_t)per * 4;          // <= 16 KiB (T <= 32768)
  hipLaunchKernelGGL(tb_count_kernel, dim3(kBands * max_chunks), dim3(kTbThreads), lds, s, keys, L_cap, band_totals, col, per,
                     chunk, max_chunks, t.table, t.gsum, t.totals, t.arrive, Tp, super,
                     super ? (int)(super_block_bytes() / sizeof(uint32_t)) : 0);
  HGS_LAUNCH_CHECK("tile_bin_count", s, debug);
  // one batch per workgroup: the scatter groups its chunk by tile in LDS first (cursors + counts + 8 bytes per instance)
  const bool staged = chunk == (uint32_t)(kTbThreads * kTbBatch);
  if (staged)
    hipLaunchKernelGGL(tb_scatter_kernel<true>, dim3(kBands * max_chunks + kBands), dim3(kTbThreads),
                       lds * 2 + (size_t)chunk * 8, s, keys, vals, L_cap, band_totals, col, per, chunk, max_chunks, T, Tp,
                       t.table, t.gsum, t.totals, vals_out, ranges, big, tile_order);
  else
    hipLaunchKernelGGL(tb_scatter_kernel<false>, dim3(kBands * max_chunks + kBands), dim3(kTbThreads), lds, s, keys, vals,
                       L_cap, band_totals, col, per, chunk, max_chunks, T, Tp, t.table, t.gsum, t.totals, vals_out, ranges,
                       big, tile_order);
  HGS_LAUNCH_CHECK("tile_bin_scatter", s, debug);
  return HGS_OK;
}

}  // namespace hgs
