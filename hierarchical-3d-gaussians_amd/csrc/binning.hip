// K3 (duplicateWithKeys), the per-tile depth sort, and K5 (identifyTileRanges).
//
// The reference sorts L 64-bit (tile | depth) keys in one global radix sort.  Here the two key
// halves are handled where they are cheap:
//   1. K3 emits (tile id, Gaussian id) per instance in ascending Gaussian index;
//   2. a STABLE global radix sort by the tile id alone (13 bits at 1080p: 2 passes of 32-bit keys)
//      groups the instances per tile, each group still in ascending Gaussian index;
//   3. every tile's group (327 instances on average in the benchmark) is sorted by (depth bits,
//      Gaussian id) inside LDS by one workgroup -- bitonic network on 64-bit composite keys; groups
//      above 16 Ki instances fall back to a single-workgroup stable LSD radix sort through global
//      scratch.
// The result is bit-identical to the reference's stable 64-bit sort (depth ties keep ascending
// Gaussian index) with 2 instead of 6 passes over L and no 64-bit keys in HBM at all.
//
// K3 is a load-balanced expansion: a workgroup owns 256 consecutive Gaussians, scans their tile
// counts in LDS and then assigns OUTPUT slots (not Gaussians) to lanes, so a Gaussian covering
// hundreds of tiles does not serialise one lane and the stores are coalesced.
#include "common.h"

#include <stdlib.h>

namespace hgs {
namespace {

// k / w and k % w for the rank k of a tile inside a rectangle of width w: k < 1023^2 < 2^20, 1 <= w <= 1023.  The quotient
// of (k + 0.5) * rcp(w) is exact for k < 2^21 (the half keeps exact multiples half a step away from the rounding
// edge: the error of v_rcp_f32 and of the product, 2^-22 of a quotient < 2^21 / w, stays below 0.5 / w) -- six
// instructions where the 32-bit integer division expands to about forty.
__device__ __forceinline__ void divmod_rect(uint32_t k, uint32_t w, uint32_t& q, uint32_t& r) {
  q = (uint32_t)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)w));
  r = k - q * w;
}


// Per-workgroup instance counts in DEPTH-SORTED Gaussian order (feeds the emission-offset scan).
__device__ __forceinline__ uint32_t rect_count(uint2 r) {     // rects are zero for culled Gaussians
  return ((r.y & 0xffffu) - (r.x & 0xffffu)) * ((r.y >> 16) - (r.x >> 16));
}

// Emits (tile id, Gaussian id) for every tile of every visible Gaussian: ascending Gaussian index,
// row-major tiles inside a rectangle (SURVEY.md App. A.7 emission order).
__global__ __launch_bounds__(kPreBlock) void duplicate_tiles_kernel(int P, int gx, GeomWs g, uint32_t cap,
                                                                    uint32_t* __restrict__ tile_keys,
                                                                    uint32_t* __restrict__ vals,
                                                                    uint32_t* __restrict__ ranges, int n_ranges) {
  // the tile ranges must be zero before tile_ranges_kernel fills them: cleared here instead of by a memset launch
  for (int r = blockIdx.x * kPreBlock + threadIdx.x; r < n_ranges; r += gridDim.x * kPreBlock) ranges[r] = 0u;
  __shared__ uint32_t excl[kPreBlock + 1];
  __shared__ uint2 lrect[kPreBlock];
  __shared__ uint32_t wave_tot[kPreBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * kPreBlock + tid;
  const uint32_t gid = (uint32_t)i;
  const uint2 myrect = (i < P) ? reinterpret_cast<const uint2*>(g.rects)[gid] : make_uint2(0u, 0u);
  const uint32_t cnt = rect_count(myrect);
  lrect[tid] = myrect;
  uint32_t inc = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
  const uint32_t my_excl = wbase + inc - cnt;
  excl[tid] = my_excl;
  if (tid == kPreBlock - 1) excl[kPreBlock] = wbase + inc;
  const uint32_t block_base = g.block_sums[blockIdx.x];
  if (i < P && cnt) g.offsets[gid] = block_base + my_excl;     // emission offset of this Gaussian's run
  __syncthreads();
  const uint32_t total = excl[kPreBlock];
  for (uint32_t s = tid; s < total; s += kPreBlock) {
    // largest j with excl[j] <= s (zero-count entries are skipped: the search lands on the LAST
    // index whose start is <= s)
    int lo = 0, hi = kPreBlock;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int mid = (lo + hi) >> 1;
      if (excl[mid] <= s) lo = mid; else hi = mid;
    }
    const uint32_t gg = blockIdx.x * kPreBlock + lo;
    const uint32_t k = s - excl[lo];
    const uint2 rc = lrect[lo];
    const uint32_t minx = rc.x & 0xffffu, miny = rc.x >> 16;
    const uint32_t w = (rc.y & 0xffffu) - minx;
    uint32_t kq, kr;
    divmod_rect(k, w, kq, kr);
    const uint32_t ty = miny + kq, tx = minx + kr;
    if (block_base + s < cap) {     // cap < L only when a speculative capacity was too small (caller retries)
      tile_keys[block_base + s] = ty * (uint32_t)gx + tx;
      vals[block_base + s] = gg;
    }
  }
}


// Inclusive prefix sum over the 64 lanes of a wave in six DPP adds (row_shr 1 / 2 / 4 / 8 inside each row of 16, then
// row_bcast 15 / 31 across the rows) -- __shfl_up is a ds_bpermute per step, through the LDS pipe.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
  v = dpp_add_u32<0x111, 0xf>(v);    // row_shr:1
  v = dpp_add_u32<0x112, 0xf>(v);    // row_shr:2
  v = dpp_add_u32<0x114, 0xf>(v);    // row_shr:4
  v = dpp_add_u32<0x118, 0xf>(v);    // row_shr:8
  v = dpp_add_u32<0x142, 0xa>(v);    // row_bcast:15 -> rows 1, 3
  v = dpp_add_u32<0x143, 0xc>(v);    // row_bcast:31 -> rows 2, 3
  return v;
}

// The same expansion with ONE OUTPUT STREAM PER TILE BAND (common.h: kBands): an instance goes to the stream of the band
// its tile lies in, at  band base + this workgroup's exclusive offset in the band (K1 counted, the scan kernel
// accumulated: g.block_band) + the number of the workgroup's EARLIER Gaussians' instances in that band + the
// instance's rank among its own Gaussian's instances in that band.  A rectangle's tiles are emitted in ascending tile
// id, so that rank is  k - (# of the rectangle's tiles below the band's first tile), both in closed form: the slot loop
// needs no ballots and no barriers.  The key is the BAND-LOCAL tile id.  The tile-binning kernels then run band x's
// workgroups on XCD x, the XCD that later composites those tiles: every list is assembled in one L2.  Deterministic:
// no atomics, the order inside a stream is (workgroup, emission order).
// `super` (K1's superblock totals, preprocess.hip): g.block_sums / g.block_band hold the RAW workgroup sums and this
// workgroup builds its nine exclusive prefixes and the nine grand totals itself; the last workgroup stores the totals
// (entry nblk of each array) and the instance count into `total_mirror` (mapped host word, may be null).
// `zero_words`: n_zero words cleared for the counting kernels that follow (their arrival counters and tile totals).
// SHARING (round 5, after the reference's scripts at scale): a hierarchy cut lists its big nodes side by side, and one
// workgroup's 256 Gaussians then emit a quarter of a million instances while the mean is a few thousand -- one compute
// unit emitting for 0.25 ms after everybody else has finished.  Positions are closed-form, so ANY workgroup can emit any
// slot of any block once it has rebuilt that block's prologue in its LDS.  The HOST hands K1 and K3 the same threshold
// (k3_heavy_threshold: a multiple of the mean emission the frame's capacity allows, at least kHeavyFloor); a K1 workgroup
// whose sum exceeds it files (block, sum) into the HEAVY LIST that lives in the last row of the superblock totals (one
// atomic add reserves the place: the list's ORDER differs from run to run, no result depends on it).  A K3 workgroup
// reads the list's length with its prefix loads and, only if it is not zero, takes the rare path: the (at most share_max)
// listed blocks' own workgroups emit their first thr slots, and the excess of all listed blocks is dealt out in equal
// contiguous shares by blockIdx.  Static inside a launch, no flags, nobody waits: the result does not depend on who emits
// a slot.  (Round 5 had every workgroup rebuild the list from all raw sums -- 2 x nblk / 256 dependent loads per lane, 12 ms
// per frame at the 61 000 workgroups of a 50 M-node cut, so grids above 4 096 workgroups went without sharing.)
constexpr uint32_t kHeavyFloor = 4096u, kHeavyFactor = 2u, kShareMin = 1024u;

struct K3Lds {
  uint32_t pre9[1 + kBands], tot9[1 + kBands];
  uint32_t excl[kPreBlock + 1];
  uint2 lrect[kPreBlock];
  uint32_t wave_tot[kPreBlock / 64][kBands + 1];   // [.][kBands]: all bands
  uint32_t bpos[kBands];                 // where the block's part of each band's stream begins
  uint32_t rb[kBands][kPreBlock];        // per band and Gaussian: (earlier Gaussians' instances in the band) -
                                         // (own tiles below the band): position of instance k = bpos + rb + k
  uint32_t hl_blk[kMaxHeavy], hl_pe[kMaxHeavy + 1];   // the rare path's list: block, exclusive prefix of the excesses
  uint32_t scan_tmp[kPreBlock / 64];
};

// exclusive scan over the workgroup (two barriers); `total` = the sum
__device__ __forceinline__ uint32_t k3_wg_scan_excl(uint32_t v, uint32_t* tmp, uint32_t& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_scan_incl(v);
  __syncthreads();
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  uint32_t base = 0, t = 0;
#pragma unroll
  for (int w = 0; w < kPreBlock / 64; ++w) {
    const uint32_t x = tmp[w];
    base += (w < wave) ? x : 0u;
    t += x;
  }
  total = t;
  return base + inc - v;
}

// The prologue of block `blk` (a workgroup of K1's grid of nblk): fills l.excl / lrect / rb / bpos; ends with a barrier.
// OWN: blk is this workgroup's own block -- it also stores its Gaussians' emission offsets, the last block stores the
// totals, and `heavy_n` returns the length of K1's heavy list (uniform)
template <bool OWN>
__device__ __forceinline__ void k3_prologue(K3Lds& l, int blk, int nblk, int P, int gx, int T, int per, const GeomWs& g,
                                            const uint32_t* __restrict__ super, uint32_t* __restrict__ total_mirror,
                                            uint32_t& heavy_n) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blk * kPreBlock + tid;
  const uint32_t gid = (uint32_t)i;
  // every global load of the prologue is issued here, together: the kernel is latency-bound (3 900 small workgroups)
  const uint2 myrect = (i < P) ? reinterpret_cast<const uint2*>(g.rects)[gid] : make_uint2(0u, 0u);
  const int col = nblk + 1;                               // column stride of g.block_band
  uint32_t block_base = 0;
  uint32_t band_tot = 0, band_off = 0;                    // lanes 0..7 of wave 0: total of band `tid`, the block's offset in it
  heavy_n = 0u;
  if (!super) {
    block_base = g.block_sums[blk];
    if (tid < kBands) {
      band_tot = g.block_band[(size_t)tid * col + nblk];
      band_off = g.block_band[(size_t)tid * col + blk];
    }
  } else {
    // wave w: arrays w, w + 4, w + 8 (0 = all instances, 1 + b = band b).  prefix = superblock totals before the block's
    // superblock + raw sums of the workgroups before it inside the superblock; total = all superblocks
    const int sb = blk / kSuper, nsb = (nblk + kSuper - 1) / kSuper;
    // the wave's (up to) three arrays TOGETHER: their loads are issued before any of them is waited for (array by array,
    // the loop was three dependent round trips in front of the first barrier)
    constexpr int kWaves = kPreBlock / 64, kPer = (1 + kBands + kWaves - 1) / kWaves;
    uint32_t pre[kPer], x0[kPer];
    const int j = min(sb * kSuper + lane, blk);                       // clamped: loads without branches, masked on use
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int a = min(wave + u * kWaves, kBands);                   // (waves 1-3's third array: array 8 once more)
      const uint32_t* raw = a == 0 ? g.block_sums : g.block_band + (size_t)(a - 1) * col;
      pre[u] = raw[j];
      x0[u] = super[a * kMaxSuper + min(lane, nsb - 1)];
    }
    if constexpr (OWN) heavy_n = super[kHeavyRow];           // (a scalar load, issued with the others)
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int a = min(wave + u * kWaves, kBands);
      uint32_t p = (sb * kSuper + lane < blk) ? pre[u] : 0u;
      uint32_t tot = (lane < nsb) ? x0[u] : 0u;
      p += (lane < sb) ? tot : 0u;
      for (int s0 = 64; s0 < nsb; s0 += 64) {                         // more than 64 superblocks: P > 1 M
        const uint32_t x = (s0 + lane < nsb) ? super[a * kMaxSuper + s0 + lane] : 0u;
        tot += x;
        p += (s0 + lane < sb) ? x : 0u;
      }
      // (DPP scans, the total in lane 63: six register-file adds each instead of six trips through the LDS crossbar)
      p = wave_scan_incl(p);
      tot = wave_scan_incl(tot);
      if (lane == 63) { l.pre9[a] = p; l.tot9[a] = tot; }
    }
  }
  const uint32_t cnt = rect_count(myrect);
  l.lrect[tid] = myrect;
  // own instances per band (as K1 counted them)
  uint32_t cb[kBands];
#pragma unroll
  for (int b = 0; b < kBands; ++b) cb[b] = 0u;
  if (cnt) {
    const int minx = (int)(myrect.x & 0xffffu), miny = (int)(myrect.x >> 16);
    const int maxx = (int)(myrect.y & 0xffffu), maxy = (int)(myrect.y >> 16);
    const int w = maxx - minx;
    const int t_first = miny * gx + minx, t_last = (maxy - 1) * gx + maxx - 1;
    int b_first = 0, b_last = 0;
#pragma unroll
    for (int b = 1; b < kBands; ++b) {
      b_first += (t_first >= b * per) ? 1 : 0;
      b_last += (t_last >= b * per) ? 1 : 0;
    }
    if (b_first == b_last) {
#pragma unroll
      for (int b = 0; b < kBands; ++b) cb[b] = (b == b_first) ? cnt : 0u;
    } else {
      int prev = 0;
#pragma unroll
      for (int b = 0; b < kBands; ++b) {
        const int x = min((b + 1) * per, T);
        const int xr = x / gx, xc = x - xr * gx;
        int c = (min(max(xr, miny), maxy) - miny) * w;
        if (xr >= miny && xr < maxy) c += min(max(xc - minx, 0), w);
        cb[b] = (uint32_t)(c - prev);
        prev = c;
      }
    }
  }
  // inclusive scans over the block's Gaussians: all instances (emission offsets) and each band's
  const uint32_t inc = wave_scan_incl(cnt);
  uint32_t incb[kBands];
#pragma unroll
  for (int b = 0; b < kBands; ++b) incb[b] = wave_scan_incl(cb[b]);
  if (lane == 63) {
    l.wave_tot[wave][kBands] = inc;
#pragma unroll
    for (int b = 0; b < kBands; ++b) l.wave_tot[wave][b] = incb[b];
  }
  if (!super && wave == 0) {                              // band base = the totals of the bands before it
    const uint32_t incl = wave_scan_incl(tid < kBands ? band_tot : 0u);
    if (tid < kBands) l.bpos[tid] = incl - band_tot + band_off;
  }
  __syncthreads();
  if (super) {                                            // (pre9 / tot9 of all four waves are there now)
    block_base = l.pre9[0];
    if (wave == 0) {
      if (tid < kBands) { band_tot = l.tot9[1 + tid]; band_off = l.pre9[1 + tid]; }
      const uint32_t incl = wave_scan_incl(tid < kBands ? band_tot : 0u);
      if (tid < kBands) l.bpos[tid] = incl - band_tot + band_off;     // (read after the barrier in front of the slot loop)
    }
    if (OWN && blk == nblk - 1 && tid <= kBands) {        // the totals, where the scan launch used to leave them
      uint32_t* arr = tid == 0 ? g.block_sums : g.block_band + (size_t)(tid - 1) * col;
      arr[nblk] = l.tot9[tid];
      if (tid == 0 && total_mirror) *total_mirror = l.tot9[0];
    }
  }
  uint32_t wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += l.wave_tot[w][kBands];
  const uint32_t my_excl = wbase + inc - cnt;
  l.excl[tid] = my_excl;
  if (tid == kPreBlock - 1) l.excl[kPreBlock] = wbase + inc;
  {
    uint32_t below = 0;                                   // own tiles in the bands before b
#pragma unroll
    for (int b = 0; b < kBands; ++b) {
      uint32_t wb = 0;
      for (int w = 0; w < wave; ++w) wb += l.wave_tot[w][b];
      l.rb[b][tid] = wb + incb[b] - cb[b] - below;        // (may wrap: added to k >= below modulo 2^32)
      below += cb[b];
    }
  }
  if (OWN && i < P && cnt) g.offsets[gid] = block_base + my_excl;   // emission offset of this Gaussian's run (K7 / K8 slots)
  __syncthreads();
}

// The rare path's prologue of ANOTHER block, as a function of its own: inlined into the kernel a second time it doubled
// the kernel's register count (82 for 36: 5 waves per SIMD for 8 in a latency-bound kernel); as a call it costs the
// kernel 16 registers and nothing on the usual path.  Plain pointers: a struct by reference would go through scratch.
__device__ __attribute__((noinline)) void k3_prologue_other(K3Lds* l, int blk, int nblk, int P, int gx, int T, int per,
                                                            uint32_t* rects, uint32_t* block_sums, uint32_t* block_band,
                                                            const uint32_t* super) {
  GeomWs g{};
  g.rects = rects;
  g.block_sums = block_sums;
  g.block_band = block_band;
  uint32_t unused;
  k3_prologue<false>(*l, blk, nblk, P, gx, T, per, g, super, nullptr, unused);
}

// slots [s_begin, s_end) of block `blk`, whose prologue is in l
__device__ __forceinline__ void k3_emit(const K3Lds& l, int blk, uint32_t s_begin, uint32_t s_end, int gx, int per,
                                        uint32_t cap, uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ vals) {
  const float rcp_per = __builtin_amdgcn_rcpf((float)per);
  for (uint32_t s = s_begin + threadIdx.x; s < s_end; s += kPreBlock) {
    // largest j with excl[j] <= s (zero-count entries are skipped: the search lands on the LAST index whose start is <= s)
    int lo = 0, hi = kPreBlock;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int mid = (lo + hi) >> 1;
      if (l.excl[mid] <= s) lo = mid; else hi = mid;
    }
    const uint32_t gg = (uint32_t)blk * kPreBlock + lo;
    const uint32_t k = s - l.excl[lo];
    const uint2 rc = l.lrect[lo];
    const uint32_t minx = rc.x & 0xffffu, miny = rc.x >> 16;
    const uint32_t w = (rc.y & 0xffffu) - minx;
    uint32_t kq, kr;
    divmod_rect(k, w, kq, kr);
    const uint32_t tile = (miny + kq) * (uint32_t)gx + minx + kr;
    // band = tile / per by the same reciprocal trick (tile < 2^20; rcp_per is wave-uniform): three instructions for
    // the seven compare-and-add pairs of a boundary count
    const uint32_t band = (uint32_t)(((float)tile + 0.5f) * rcp_per);
    const uint32_t pos = l.bpos[band] + l.rb[band][lo] + k;
    if (pos < cap) {       // cap < L only when a speculative capacity was too small (caller retries)
      tile_keys[pos] = tile - band * (uint32_t)per;
      vals[pos] = gg;
    }
  }
}

__global__ __launch_bounds__(kPreBlock) void duplicate_tiles_banded_kernel(int P, int gx, int T, int per, GeomWs g,
                                                                           uint32_t cap,
                                                                           uint32_t* __restrict__ tile_keys,
                                                                           uint32_t* __restrict__ vals,
                                                                           uint32_t* __restrict__ ranges, int n_ranges,
                                                                           const uint32_t* __restrict__ super,
                                                                           uint32_t* __restrict__ total_mirror,
                                                                           uint32_t* __restrict__ zero_words, int n_zero,
                                                                           int share_max, uint32_t thr) {
  for (int r = blockIdx.x * kPreBlock + threadIdx.x; r < n_ranges; r += gridDim.x * kPreBlock) ranges[r] = 0u;
  for (int r = blockIdx.x * kPreBlock + threadIdx.x; r < n_zero; r += gridDim.x * kPreBlock) zero_words[r] = 0u;
  __shared__ K3Lds l;
  const int tid = threadIdx.x, nblk = (int)gridDim.x, own = (int)blockIdx.x;
  uint32_t heavy_n;
  k3_prologue<true>(l, own, nblk, P, gx, T, per, g, super, total_mirror, heavy_n);
  const uint32_t total = l.excl[kPreBlock];
  if (!(super && share_max > 0 && thr != 0u && heavy_n != 0u)) {        // (uniform; the benchmark's frames list nothing)
    k3_emit(l, own, 0u, total, gx, per, cap, tile_keys, vals);
    return;
  }
  // ---- rare path: K1's list of heavy blocks (the same in every workgroup), at most share_max of them ----------------------
  const int nh = (int)min(min(heavy_n, (uint32_t)kMaxHeavy), (uint32_t)share_max);
  if (tid < nh) {
    l.hl_blk[tid] = super[kHeavyRow + 1 + 2 * tid];
    l.hl_pe[tid] = super[kHeavyRow + 2 + 2 * tid] - thr;                 // the block's excess (K1 listed it: sum > thr)
  }
  __syncthreads();
  const uint32_t e = tid < nh ? l.hl_pe[tid] : 0u;
  const bool mine_listed = tid < nh && l.hl_blk[tid] == (uint32_t)own;
  uint32_t E;
  const uint32_t pe = k3_wg_scan_excl(e, l.scan_tmp, E);
  if (tid < nh) l.hl_pe[tid] = pe;
  if (tid == 0) l.hl_pe[nh] = E;
  const bool own_listed = __syncthreads_or(mine_listed) != 0;          // (also orders the list for the loop below)
  k3_emit(l, own, 0u, own_listed ? thr : total, gx, per, cap, tile_keys, vals);
  // this workgroup's share of the excess: slots [lo, hi) of the listed blocks' excesses laid end to end
  const uint64_t q = max((uint64_t)kShareMin, ((uint64_t)E + (uint64_t)nblk - 1u) / (uint64_t)nblk);
  const uint64_t lo = (uint64_t)own * q, hi = min(lo + q, (uint64_t)E);
  if (lo >= hi) return;
#pragma nounroll
  for (int i = 0; i < nh; ++i) {
    // (LDS reads land in vector registers: readfirstlane tells the compiler that these are the same in every lane --
    // with a per-lane block number the second prologue's address arithmetic doubled the kernel's register count)
    const uint64_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.hl_pe[i]);
    const uint64_t p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.hl_pe[i + 1]);
    if (p1 <= lo) continue;
    if (p0 >= hi) break;
    const int blk = __builtin_amdgcn_readfirstlane((int)l.hl_blk[i]);
    const uint32_t s0 = thr + (uint32_t)(max(lo, p0) - p0), s1 = thr + (uint32_t)(min(hi, p1) - p0);
    __syncthreads();                                       // the previous emission has read its prologue
    k3_prologue_other(&l, blk, nblk, P, gx, T, per, g.rects, g.block_sums, g.block_band, super);
    k3_emit(l, blk, s0, s1, gx, per, cap, tile_keys, vals);
  }
}

// ---------------------------------------------------------------------------------------------
// Per-tile depth sort.  vals[r0..r1) holds a tile's Gaussian ids in ascending order; afterwards it
// holds them ordered by (depth bits, id).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kLargeCap = 16384;   // 1024-thread workgroup, 128 KiB LDS

// ---- register-resident bitonic sort: one WAVE per tile, E keys per lane, no LDS, no barriers ----------------
// Logical element index i = lane * E + e.  Compare-exchange partners at distance j < E sit in the same lane
// (pure VALU); at distance j >= E they sit in lane ^ (j / E) and are fetched with a 64-bit lane shuffle.  The
// A network kept in LDS pays a workgroup barrier per stage (45 stages for 512 keys) and is latency-bound; here the
// stages of a wave simply follow each other.  Keys are unique ((depth bits << 32) | id), so the unstable network
// yields the stable (depth, id) order.
template <typename T>
__device__ __forceinline__ void cmp_swap(T& a, T& b, bool up) {
  const bool sw = (a > b) == up;
  const T lo = sw ? b : a, hi = sw ? a : b;
  a = lo;
  b = hi;
}

// Value of lane ^ LX.  ds_bpermute (what __shfl_xor compiles to) goes through the LDS crossbar and its issue rate
// bounds this kernel, so every distance that has a register-file path uses it: DPP quad permutes (1, 2), DPP row
// rotate by half a row (8), v_permlane16_swap / v_permlane32_swap (16, 32); only 4 is left to the crossbar.
template <int LX>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v, int lane) {
  if constexpr (LX == 1) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
  } else if constexpr (LX == 2) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
  } else if constexpr (LX == 8) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, true);    // row_ror:8
  } else if constexpr (LX == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);          // r[0] = rows (0,0,2,2), r[1] = (1,1,3,3)
    return (lane & 16) ? r[0] : r[1];
  } else if constexpr (LX == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);          // r[0] = (lo, lo), r[1] = (hi, hi)
    return (lane & 32) ? r[0] : r[1];
  } else {
    return (uint32_t)__shfl_xor((int)v, LX, 64);
  }
}
template <int LX>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v, int lane) {
  return ((uint64_t)lane_xor32<LX>((uint32_t)(v >> 32), lane) << 32) | lane_xor32<LX>((uint32_t)v, lane);
}

template <int LX>
__device__ __forceinline__ uint64_t lane_xor(uint64_t v, int lane) { return lane_xor64<LX>(v, lane); }
template <int LX>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) { return lane_xor32<LX>(v, lane); }

// One compare-exchange stage (K = bitonic block size, J = partner distance) of the network over i = lane * E + e.
// T: uint64_t ((depth bits, id) keys) or uint32_t (the depth bits of a sample, below).
template <int E, int K, int J, typename T>
__device__ __forceinline__ void wave_sort_stage(T (&key)[E], int lane, uint32_t base) {
  if constexpr (J >= E) {                             // partner in lane ^ (J / E)
    constexpr int LX = J / E;
    const bool lower = (lane & LX) == 0;
    const bool up = (base & (uint32_t)K) == 0;        // K >= 2E: the direction bit is a lane bit
    const bool keep_min = lower == up;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const T other = lane_xor<LX>(key[e], lane);
      const bool other_less = other < key[e];
      key[e] = (other_less == keep_min) ? other : key[e];
    }
  } else {                                            // partner in the same lane
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if ((e & J) == 0) {
        // direction bit of i = base + e: a bit of e while K < E (base is a multiple of E), a lane bit after
        const bool up = K < E ? ((e & K) == 0) : ((base & (uint32_t)K) == 0);
        cmp_swap(key[e], key[e | J], up);
      }
    }
  }
}
template <int E, int K, int J, typename T>
__device__ __forceinline__ void wave_sort_block(T (&key)[E], int lane, uint32_t base) {
  wave_sort_stage<E, K, J>(key, lane, base);
  if constexpr (J > 1) wave_sort_block<E, K, J / 2>(key, lane, base);
}
template <int E, int K, typename T>
__device__ __forceinline__ void wave_sort_network(T (&key)[E], int lane, uint32_t base) {
  wave_sort_block<E, K, K / 2>(key, lane, base);
  if constexpr (K < 64 * E) wave_sort_network<E, K * 2>(key, lane, base);
}

template <int E>
__device__ __forceinline__ void wave_sort_tile(const float* __restrict__ depths, uint32_t* __restrict__ vals,
                                               uint32_t r0, uint32_t n, int lane) {
  uint64_t key[E];
  // coalesced load (register slot e of lane l <- list entry e * 64 + l): the network sorts whatever permutation
  // it is given, only the OUTPUT position is tied to the logical index
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t i = (uint32_t)(e * 64 + lane);
    uint64_t k = ~0ull;
    if (i < n) {
      const uint32_t gid = vals[r0 + i];
      k = ((uint64_t)__float_as_uint(depths[gid]) << 32) | gid;
    }
    key[e] = k;
  }
  const uint32_t base = (uint32_t)lane * E;
  wave_sort_network<E, 2>(key, lane, base);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const uint32_t i = base + (uint32_t)e;
    if (i < n) vals[r0 + i] = (uint32_t)key[e];
  }
}

// ---- sample sort (round 6) -----------------------------------------------------------------------------------------------
// The network above costs 3.5 vector instructions per key at 1 024 keys and 4.5 at 4 096, pads every list to a power of
// two, and on frames of long lists its pair / quad classes serialise a workgroup: 0.27 ms of a 1.5 ms frame on a trained
// scene (lists of 1 300 on average, profiles/r05_config2_config3_scripts.log).  A tile's depths have no structure a
// radix digit could use, but they have QUANTILES, and between two quantiles they are smooth:
//   * 64 keys taken at equal strides of the list are sorted by one wave (a 32-bit network over the lanes, 21 stages) and,
//     with the list's smallest and largest depth at the ends, cut the depth axis into 64 COARSE buckets of about n / 64 keys
//     whatever the distribution is;
//   * a key finds its coarse bucket by a six-step binary search over the splitters in LDS and its FINE bucket -- one of F
//     per coarse bucket -- by interpolating between the two splitters: fine buckets hold one to three keys;
//   * an LDS fetch-and-add per key counts the fine bucket and hands the key a place inside it; one scan of the 64 F counts
//     later the keys are parked bucket by bucket in LDS, and every key counts the keys of ITS bucket that are smaller --
//     its exact rank -- and stores its id there.
// Exact for any input: the fine bucket is a monotone function of the depth bits, equal depths share a bucket, inside a
// bucket the full (depth bits, id) keys are compared.  A bucket of more than kSsMaxBucket keys (hundreds of equal depths)
// sends the tile to the network instead.
//   NW   waves that sort one tile together (1: the wave's own tile, no workgroup barrier; 4: the workgroup)
//   KPL  keys per lane (n <= 64 NW KPL)
//   F    fine buckets per coarse bucket; 64 F / (64 NW) counters per thread in the scan
// area: 64 NW KPL keys of LDS; ext: 65 words (+ 8 for NW = 4); start: 64 F + 2 HALF words (counts and places stay below
// 65 536: two counters share a word, the fetch-and-add goes to the word); tmp: scan scratch (NW = 4).
// Returns false (uniform over the group, nothing written) when a bucket is too full.
constexpr uint32_t kSsMaxBucket = 64;
__device__ __forceinline__ uint32_t sort_depth_bits(const float* __restrict__ depths, uint32_t gid) {
  return __float_as_uint(depths[gid]);       // (the gather costs 6 - 13 % of the kernel: profiles/r06_depth_sort.md)
}
template <int NW>
__device__ __forceinline__ void group_barrier() {
  if constexpr (NW == 1) {                       // (the DS operations of one wave execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, 64));
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
  return v;
}
template <int NW, int KPL, int F>
__device__ __forceinline__ bool sample_sort_tile(uint64_t* __restrict__ area, uint32_t* __restrict__ ext,
                                                 uint16_t* __restrict__ start, uint32_t* __restrict__ tmp,
                                                 const float* __restrict__ depths, uint32_t* __restrict__ vals,
                                                 uint32_t r0, uint32_t n, int wv) {
  constexpr int NT = 64 * NW, NB = 64 * F, CPT = NB / NT;
  static_assert(NW == 1 || NW == 4, "one wave, or the workgroup's four");
  static_assert(CPT >= 2 && CPT % 2 == 0 && CPT * NT == NB, "whole words of counters per thread");
  const int lane = threadIdx.x & 63, tg = wv * 64 + lane;
  uint32_t kd[KPL], kg[KPL];
#pragma unroll
  for (int e = 0; e < KPL; ++e) {               // coalesced; the sample's id is requested with them
    const uint32_t i = (uint32_t)(e * NT + tg);
    kg[e] = vals[r0 + min(i, n - 1u)];
  }
  uint32_t sg = 0u;
  if (wv == 0) sg = vals[r0 + ((uint32_t)lane * n) / 64u];
#pragma unroll
  for (int e = 0; e < KPL; ++e) kd[e] = sort_depth_bits(depths, kg[e]);     // (slots past n repeat the last key: harmless)
  // smallest / largest depth of the list: the outer ends of the first and the last coarse bucket
  uint32_t dmin = kd[0], dmax = kd[0];
#pragma unroll
  for (int e = 1; e < KPL; ++e) { dmin = min(dmin, kd[e]); dmax = max(dmax, kd[e]); }
  dmin = wave_min_u32(dmin);
  dmax = wave_max_u32(dmax);
  if constexpr (NW > 1) {
    if (lane == 0) { ext[65 + 2 * wv] = dmin; ext[66 + 2 * wv] = dmax; }
  }
  if (wv == 0) {
    uint32_t sd[1] = {sort_depth_bits(depths, sg)};
    wave_sort_network<1, 2>(sd, lane, (uint32_t)lane);
    if (lane < 63) ext[lane + 1] = sd[0];      // ext[1 .. 63]: the 63 splitters in use
  }
  uint32_t* start32 = reinterpret_cast<uint32_t*>(start);
  for (int t = tg; t <= NB / 2; t += NT) start32[t] = 0u;
  group_barrier<NW>();
  if constexpr (NW > 1) {
#pragma unroll
    for (int w = 0; w < NW; ++w) { dmin = min(dmin, ext[65 + 2 * w]); dmax = max(dmax, ext[66 + 2 * w]); }
  }
  // ---- coarse bucket = number of splitters below the key's depth (all keys of the lane step through the search TOGETHER:
  // one LDS round trip per step); fine bucket by interpolation between the bucket's two ends; place from the counter ----
  // keys that step together (registers: G members in flight): the largest divisor of KPL up to 8
  constexpr int G = KPL <= 8 ? KPL : (KPL % 8 == 0 ? 8 : (KPL % 6 == 0 ? 6 : (KPL % 5 == 0 ? 5 : 4)));
  static_assert(KPL % G == 0, "whole groups");
  uint32_t fb[KPL], rk[KPL];
#pragma unroll
  for (int g0 = 0; g0 < KPL; g0 += G) {
    uint32_t cb[G];
#pragma unroll
    for (int e = 0; e < G; ++e) cb[e] = 0u;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      uint32_t sv[G];
#pragma unroll
      for (int e = 0; e < G; ++e) sv[e] = ext[cb[e] + step];
#pragma unroll
      for (int e = 0; e < G; ++e) cb[e] += (sv[e] < kd[g0 + e]) ? (uint32_t)step : 0u;
    }
    uint32_t lo[G], hi[G];
#pragma unroll
    for (int e = 0; e < G; ++e) {
      lo[e] = ext[cb[e]];                       // (entries 0 and 64 are never written: the list's ends stand in below)
      hi[e] = ext[cb[e] + 1];
    }
#pragma unroll
    for (int e = 0; e < G; ++e) {
      const uint32_t l = cb[e] == 0u ? dmin : lo[e], h = cb[e] == 63u ? dmax : hi[e];
      // monotone in the depth bits: conversion, product with a positive factor and truncation all are
      const float t = (float)(kd[g0 + e] - l) * ((float)F * __builtin_amdgcn_rcpf((float)(h - l) + 1.0f));
      const uint32_t sub = min((uint32_t)t, (uint32_t)(F - 1));
      fb[g0 + e] = cb[e] * (uint32_t)F + sub;
    }
#pragma unroll
    for (int e = 0; e < G; ++e) {
      rk[g0 + e] = 0u;
      if ((uint32_t)((g0 + e) * NT + tg) < n) {
        const uint32_t sh = (fb[g0 + e] & 1u) << 4;
        rk[g0 + e] = (atomicAdd(&start32[fb[g0 + e] >> 1], 1u << sh) >> sh) & 0xffffu;
      }
    }
  }
  group_barrier<NW>();
  // ---- counts -> first places; a bucket that is too full sends the tile to the network ------------------------------------
  bool heavy;
  {
    uint32_t c[CPT], sum = 0u;
    bool big = false;
#pragma unroll
    for (int q = 0; q < CPT; q += 2) {
      const uint32_t w = start32[(tg * CPT + q) >> 1];
      c[q] = w & 0xffffu;
      c[q + 1] = w >> 16;
      sum += c[q] + c[q + 1];
      big = big || c[q] > kSsMaxBucket || c[q + 1] > kSsMaxBucket;
    }
    uint32_t run;
    if constexpr (NW == 1) {
      run = wave_scan_incl(sum) - sum;
      heavy = __ballot(big) != 0ull;
    } else {
      uint32_t total;
      run = k3_wg_scan_excl(sum, tmp, total);
      heavy = __syncthreads_or(big) != 0;
    }
#pragma unroll
    for (int q = 0; q < CPT; q += 2) {
      start32[(tg * CPT + q) >> 1] = run | ((run + c[q]) << 16);
      run += c[q] + c[q + 1];
    }
    if (tg == NT - 1) start[NB] = (uint16_t)run;
  }
  group_barrier<NW>();
  if (heavy) return false;
  // ---- park the keys bucket by bucket ------------------------------------------------------------------------------------------
  // sl: first place of the key's bucket (13 bits) | keys in the bucket (bits 13 .. 19) | bucket keys below this one (from
  // bit 20; counted below) -- one register per key instead of three (the 16-keys-per-lane instantiations spilled)
  static_assert(kSsMaxBucket < 128 && 64 * NW * KPL <= 8192, "the packed fields hold their values");
  uint32_t sl[KPL];
#pragma unroll
  for (int e = 0; e < KPL; ++e) {
    const uint32_t s0 = start[fb[e]], s1 = start[fb[e] + 1];
    sl[e] = ((uint32_t)(e * NT + tg) < n) ? (s0 | ((s1 - s0) << 13)) : 0u;
  }
#pragma unroll
  for (int e = 0; e < KPL; ++e)
    if ((uint32_t)(e * NT + tg) < n) area[(sl[e] & 0x1fffu) + rk[e]] = ((uint64_t)kd[e] << 32) | kg[e];
  group_barrier<NW>();
  // ---- exact rank = first place of the bucket + the bucket's keys below this one (buckets of one to three keys: the
  // lane's keys step through their buckets together) ------------------------------------------------------------------------
#pragma unroll
  for (int g0 = 0; g0 < KPL; g0 += G) {
    uint32_t len = 0u;
#pragma unroll
    for (int e = g0; e < g0 + G; ++e) len = max(len, (sl[e] >> 13) & 0x7fu);
    for (uint32_t k = 0; k < len; ++k) {
      uint64_t m[G];
#pragma unroll
      for (int e = 0; e < G; ++e) m[e] = area[min((sl[g0 + e] & 0x1fffu) + k, (uint32_t)(NT * KPL - 1))];
#pragma unroll
      for (int e = 0; e < G; ++e)
        sl[g0 + e] += (k < ((sl[g0 + e] >> 13) & 0x7fu) && m[e] < (((uint64_t)kd[g0 + e] << 32) | kg[g0 + e])) ? (1u << 20) : 0u;
    }
  }
#pragma unroll
  for (int e = 0; e < KPL; ++e)
    if ((uint32_t)(e * NT + tg) < n) vals[r0 + (sl[e] & 0x1fffu) + (sl[e] >> 20)] = kg[e];
  return true;
}

constexpr uint32_t kWaveCap = 1024;     // one wave, 16 keys per lane
constexpr uint32_t kQuadCap = 4096;     // the four waves of a workgroup together

// ---- 1025 .. 4096 instances: two or four waves of the workgroup sort one tile together -------------------------------
// Every wave sorts a block of 1024 keys in registers (the network above, blocks alternately ascending / descending: the
// direction bits of `base` include the wave), then the remaining stages of the bitonic network over 2048 and 4096 keys
// run: a stage whose partner distance is >= 1024 pairs keys of two WAVES and goes through LDS (write the block, barrier,
// read the partner's key: three such round trips in all), the stages below it are again in registers.  The LDS-only
// network used for larger tiles pays a workgroup barrier for each of its 78 stages at this size.
template <int K>
__device__ __forceinline__ void quad_cross_stage(uint64_t (&key)[16], uint64_t* lk, uint32_t base, int J) {
  __syncthreads();                              // the previous round trip's reads are done
#pragma unroll
  for (int e = 0; e < 16; ++e) lk[base + e] = key[e];
  __syncthreads();
  const bool lower = (base & (uint32_t)J) == 0;
  const bool up = (base & (uint32_t)K) == 0;
  const bool keep_min = lower == up;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const uint64_t other = lk[(base + e) ^ (uint32_t)J];
    const bool other_less = other < key[e];
    key[e] = (other_less == keep_min) ? other : key[e];
  }
}

// The merge levels above the waves' own 1024-key blocks: level K pairs keys of different waves at the distances K / 2 ..
// 1024 (one LDS round trip each) and finishes inside the waves (distances 512 .. 1).  Levels up to `kmax` run (uniform).
template <int K, int J>
__device__ __forceinline__ void coop_cross_stages(uint64_t (&key)[16], uint64_t* lk, uint32_t base) {
  quad_cross_stage<K>(key, lk, base, J);
  if constexpr (J > 1024) coop_cross_stages<K, J / 2>(key, lk, base);
}
template <int K, int KMAX>
__device__ __forceinline__ void coop_merge_levels(uint64_t (&key)[16], uint64_t* lk, uint32_t base, int lane, uint32_t kmax) {
  if ((uint32_t)K > kmax) return;
  coop_cross_stages<K, K / 2>(key, lk, base);
  wave_sort_block<16, K, 512>(key, lane, base);
  if constexpr (K < KMAX) coop_merge_levels<K * 2, KMAX>(key, lk, base, lane, kmax);
}

// GROUP waves sort one tile of up to 1024 GROUP instances together (`wv` = the wave's number inside its group, `lk` = the
// group's part of the LDS array, 1024 GROUP keys):
//   GROUP = 4   the workgroup's four waves, 2049 .. 4096 instances;
//   GROUP = 2   a PAIR of waves, 1025 .. 2048 -- the two pairs of a workgroup sort two such tiles at the same time (four
//               waves on one 2048-key tile leave two of them sorting padding); n == 0: the pair has no tile this round
//               and only keeps the workgroup's barriers company;
//   GROUP = 8 / 16  the 512 / 1024 lanes of the oversized-classes launches, up to 8192 / 16384 instances; only the merge
//               levels up to the next power of two above n run (the waves above it hold padding and merge it among
//               themselves).  The LDS-only network this replaces pays a workgroup barrier and a pass over all keys in
//               LDS for each of its 105 stages at 16 Ki keys; here 10 stages cross waves.
template <int GROUP>
__device__ __forceinline__ void coop_sort_tile(uint64_t* lk, const float* __restrict__ depths,
                                               uint32_t* __restrict__ vals, uint32_t r0, uint32_t n, int wv) {
  static_assert(GROUP == 2 || GROUP == 4 || GROUP == 8 || GROUP == 16, "pair, quad, 512 or 1024 lanes");
  const int lane = threadIdx.x & 63;
  if (GROUP == 2 && n == 0) {                   // (uniform per pair) the one cross stage's two barriers
    __syncthreads();
    __syncthreads();
    return;
  }
  uint64_t key[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {                // coalesced load: any permutation inside the wave's block will do
    const uint32_t i = (uint32_t)(wv * 1024 + e * 64 + lane);
    uint64_t k = ~0ull;
    if (i < n) {
      const uint32_t gid = vals[r0 + i];
      k = ((uint64_t)__float_as_uint(depths[gid]) << 32) | gid;
    }
    key[e] = k;
  }
  const uint32_t base = (uint32_t)(wv * 1024 + lane * 16);
  uint32_t kmax = 1024u * GROUP;
  if constexpr (GROUP >= 8) {
    kmax = 2048u;
    while (kmax < n) kmax <<= 1;
  }
  wave_sort_network<16, 2>(key, lane, base);                  // blocks of 1024, directions by bit 10 of the index
  coop_merge_levels<2048, 1024 * GROUP>(key, lk, base, lane, kmax);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const uint32_t i = base + (uint32_t)e;
    if (i < n) vals[r0 + i] = (uint32_t)key[e];
  }
}

// Size-unbounded fallback: the NW waves of a workgroup run a stable LSD radix sort (4 x 8 bits of the id, then 4 x 8 bits
// of the depth key, id as payload) through the tile's own slice of two global scratch arrays.  Slow (one workgroup per
// such tile) but it needs 2 KB + NW KB of LDS whatever the tile holds.  Used by the 1024-lane kernel for tiles above
// kLargeCap (NW = 16) and by the one-wave-per-tile kernel for the rare crowded tile of a light frame (NW = 4, below).
template <int NW>
struct RadixLds {
  uint32_t hist[256];
  uint32_t base[256];
  uint32_t wave_cnt[NW][256];
};
template <int NW>
__device__ __forceinline__ void radix_sort_tile(RadixLds<NW>& l, uint32_t r0, uint32_t n,
                                                const float* __restrict__ depths, uint32_t* __restrict__ vals,
                                                uint32_t* __restrict__ scratch_k, uint32_t* __restrict__ scratch_v,
                                                uint32_t* __restrict__ scratch_k2) {
  constexpr uint32_t NT = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __syncthreads();
  uint32_t* kA = scratch_k + r0;    // keys ping
  uint32_t* kB = scratch_k2 + r0;   // keys pong
  uint32_t* vA = vals + r0;         // values ping (final result lands here: 8 passes = even)
  uint32_t* vB = scratch_v + r0;    // values pong
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // 8 stable passes: first by id (the tile binning does not order a tile's entries), then by depth bits
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = (pass & 3) * 8;
    if (pass == 0) {
      for (uint32_t i = tid; i < n; i += NT) kA[i] = vA[i];
      __syncthreads();
    } else if (pass == 4) {     // 4 passes done: the data is back in vA / kA
      for (uint32_t i = tid; i < n; i += NT) kA[i] = __float_as_uint(depths[vA[i]]);
      __syncthreads();
    }
    if (tid < 256) l.hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += NT) atomicAdd(&l.hist[(kA[i] >> shift) & 255u], 1u);
    __syncthreads();
    {   // exclusive scan of the 256 digit counts by the first four waves (one lane walking them was a third of a pass)
      const uint32_t v = tid < 256 ? l.hist[tid] : 0u;
      const uint32_t inc = wave_scan_incl(v);
      if (tid < 256 && lane == 63) l.wave_cnt[0][wave] = inc;          // (wave_cnt is cleared again before its own use)
      __syncthreads();
      if (tid < 256) {
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += l.wave_cnt[0][w];
        l.base[tid] = before + inc - v;
      }
    }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n; c0 += NT) {
      const uint32_t i = c0 + tid;
      const bool valid = i < n;
      const uint32_t key = valid ? kA[i] : 0u;
      const uint32_t val = valid ? vA[i] : 0u;
      const uint32_t d = (key >> shift) & 255u;
      uint64_t same = __ballot(valid);
      if (!valid) same = 0;
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const uint64_t bal = __ballot((d >> bit) & 1u);
        same &= ((d >> bit) & 1u) ? bal : ~bal;
      }
      for (int t = tid; t < NW * 256; t += NT) (&l.wave_cnt[0][0])[t] = 0;
      __syncthreads();
      if (valid && (same & lt_mask) == 0) l.wave_cnt[wave][d] = (uint32_t)__popcll(same);
      __syncthreads();
      uint32_t before = 0;                       // same digit in earlier waves of this chunk
      if (valid) for (int w = 0; w < wave; ++w) before += l.wave_cnt[w][d];
      const uint32_t pos = valid ? l.base[d] + before + (uint32_t)__popcll(same & lt_mask) : 0u;
      __syncthreads();
      if (tid < 256) {                           // advance the digit bases by this chunk's counts
        uint32_t add = 0;
        for (int w = 0; w < NW; ++w) add += l.wave_cnt[w][tid];
        l.base[tid] += add;
      }
      if (valid) { kB[pos] = key; vB[pos] = val; }
      __syncthreads();
    }
    uint32_t* t;
    t = kA; kA = kB; kB = t;
    t = vA; vA = vB; vB = t;
    __syncthreads();
  }
}

// One wave per tile (4 tiles per workgroup) up to kWaveCap instances.
// QUAD (chosen by the host when the frame's mean list length says long lists are common; its 32 KiB of LDS cost the
// one-wave path a fifth of its occupancy): a tile with up to kQuadCap is then sorted by the workgroup's four waves
// together, larger tiles are filed into the class lists for the oversized-classes launches that follow:
// big[0] / big[1] = number of large / huge tiles; big + 3: large list [T], huge list [T].
// !QUAD (light frames): a crowded tile (n > kWaveCap) is rare; the workgroup that meets one sorts it itself with the
// radix fallback once its waves are done with their own tiles -- the lists stay empty and NO further launch follows
// (rounds 1-4 launched the 1024-lane kernel with 146 KB of LDS per workgroup behind every frame to find its lists empty:
// 5 us and a kernel boundary).
template <bool QUAD>
__device__ __forceinline__ void tile_depth_sort_wave_body(const uint32_t* __restrict__ ranges,
                                                          const float* __restrict__ depths,
                                                          uint32_t* __restrict__ vals, uint32_t* big, int T,
                                                          uint32_t* __restrict__ tile_ids,
                                                          uint32_t* __restrict__ scratch_k,
                                                          uint32_t* __restrict__ scratch_v,
                                                          uint32_t* __restrict__ scratch_k2) {
  // the sample sort's key area: kOwn keys per wave for the waves' own tiles; QUAD: the same 32 KB hold a tile of up to
  // kQuadCap keys that the four waves sort together (and the cross-wave stages of the network, when a tile falls back);
  // light frames: the radix fallback's arrays lie over it (it runs when the waves are done with their own tiles)
  constexpr uint32_t kOwn = QUAD ? kWaveCap : 512u;
  constexpr uint32_t kOwnMax = QUAD ? kOwn : kWaveCap;          // longest list a wave sorts alone
  constexpr size_t kAreaBytes = (QUAD ? (size_t)kQuadCap : 4 * (size_t)kOwn) * sizeof(uint64_t);
  static_assert(!QUAD || 4 * kOwn <= kQuadCap, "the waves' own areas fit the workgroup's");
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[QUAD ? kAreaBytes : (kAreaBytes > sizeof(RadixLds<4>) ? kAreaBytes : sizeof(RadixLds<4>))];
  uint64_t* lk = reinterpret_cast<uint64_t*>(lds_raw);
  constexpr int kFOwn = 8, kFCoop = 32;                   // fine buckets per coarse one: 512 per wave, 2 048 per workgroup
  __shared__ uint32_t ss_ext[4 * 80], ss_tmp[4];
  __shared__ __attribute__((aligned(4))) uint16_t ss_start[64 * kFCoop + 8];
  static_assert(4 * (64 * kFOwn + 2) <= 64 * kFCoop + 8, "the waves' own counters fit the workgroup's");
  __shared__ uint32_t quad_n[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wave;
  uint32_t r0 = 0, n = 0;
  if (tile < T) {
    r0 = ranges[tile * 2 + 0];
    n = ranges[tile * 2 + 1] - r0;
  }
  if (tile_ids)          // sorted tile-id column (introspection / parity tests) when no global sort produced it
    for (uint32_t i = (uint32_t)lane; i < n; i += 64u) tile_ids[r0 + i] = (uint32_t)tile;
  if (lane == 0) {
    if (QUAD) {
      quad_n[wave] = (n > kOwnMax && n <= kQuadCap) ? n : 0u;
      if (n > kQuadCap) {
        const int cls = n > kLargeCap ? 1 : 0;
        const uint32_t slot = atomicAdd(&big[cls], 1u);
        big[3 + cls * T + slot] = (uint32_t)tile;
      }
    } else {
      quad_n[wave] = n > kWaveCap ? n : 0u;
    }
  }
  // ---- the wave's own tile: the network up to 128 keys, the sample sort above (the network where a bucket overflows) ----
  if (n > 1 && n <= kOwnMax) {
    uint64_t* area = lk + wave * kOwn;
    uint32_t* ex = ss_ext + wave * 80;
    uint16_t* st = ss_start + wave * (64 * kFOwn + 2);
    // keys per lane in steps of a third: a list fills at least three quarters of the lanes' slots
    bool ok = true;
    if (n <= 128) wave_sort_tile<2>(depths, vals, r0, n, lane);
    else if (n <= 192) ok = sample_sort_tile<1, 3, kFOwn>(area, ex, st, nullptr, depths, vals, r0, n, 0);
    else if (n <= 256) ok = sample_sort_tile<1, 4, kFOwn>(area, ex, st, nullptr, depths, vals, r0, n, 0);
    else if (n <= 384) ok = sample_sort_tile<1, 6, kFOwn>(area, ex, st, nullptr, depths, vals, r0, n, 0);
    else if (n <= 512) ok = sample_sort_tile<1, 8, kFOwn>(area, ex, st, nullptr, depths, vals, r0, n, 0);
    else if (!QUAD) wave_sort_tile<16>(depths, vals, r0, n, lane);      // (rare in a light frame)
    else if (n <= 768) ok = sample_sort_tile<1, 12, kFOwn>(area, ex, st, nullptr, depths, vals, r0, n, 0);
    else ok = sample_sort_tile<1, 16, kFOwn>(area, ex, st, nullptr, depths, vals, r0, n, 0);
    if (!ok) {                                            // a bucket overflowed: the network
      if (lane == 0) atomicAdd(&big[2], 1u);
      if (n <= 256) wave_sort_tile<4>(depths, vals, r0, n, lane);
      else if (n <= 512) wave_sort_tile<8>(depths, vals, r0, n, lane);
      else wave_sort_tile<16>(depths, vals, r0, n, lane);
    }
  }
  if constexpr (QUAD) {
    __syncthreads();
    // the workgroup's tiles of 1025 .. 4096, one after the other, all four waves: uniform control flow, every wave meets
    // every barrier
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      const uint32_t qn = quad_n[w];                      // (uniform)
      if (qn == 0u) continue;
      const uint32_t qr0 = ranges[(blockIdx.x * 4 + w) * 2 + 0];
      bool ok;
      if (qn <= 1280u) ok = sample_sort_tile<4, 5, kFCoop>(lk, ss_ext, ss_start, ss_tmp, depths, vals, qr0, qn, wave);
      else if (qn <= 1536u) ok = sample_sort_tile<4, 6, kFCoop>(lk, ss_ext, ss_start, ss_tmp, depths, vals, qr0, qn, wave);
      else if (qn <= 2048u) ok = sample_sort_tile<4, 8, kFCoop>(lk, ss_ext, ss_start, ss_tmp, depths, vals, qr0, qn, wave);
      else if (qn <= 2560u) ok = sample_sort_tile<4, 10, kFCoop>(lk, ss_ext, ss_start, ss_tmp, depths, vals, qr0, qn, wave);
      else if (qn <= 3072u) ok = sample_sort_tile<4, 12, kFCoop>(lk, ss_ext, ss_start, ss_tmp, depths, vals, qr0, qn, wave);
      else ok = sample_sort_tile<4, 16, kFCoop>(lk, ss_ext, ss_start, ss_tmp, depths, vals, qr0, qn, wave);
      if (!ok) {
        if (threadIdx.x == 0) atomicAdd(&big[2], 1u);
        __syncthreads();
        coop_sort_tile<4>(lk, depths, vals, qr0, qn, wave);
      }
      __syncthreads();
    }
  } else {
    // a light frame's rare crowded tile: the radix fallback through global scratch, whatever the tile holds (the four-wave
    // sample sort in here would take the kernel from 71 to 131 registers -- 7 to 3 waves per SIMD for every light frame)
    RadixLds<4>& rl = *reinterpret_cast<RadixLds<4>*>(lds_raw);
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      const uint32_t cn = quad_n[w];                      // (uniform)
      if (cn) radix_sort_tile<4>(rl, ranges[(blockIdx.x * 4 + w) * 2 + 0], cn, depths, vals, scratch_k, scratch_v, scratch_k2);
    }
  }
}

// (two kernels for the two bodies: the register budgets differ -- 32 KB of LDS per workgroup hold the QUAD kernel at
// four waves per SIMD anyway, so it is told to fit them)
constexpr int kSortQuadWaves = 4;
template <bool QUAD>
__global__ __launch_bounds__(256) void tile_depth_sort_wave_kernel(const uint32_t* __restrict__ ranges,
                                                                   const float* __restrict__ depths,
                                                                   uint32_t* __restrict__ vals, uint32_t* big, int T,
                                                                   uint32_t* __restrict__ tile_ids,
                                                                   uint32_t* __restrict__ scratch_k,
                                                                   uint32_t* __restrict__ scratch_v,
                                                                   uint32_t* __restrict__ scratch_k2) {
  tile_depth_sort_wave_body<false>(ranges, depths, vals, big, T, tile_ids, scratch_k, scratch_v, scratch_k2);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(kSortQuadWaves, kSortQuadWaves))) void
tile_depth_sort_quad_kernel(const uint32_t* __restrict__ ranges, const float* __restrict__ depths,
                            uint32_t* __restrict__ vals, uint32_t* big, int T, uint32_t* __restrict__ tile_ids,
                            uint32_t* __restrict__ scratch_k, uint32_t* __restrict__ scratch_v,
                            uint32_t* __restrict__ scratch_k2) {
  tile_depth_sort_wave_body<true>(ranges, depths, vals, big, T, tile_ids, scratch_k, scratch_v, scratch_k2);
}

// The large class (kQuadCap < n <= kLargeCap, filed by the QUAD kernel): one workgroup of 64 GROUP lanes per listed tile,
// the register-block sort above.  Large tiles outside (lo, hi] are left to the other launch.
template <int GROUP>
__device__ __forceinline__ void sort_large_tiles(unsigned char* smem, const uint32_t* __restrict__ ranges,
                                                 const float* __restrict__ depths, uint32_t* __restrict__ vals,
                                                 const uint32_t* __restrict__ big, int T, uint32_t lo, uint32_t hi) {
  const uint32_t count = big[0];
  for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
    const uint32_t tile = big[3 + e];
    const uint32_t r0 = ranges[tile * 2 + 0], r1 = ranges[tile * 2 + 1];
    if (r1 - r0 <= lo || r1 - r0 > hi) continue;          // (uniform)
    __syncthreads();
    coop_sort_tile<GROUP>(reinterpret_cast<uint64_t*>(smem), depths, vals, r0, r1 - r0, (int)(threadIdx.x >> 6));
  }
}

__device__ __forceinline__ void sort_huge_tiles(const uint32_t* __restrict__ ranges,
                                                const float* __restrict__ depths, uint32_t* __restrict__ vals,
                                                uint32_t* __restrict__ scratch_k, uint32_t* __restrict__ scratch_v,
                                                uint32_t* __restrict__ scratch_k2,
                                                const uint32_t* __restrict__ big, int T) {
  __shared__ RadixLds<16> l;
  const uint32_t count = big[1];
  for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
    const uint32_t tile = big[3 + T + e];
    const uint32_t r0 = ranges[tile * 2 + 0], r1 = ranges[tile * 2 + 1];
    radix_sort_tile<16>(l, r0, r1 - r0, depths, vals, scratch_k, scratch_v, scratch_k2);
  }
}

// The oversized classes.  A light frame (their lists are almost always empty: a launch each would cost more than the
// work) takes ONE launch of the 1024-lane kernel for all of them: 128 KiB of dynamic LDS for the large class + 18 KiB
// static for the huge class.  A heavy frame (mean list above 1024: thousands of tiles of 4 .. 16 Ki instances -- coarse
// LOD cuts, big footprints) adds the 512-lane kernel in front for the medium tiles and the large tiles up to 8192: 64
// KiB of LDS, so TWO tiles per CU are in flight (a tile's sort is a chain of barriers and cross-lane stages: latency,
// not throughput), and the 1024-lane kernel keeps the tiles above 8192 and the huge class.
constexpr uint32_t kMidCap = 8192;
__global__ __launch_bounds__(512) void tile_depth_sort_mid_kernel(const uint32_t* __restrict__ ranges,
                                                                  const float* __restrict__ depths,
                                                                  uint32_t* __restrict__ vals,
                                                                  const uint32_t* __restrict__ big, int T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  sort_large_tiles<8>(smem, ranges, depths, vals, big, T, 0u, kMidCap);
}

__global__ __launch_bounds__(1024) void tile_depth_sort_big_kernel(const uint32_t* __restrict__ ranges,
                                                                   const float* __restrict__ depths,
                                                                   uint32_t* __restrict__ vals,
                                                                   uint32_t* __restrict__ scratch_k,
                                                                   uint32_t* __restrict__ scratch_v,
                                                                   uint32_t* __restrict__ scratch_k2,
                                                                   const uint32_t* __restrict__ big, int T,
                                                                   uint32_t large_lo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  sort_large_tiles<16>(smem, ranges, depths, vals, big, T, large_lo, kLargeCap);
  __syncthreads();
  sort_huge_tiles(ranges, depths, vals, scratch_k, scratch_v, scratch_k2, big, T);
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t* __restrict__ keys, uint32_t L_cap,
                                                          const uint32_t* __restrict__ L_dev,
                                                          uint32_t* __restrict__ ranges, uint32_t* __restrict__ big) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t L = L_dev ? min(*L_dev, L_cap) : L_cap;
  if (i == 0) { big[0] = 0; big[1] = 0; big[2] = 0; }   // class lists of the depth sort that follows
  if (i >= L) return;
  const uint32_t t = keys[i];
  if (i == 0) {
    ranges[t * 2 + 0] = 0;
  } else {
    const uint32_t tp = keys[i - 1];
    if (tp != t) {
      ranges[tp * 2 + 1] = i;
      ranges[t * 2 + 0] = i;
    }
  }
  if (i == L - 1) ranges[t * 2 + 1] = L;
}

}  // namespace

// HGS_K3_SHARE: how many heavy blocks K3 shares out at most (0: off, K1 then lists nothing; default and maximum kMaxHeavy) -- tests and A/B runs
static int k3_share_max() {
  static const int v = [] {
    const char* e = getenv("HGS_K3_SHARE");
    const int x = e ? atoi(e) : kMaxHeavy;
    return x < 0 ? 0 : (x > kMaxHeavy ? kMaxHeavy : x);
  }();
  return v;
}
uint32_t k3_heavy_threshold(uint32_t L_cap, int32_t P) {
  const uint32_t nblk = (uint32_t)((P + kPreBlock - 1) / kPreBlock);
  if (nblk == 0u || k3_share_max() == 0) return 0u;
  const uint64_t t = (uint64_t)kHeavyFactor * (((uint64_t)L_cap + nblk - 1u) / nblk);
  return (uint32_t)(t > 0x7fffffffull ? 0x7fffffffull : (t < kHeavyFloor ? kHeavyFloor : t));
}
int launch_duplicate_tiles(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, uint32_t L_cap, bool banded,
                           hipStream_t s, const uint32_t* super, uint32_t* total_mirror) {
  const int nblk = (a.P + kPreBlock - 1) / kPreBlock;
  if (super && !(banded && nblk > 0 && L_cap > 0)) { set_error("duplicate_tiles: superblock totals without the banded path"); return HGS_ERR_INVALID; }
  if (nblk > 0 && L_cap > 0) {
    const int T = grid_x(a.width) * grid_y(a.height);
    if (banded)
      hipLaunchKernelGGL(duplicate_tiles_banded_kernel, dim3(nblk), dim3(kPreBlock), 0, s, a.P, grid_x(a.width), T,
                         band_tiles(T), g, L_cap, b.keys_in, b.vals_in, b.ranges, T * 2, super, total_mirror,
                         tile_bin_zero_words(b.sort_tmp, L_cap, T), tile_bin_zero_count(T), k3_share_max(),
                         super ? k3_heavy_threshold(L_cap, a.P) : 0u);
    else
      hipLaunchKernelGGL(duplicate_tiles_kernel, dim3(nblk), dim3(kPreBlock), 0, s, a.P, grid_x(a.width), g, L_cap,
                         b.keys_in, b.vals_in, b.ranges, T * 2);
    HGS_LAUNCH_CHECK("duplicate_tiles", s, a.debug);
  } else {
    HGS_HIP(hipMemsetAsync(b.ranges, 0, (size_t)grid_x(a.width) * grid_y(a.height) * 2 * sizeof(uint32_t), s));
  }
  return HGS_OK;
}

int launch_tile_ranges(const BinWs& b, uint32_t L_cap, const uint32_t* L_dev, int32_t T, hipStream_t s, bool debug) {
  // b.ranges was zeroed by launch_duplicate_tiles
  if (L_cap > 0) {
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((L_cap + 255) / 256), dim3(256), 0, s, b.keys_out, L_cap, L_dev, b.ranges,
                       b.big_tiles);
    HGS_LAUNCH_CHECK("tile_ranges", s, debug);
  }
  return HGS_OK;
}

int launch_tile_depth_sort(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, uint32_t L, int32_t T,
                           bool fill_tile_ids, hipStream_t s) {
  if (L == 0) return HGS_OK;
  static bool attr_set = false;
  if (!attr_set) {   // 128 KiB of dynamic LDS needs an explicit opt-in
    HGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_depth_sort_big_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLargeCap * 8)));
    HGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_depth_sort_mid_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kMidCap * 8)));
    attr_set = true;
  }
  // lists of more than 1024 instances are common when the mean list is long: then the four-wave variant, and the launches
  // for the classes it files.  A light frame's rare crowded tile is sorted inside the one-wave kernel: one launch.
  // scratch: keys_in / vals_in and the radix sort's alternate key buffer are free once the tile sort is done
  const bool quad = (uint64_t)L > (uint64_t)T * 512u;
  auto kern = quad ? tile_depth_sort_quad_kernel : tile_depth_sort_wave_kernel<false>;
  hipLaunchKernelGGL(kern, dim3((T + 3) / 4), dim3(256), 0, s, b.ranges, g.depths, b.vals_out, b.big_tiles, T,
                     fill_tile_ids ? b.keys_out : nullptr, b.keys_in, b.vals_in, reinterpret_cast<uint32_t*>(b.sort_tmp));
  HGS_LAUNCH_CHECK("tile_depth_sort_wave", s, a.debug);
  static const bool dbg = getenv("HGS_SORT_DEBUG") != nullptr;       // diagnostic: tiles the sample sort handed to the network
  if (dbg) {
    uint32_t c[3] = {0, 0, 0};
    if (hipMemcpyAsync(c, b.big_tiles, sizeof(c), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess)
      fprintf(stderr, "[hgs] tile_depth_sort: L %u T %d quad %d: large %u huge %u network fallbacks %u\n", L, T, (int)quad, c[0], c[1], c[2]);
  }
  if (!quad) return HGS_OK;
  const int big_grid = T < 256 ? T : 256;
  const bool heavy = (uint64_t)L > (uint64_t)T * 1024u;
  if (heavy) {
    hipLaunchKernelGGL(tile_depth_sort_mid_kernel, dim3(T < 512 ? T : 512), dim3(512), kMidCap * 8, s, b.ranges, g.depths,
                       b.vals_out, b.big_tiles, T);
    HGS_LAUNCH_CHECK("tile_depth_sort_mid", s, a.debug);
  }
  hipLaunchKernelGGL(tile_depth_sort_big_kernel, dim3(big_grid), dim3(1024), kLargeCap * 8, s, b.ranges, g.depths,
                     b.vals_out, b.keys_in, b.vals_in, reinterpret_cast<uint32_t*>(b.sort_tmp), b.big_tiles, T,
                     heavy ? kMidCap : 0u);
  HGS_LAUNCH_CHECK("tile_depth_sort_big", s, a.debug);
  return HGS_OK;
}

// Launch order of the one-wave-per-tile kernels (K6, K7).  Workgroup b runs on XCD b % 8 and every XCD has its own L2,
// so the tiles stay split into 8 contiguous BANDS (band x = tiles [x * per, (x + 1) * per), per = ceil(T / 8): neighbouring
// tiles share Gaussians and therefore that XCD's L2), and inside its band every XCD takes the tiles by DESCENDING
// instance count: the crowded tiles start first and the tail of the launch is made of light tiles (the kernels run ~1.6
// rounds of waves; an unsorted launch ends with a few heavy tiles keeping a handful of SIMDs busy).  A global
// heavy-first order was measured first: same kernel time, but 40 % more HBM fetches (the record gathers lose the L2).
// One workgroup per band: counting sort over 1024 quantised counts (count / 4, everything above 4092 in the first
// bucket).  The order inside a bucket is whatever the LDS atomics produce: it only permutes which SIMD renders which
// tile, never a result.  order[b] = tile of workgroup b (0xffffffff: none).
namespace {
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ ranges, int T, int per,
                                                          uint32_t* __restrict__ order) {
  __shared__ uint32_t hist[1024];
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int band = blockIdx.x;
  const int t0 = band * per, t1 = min(T, t0 + per);
  hist[tid] = 0;
  __syncthreads();
  for (int t = t0 + tid; t < t1; t += 1024) {
    const uint32_t c = ranges[2 * t + 1] - ranges[2 * t];
    atomicAdd(&hist[1023u - min(c >> 2, 1023u)], 1u);
  }
  __syncthreads();
  const uint32_t v = hist[tid];
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(inc, off, 64);
    if (lane >= off) inc += u;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  hist[tid] = base + inc - v;          // exclusive prefix = first slot of the bucket inside the band
  __syncthreads();
  for (int t = t0 + tid; t < t1; t += 1024) {
    const uint32_t c = ranges[2 * t + 1] - ranges[2 * t];
    const uint32_t k = atomicAdd(&hist[1023u - min(c >> 2, 1023u)], 1u);
    order[k * 8u + (uint32_t)band] = (uint32_t)t;          // workgroup b = k * 8 + band
  }
  for (int k = (t1 > t0 ? t1 - t0 : 0) + tid; k < per; k += 1024) order[(uint32_t)k * 8u + (uint32_t)band] = 0xffffffffu;
}
}  // namespace

int launch_tile_order(const BinWs& b, int32_t T, hipStream_t s, bool debug) {
  const int per = (T + 7) / 8;
  hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(1024), 0, s, b.ranges, T, per, b.tile_order);
  HGS_LAUNCH_CHECK("tile_order", s, debug);
  return HGS_OK;
}

}  // namespace hgs
