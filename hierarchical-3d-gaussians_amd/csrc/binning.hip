// K3 (duplicateWithKeys) and K5 (identifyTileRanges).
//
// The (tile | depth) sort is split: the P Gaussians are sorted by depth first (32-bit keys,
// 4 passes over P elements), K3 then emits the L (tile, Gaussian) instances in that order, and a
// stable sort by the tile id alone (13 bits at 1080p: 2 passes over L) finishes the job -- the
// result is identical to one stable 45-bit sort of L 64-bit keys (6 passes over L), at less than
// half the traffic.
//
// K3 is a load-balanced expansion: a workgroup owns 256 consecutive (depth-ordered) Gaussians,
// scans their tile counts in LDS and then assigns OUTPUT slots (not Gaussians) to lanes, so a
// Gaussian covering hundreds of tiles does not serialise one lane and the stores are coalesced.
#include "common.h"

namespace hgs {
namespace {

// Per-workgroup instance counts in DEPTH-SORTED Gaussian order (feeds the emission-offset scan).
__device__ __forceinline__ uint32_t rect_count(uint2 r) {     // rects are zero for culled Gaussians
  return ((r.y & 0xffffu) - (r.x & 0xffffu)) * ((r.y >> 16) - (r.x >> 16));
}

__global__ __launch_bounds__(kPreBlock) void sorted_block_sums_kernel(int P, const uint32_t* __restrict__ perm,
                                                                      const uint2* __restrict__ rects,
                                                                      uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t wave_tot[kPreBlock / 64];
  const int i = blockIdx.x * kPreBlock + threadIdx.x;
  uint32_t v = (i < P) ? rect_count(rects[perm[i]]) : 0u;   // one 8-byte gather per Gaussian
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kPreBlock / 64; ++w) t += wave_tot[w];
    block_sums[blockIdx.x] = t;
  }
}

// Emits (tile id, Gaussian id) for every tile of every visible Gaussian, walking the Gaussians in
// depth order (perm), row-major tiles inside a rectangle.  A stable sort of the result by tile id
// alone then equals the reference's stable sort by (tile | depth): depth ties keep ascending
// Gaussian index because the depth sort that produced perm is stable as well.
__global__ __launch_bounds__(kPreBlock) void duplicate_tiles_kernel(int P, int gx, GeomWs g,
                                                                    uint32_t* __restrict__ tile_keys,
                                                                    uint32_t* __restrict__ vals) {
  __shared__ uint32_t excl[kPreBlock + 1];
  __shared__ uint32_t gids[kPreBlock];
  __shared__ uint2 lrect[kPreBlock];
  __shared__ uint32_t wave_tot[kPreBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * kPreBlock + tid;
  const uint32_t gid = (i < P) ? g.perm[i] : 0u;
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
  gids[tid] = gid;
  if (tid == kPreBlock - 1) excl[kPreBlock] = wbase + inc;
  const uint32_t block_base = g.sorted_block_sums[blockIdx.x];
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
    const uint32_t gg = gids[lo];
    const uint32_t k = s - excl[lo];
    const uint2 rc = lrect[lo];
    const uint32_t minx = rc.x & 0xffffu, miny = rc.x >> 16;
    const uint32_t w = (rc.y & 0xffffu) - minx;
    const uint32_t ty = miny + k / w, tx = minx + k % w;
    tile_keys[block_base + s] = ty * (uint32_t)gx + tx;
    vals[block_base + s] = gg;
  }
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t* __restrict__ keys, uint32_t L,
                                                          uint32_t* __restrict__ ranges) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
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

int launch_sorted_block_sums(const hgs_raster_args& a, const GeomWs& g, hipStream_t s) {
  const int nblk = (a.P + kPreBlock - 1) / kPreBlock;
  if (nblk > 0) {
    hipLaunchKernelGGL(sorted_block_sums_kernel, dim3(nblk), dim3(kPreBlock), 0, s, a.P, g.perm,
                       reinterpret_cast<const uint2*>(g.rects), g.sorted_block_sums);
    HGS_LAUNCH_CHECK("sorted_block_sums", s, a.debug);
  }
  return HGS_OK;
}

int launch_duplicate_tiles(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, uint32_t L, hipStream_t s) {
  const int nblk = (a.P + kPreBlock - 1) / kPreBlock;
  if (nblk > 0 && L > 0) {
    hipLaunchKernelGGL(duplicate_tiles_kernel, dim3(nblk), dim3(kPreBlock), 0, s, a.P, grid_x(a.width), g,
                       b.keys_in, b.vals_in);
    HGS_LAUNCH_CHECK("duplicate_tiles", s, a.debug);
  }
  return HGS_OK;
}

int launch_tile_ranges(const BinWs& b, uint32_t L, int32_t T, hipStream_t s, bool debug) {
  HGS_HIP(hipMemsetAsync(b.ranges, 0, (size_t)T * 2 * sizeof(uint32_t), s));
  if (L > 0) {
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((L + 255) / 256), dim3(256), 0, s, b.keys_out, L, b.ranges);
    HGS_LAUNCH_CHECK("tile_ranges", s, debug);
  }
  return HGS_OK;
}

}  // namespace hgs
