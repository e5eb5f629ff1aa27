// K3 (duplicateWithKeys) and K5 (identifyTileRanges).
//
// K3 is a load-balanced expansion: a workgroup owns 256 consecutive Gaussians, scans
// their tile counts in LDS and then assigns OUTPUT slots (not Gaussians) to lanes, so
// a Gaussian covering hundreds of tiles does not serialise one lane and the key /
// value stores are fully coalesced.  Emission order = ascending Gaussian index, then
// row-major tiles inside the Gaussian's rectangle -- the order
// the stable sort's tie-breaking is defined on.
#include "common.h"

namespace hgs {
namespace {

__global__ __launch_bounds__(kPreBlock) void duplicate_keys_kernel(int P, int gx, GeomWs g,
                                                                   uint64_t* __restrict__ keys,
                                                                   uint32_t* __restrict__ vals) {
  __shared__ uint32_t excl[kPreBlock + 1];
  __shared__ uint32_t wave_tot[kPreBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int idx = blockIdx.x * kPreBlock + tid;
  const uint32_t cnt = (idx < P) ? g.tiles_touched[idx] : 0u;
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
  if (idx < P) g.offsets[idx] = block_base + my_excl;
  __syncthreads();
  const uint32_t total = excl[kPreBlock];
  for (uint32_t s = tid; s < total; s += kPreBlock) {
    // largest j with excl[j] <= s  (excl is non-decreasing; zero-count entries are skipped
    // because the search lands on the LAST index whose start is <= s)
    int lo = 0, hi = kPreBlock;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int mid = (lo + hi) >> 1;
      if (excl[mid] <= s) lo = mid; else hi = mid;
    }
    const int gid = blockIdx.x * kPreBlock + lo;
    const uint32_t k = s - excl[lo];
    const uint32_t rmin = g.rects[gid * 2 + 0], rmax = g.rects[gid * 2 + 1];
    const uint32_t minx = rmin & 0xffffu, miny = rmin >> 16;
    const uint32_t w = (rmax & 0xffffu) - minx;
    const uint32_t ty = miny + k / w, tx = minx + k % w;
    const uint64_t tile = (uint64_t)(ty * (uint32_t)gx + tx);
    keys[block_base + s] = (tile << 32) | (uint64_t)__float_as_uint(g.depths[gid]);
    vals[block_base + s] = (uint32_t)gid;
  }
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint64_t* __restrict__ keys, uint32_t L,
                                                          uint32_t* __restrict__ ranges) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= L) return;
  const uint32_t t = (uint32_t)(keys[i] >> 32);
  if (i == 0) {
    ranges[t * 2 + 0] = 0;
  } else {
    const uint32_t tp = (uint32_t)(keys[i - 1] >> 32);
    if (tp != t) {
      ranges[tp * 2 + 1] = i;
      ranges[t * 2 + 0] = i;
    }
  }
  if (i == L - 1) ranges[t * 2 + 1] = L;
}

}  // namespace

int launch_duplicate_keys(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, uint32_t L, hipStream_t s) {
  const int nblk = (a.P + kPreBlock - 1) / kPreBlock;
  if (nblk > 0 && L > 0) {
    hipLaunchKernelGGL(duplicate_keys_kernel, dim3(nblk), dim3(kPreBlock), 0, s, a.P, grid_x(a.width), g,
                       b.keys_in, b.vals_in);
    HGS_LAUNCH_CHECK("duplicate_keys", s, a.debug);
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
