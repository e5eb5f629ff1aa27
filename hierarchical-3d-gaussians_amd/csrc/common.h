// Internal helpers shared by the libhgs translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/hgs.h"

namespace hgs {

void set_error(const char* fmt, ...);

#define HGS_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t _e = (call);                                                             \
    if (_e != hipSuccess) {                                                             \
      hgs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      return HGS_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

// After a kernel launch: always pick up launch errors; in debug mode also
// synchronise so that an asynchronous fault is attributed to this kernel.
#define HGS_LAUNCH_CHECK(name, stream, debug)                                           \
  do {                                                                                  \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e == hipSuccess && (debug)) _e = hipStreamSynchronize(stream);                 \
    if (_e != hipSuccess) {                                                             \
      hgs::set_error("kernel %s failed: %s", name, hipGetErrorString(_e));              \
      return HGS_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

constexpr size_t kAlign = 256;
inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

template <typename T>
inline T* carve(char*& p, size_t count) {
  T* r = reinterpret_cast<T*>(p);
  p += align_up(count * sizeof(T));
  return r;
}

constexpr int kTile = HGS_TILE;
constexpr int kRecFloats = 16;              // per-Gaussian 2D record, 4 x float4 (one 64-byte line)
constexpr int kRecVec = kRecFloats / 4;
constexpr int kInstStride = HGS_INST_GRAD_STRIDE;
constexpr int kJacStride = 9;               // floats per row of GeomWs::shjac (36 bytes: rows are packed; 48-byte rows until round 6)
constexpr int kPreBlock = 256;              // Gaussians per preprocess / binning workgroup
// Tile BANDS: band x = tiles [x * per, (x + 1) * per), per = ceil(T / 8) -- the eighth of the frame that XCD x composites
// (render.hip: block_to_tile).  The binning keeps one instance stream per band and runs the kernels that touch band x's
// lists on XCD x (workgroup b runs on XCD b % 8), so that a list's cache lines are assembled in ONE L2.
constexpr int kBands = 8;
__host__ __device__ inline int band_tiles(int T) { return (T + kBands - 1) / kBands; }

// ---- workspace layouts ------------------------------------------------------
struct GeomWs {
  float* records;          // [P,12]
  float* depths;           // [P]
  uint32_t* rects;         // [P,2]
  uint32_t* tiles_touched; // [P]
  uint32_t* offsets;       // [P] exclusive
  uint32_t* flags;         // [P] bit0..2 colour clamp, bit3 tx clamped, bit4 ty clamped
  uint32_t* block_sums;    // [nblk+1] exclusive scan of per-workgroup instance counts (index order); [nblk] = L
  uint32_t* block_band;    // [kBands][nblk+1] the same per tile band: column b = exclusive scan of the workgroups' counts of
                           // instances whose tile lies in band b; [b][nblk] = the band's total
  float* shjac;            // [P,kJacStride] d(rgb)/d(view direction): 9 values, rows = direction component (hgs_raster_args.prepare_backward)
  unsigned long long* scan_chain;  // [(1 + kBands) * scan_chunks(nblk)] published chunk totals of the K2 scans (K1 clears it)
  static size_t bytes(int32_t P);
  static GeomWs carve_from(void* base, int32_t P);
};

struct BinWs {
  uint32_t* keys_in;   // [L] tile id per emitted instance (index order)
  uint32_t* vals_in;   // [L] Gaussian id
  uint32_t* keys_out;  // [L] tile ids, stable-sorted
  uint32_t* vals_out;  // [L] per tile ascending id after the tile sort; point_list after the per-tile depth sort
  uint32_t* ranges;    // [T,2]
  uint32_t* big_tiles; // [3 + 3T] counters + lists of the tiles too crowded for the one-wave register sort
  uint32_t* tile_order; // [8 * ceil(T/8)] tile of workgroup b: XCD bands, heavy tiles first inside a band (binning.hip)
  void* sort_tmp;
  static size_t bytes(uint32_t L, int32_t T);
  static BinWs carve_from(void* base, uint32_t L, int32_t T);
};

struct ImgWs {
  float* final_T;       // [H*W]
  uint32_t* n_contrib;  // [H*W]
  static size_t bytes(int32_t W, int32_t H);
  static ImgWs carve_from(void* base, int32_t W, int32_t H);
};

inline int grid_x(int W) { return (W + kTile - 1) / kTile; }
inline int grid_y(int H) { return (H + kTile - 1) / kTile; }
inline int tile_bits(int T) {
  int b = 0;
  while ((1 << b) < T) ++b;
  return b < 1 ? 1 : b;
}

// Chunks of the workgroup-sum scans: one 1024-thread workgroup per 8192 entries (scan_chunks(n) workgroups per array).
constexpr int kScanPer = 8;
constexpr int kScanChunk = 1024 * kScanPer;
__host__ __device__ inline int scan_chunks(size_t n) { return (int)((n + kScanChunk - 1) / kScanChunk) + (n == 0 ? 1 : 0); }

#ifdef __HIPCC__
// In-place exclusive scan of sums[0 .. n), total to sums[n], by scan_chunks(n) workgroups of 1024 threads IN ONE LAUNCH:
// workgroup c (= blockIdx.x) scans entries [c, c + 1) * kScanChunk locally, publishes its total in chain[c] (bit 32 = valid)
// and adds the published totals of the workgroups before it -- which were dispatched before it and wait for nobody, so
// the look-back cannot deadlock whatever fits on the chip.  HIP does not PROMISE that dispatch order, so the launchers
// do not lean on it: a grid of at most scan_resident_workgroups() workgroups is resident all at once (every compute unit
// holds one 1024-thread workgroup of this kernel) and then no order matters; larger jobs are split into one launch per
// array and per `resident` chunks of it (c_off: the look-back then reaches into launches that have completed).  `chain` must be ZERO when
// the launch starts (the kernel that produces the sums clears it).  One workgroup walking the array 1024 entries per round cost 0.13 ms at the 96 k
// sums of a 24.7 M-row view and 0.24 ms at the 195 k of a 50 M-node cut (one memory round trip + three barriers per
// round, nothing to overlap them with); the 24 chunk workgroups take one round each.
// Returns the grand total in the LAST chunk's threads (0 elsewhere).
// c / chunks: this workgroup's chunk and the number of chunks (default: blockIdx.x of gridDim.x; a launch of `resident`
// workgroups at a time passes its offset -- the chunks of earlier launches have published their totals long ago).
__device__ __forceinline__ uint32_t chained_scan_inplace(uint32_t* __restrict__ sums, int n,
                                                         unsigned long long* __restrict__ chain, int c_off = 0,
                                                         int chunks_all = 0) {
  __shared__ uint32_t scan_wave_tot[16];
  __shared__ uint32_t scan_prefix;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = (int)blockIdx.x + c_off, chunks = chunks_all ? chunks_all : (int)gridDim.x;
  const int i0 = c * kScanChunk + tid * kScanPer;
  // all loads are issued before the first is waited for (clamped index + select, no branch per load)
  uint32_t v[kScanPer], mine = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) v[k] = sums[max(min(i0 + k, n - 1), 0)];
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    v[k] = (i0 + k < n) ? v[k] : 0u;
    mine += v[k];
  }
  uint32_t inc = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) scan_wave_tot[wave] = inc;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const uint32_t t = scan_wave_tot[w];
    wbase += (w < wave) ? t : 0u;
    total += t;
  }
  if (tid == 0) __hip_atomic_store(&chain[c], (1ull << 32) | (unsigned long long)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (wave == 0) {            // look back: lane l sums the totals of chunks l, l + 64, ... below c
    uint32_t before = 0;
    for (int q = lane; q < c; q += 64) {
      unsigned long long x;
      do { x = __hip_atomic_load(&chain[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); } while ((x >> 32) == 0ull);
      before += (uint32_t)x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    if (lane == 0) scan_prefix = before;
  }
  __syncthreads();
  const uint32_t prefix = scan_prefix;
  uint32_t run = prefix + wbase + inc - mine;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    if (i0 + k < n) sums[i0 + k] = run;
    run += v[k];
  }
  if (c != chunks - 1) return 0u;
  if (tid == 0) sums[n] = prefix + total;
  return prefix + total;
}
#endif

// Host waits of the hot path (the instance count after K1 + scan, the size of an LOD cut): the awaited work is tens of
// microseconds to a few milliseconds away and the GPU idles until the host has reacted, so the host POLLS the
// completion signal instead of sleeping on the interrupt (hipStreamSynchronize / hipEventSynchronize block after a short
// spin; the wake-up was measured at milliseconds on virtualised hosts: 16.5 instead of 10.2 ms per frame of the
// 50 M-node render loop).  After 200 ms of polling, or with HGS_BLOCKING_WAIT set, the blocking call takes over.
hipError_t wait_stream(hipStream_t s);
hipError_t wait_event(hipEvent_t e);

// Compute units of the current device = 1024-thread workgroups of the chained scan that are certainly resident together.
int scan_resident_workgroups();
// HGS_SCAN_SPLIT (diagnostic, read once per process): one launch per array even where the single grid fits.
bool scan_split_forced();

// ---- stage launchers (each returns an HGS_* code) ----------------------------
// super (optional): zeroed superblock totals (super_block_acquire): K1 adds its workgroup sums to them, g.block_sums /
// g.block_band keep the RAW sums, no scan launch follows and K3 builds its prefixes itself (preprocess.hip, binning.hip)
// heavy_thr (with super): workgroups whose instance sum exceeds it are filed into the heavy list (0: none)
int launch_preprocess_fwd(const hgs_raster_args& a, const GeomWs& g, int32_t* radii, hipStream_t s,
                          uint32_t* super = nullptr, uint32_t heavy_thr = 0);
// Superblocks of K1's workgroup sums: kSuper consecutive workgroups; the library-owned block holds (1 + kBands) rows of
// kMaxSuper totals per (device, stream) and one row of the superblocks' largest workgroup sums.  acquire: nullptr = take the scan launch (more than kMaxSuper superblocks is
// the caller's check).  mark_dirty: an error return left totals behind, zero them before the next use.
constexpr int kSuper = 64;
constexpr int kMaxSuper = 1024;
// the block's last row: [0] = length of the HEAVY LIST, then (block, instance sum) pairs of the K1 workgroups whose sum
// exceeds the threshold the host handed to K1 and K3 (binning.hip: k3_heavy_threshold; at most kMaxHeavy are filed)
constexpr int kMaxHeavy = 256;
constexpr int kHeavyRow = (1 + kBands) * kMaxSuper;
static_assert(1 + 2 * kMaxHeavy <= kMaxSuper, "the heavy list fits the row");
uint32_t k3_heavy_threshold(uint32_t L_cap, int32_t P);     // 0: no sharing
uint32_t* super_block_acquire(hipStream_t s);
size_t super_block_bytes();
void super_block_mark_dirty(const uint32_t* words);
int super_block_release(int device);   // frees the blocks of a device (< 0: all); returns how many
// scans block_sums and the kBands columns of block_band (one launch while the grid is resident at once, else one per array)
int launch_scan_block_sums(const GeomWs& g, int32_t P, hipStream_t s, bool debug, uint32_t* total_mirror = nullptr);
// banded: one instance stream per tile band (b.keys_in = band-local tile ids), else one stream of global tile ids
// super: K1's superblock totals (g holds raw workgroup sums, see launch_preprocess_fwd), banded only; the kernel's last
// workgroup then stores the totals and, into *total_mirror (optional, device-visible), the instance count
int launch_duplicate_tiles(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, uint32_t L_cap, bool banded,
                           hipStream_t s, const uint32_t* super = nullptr, uint32_t* total_mirror = nullptr);
// the words of the tile-binning scratch that the banded K3 clears for the counting kernels (tile_bin.hip)
uint32_t* tile_bin_zero_words(void* tmp, uint32_t L_cap, int32_t T);
int tile_bin_zero_count(int32_t T);
int launch_tile_ranges(const BinWs& b, uint32_t L_cap, const uint32_t* L_dev, int32_t T, hipStream_t s, bool debug);
// b.tile_order from the final b.ranges (one small workgroup; counting sort over quantised instance counts)
int launch_tile_order(const BinWs& b, int32_t T, hipStream_t s, bool debug);
// fill_tile_ids: also write every instance's tile id into b.keys_out (the tile-binning path does not produce it)
int launch_tile_depth_sort(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, uint32_t L, int32_t T,
                           bool fill_tile_ids, hipStream_t s);
int launch_render_fwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      float* out_color, float* out_invdepth, hipStream_t s);
int launch_render_bwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      const float* out_color, const float* out_invdepth, const float* dL_dcolor,
                      const float* dL_dinvdepth, float* inst_grads, hipStream_t s);
// dmean_rows / lod_flag: scratch of the in-kernel LOD scatter (hgs_raster_args.lod_scatter): the per-row mean gradient
// K8a hands to K8b, and the "parent indices are not non-decreasing" word (set by launch_lod_monotone)
// L: the frame's instance count (a frame of long runs sums them with kernels of their own in front of K8a; inst_grads is
// consumed: they leave every segment's sums over the segment's first records); work / work_counter: bwd_ws_work*
int launch_preprocess_bwd(const hgs_raster_args& a, const GeomWs& g, const float* inst_grads, float* drgb,
                          float* dmean_rows, const uint32_t* lod_flag, const hgs_raster_grads& out, uint32_t L,
                          uint2* work, uint32_t* work_counter, hipStream_t s);
int launch_lod_monotone(const int32_t* parent_indices, int32_t n, uint32_t* flag, hipStream_t s);
// Per-view device pointers of the batched SH kernels.  Kept small (24 pointers): they are kernel arguments and must
// stay in scalar registers across the view loop.
struct ShBwdViews {
  const void* mask[HGS_MAX_DEFERRED_VIEWS];    // deferred raster backward: uint32 tiles_touched[P] (visibility);
                                               // colour route: uint8 clamp mask[P] of the forward
  const float* drgb[HGS_MAX_DEFERRED_VIEWS];
  const float* campos[HGS_MAX_DEFERRED_VIEWS];
  int n;
  int color;                                   // which of the two meanings `mask` has
};
struct ShFwdViews {
  const float* campos[HGS_MAX_DEFERRED_VIEWS];
  float* rgb_out[HGS_MAX_DEFERRED_VIEWS];
  uint8_t* clamp_out[HGS_MAX_DEFERRED_VIEWS];
  int n;
};
int launch_sh_colors_batched(const ShFwdViews& v, int32_t P, int32_t M, int32_t sh_degree, const float* means3D,
                             const float* shs, hipStream_t s);
int launch_sh_bwd_batched(const ShBwdViews& v, int32_t P, int32_t M, int32_t sh_degree, const float* means3D,
                          const float* shs, float* dL_dshs, float* dL_dmeans3D, bool accumulate, hipStream_t s);
// the per-Gaussian colour gradients sit behind the instance gradients in the backward scratch
inline float* bwd_ws_drgb(void* bwd_ws, uint32_t L) {
  return reinterpret_cast<float*>(static_cast<char*>(bwd_ws) + align_up((size_t)(L ? L : 1) * kInstStride * 4));
}
// behind them: [P,3] mean gradients of the rows and one flag word (in-kernel LOD scatter only)
inline float* bwd_ws_dmean(void* bwd_ws, uint32_t L, int32_t P) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(bwd_ws_drgb(bwd_ws, L)) + align_up((size_t)(P > 0 ? P : 1) * 3 * 4));
}
inline uint32_t* bwd_ws_lod_flag(void* bwd_ws, uint32_t L, int32_t P) {
  return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(bwd_ws_dmean(bwd_ws, L, P)) + align_up((size_t)(P > 0 ? P : 1) * 3 * 4));
}
// behind the flag word (same 256-byte block): the counter of the long-run worklist of K8, and the worklist itself -- one
// (Gaussian, segment) pair per kK8Seg records of every run of more than kK8LongRun records (preprocess.hip)
constexpr uint32_t kK8LongRun = 48;
constexpr uint32_t kK8Seg = 512;
inline uint32_t* bwd_ws_work_counter(void* bwd_ws, uint32_t L, int32_t P) { return bwd_ws_lod_flag(bwd_ws, L, P) + 16; }
inline uint2* bwd_ws_work(void* bwd_ws, uint32_t L, int32_t P) {
  return reinterpret_cast<uint2*>(reinterpret_cast<char*>(bwd_ws_lod_flag(bwd_ws, L, P)) + kAlign);
}
inline size_t bwd_ws_work_items(uint32_t L) { return (size_t)(L ? L : 1) / kK8LongRun + 2; }   // a run of n > 48 records has <= n / 48 segments
inline size_t bwd_ws_bytes(uint32_t L, int32_t P) {
  return align_up((size_t)(L ? L : 1) * kInstStride * 4) + 2 * align_up((size_t)(P > 0 ? P : 1) * 3 * 4) + 2 * kAlign +
         align_up(bwd_ws_work_items(L) * sizeof(uint2));
}
// tile binning without a sort (tile_bin.hip); tmp shares BinWs::sort_tmp
bool tile_bin_supported(int32_t T);
size_t tile_bin_tmp_bytes(uint32_t L, int32_t T);
// keys / vals: the banded streams of launch_duplicate_tiles; band_totals: g.block_band (column stride nblk + 1, totals in
// row nblk)
int launch_tile_bin(const uint32_t* keys, const uint32_t* vals, uint32_t* vals_out, void* tmp, uint32_t L_cap,
                    const uint32_t* band_totals, int32_t nblk, int32_t T, uint32_t* ranges, uint32_t* big,
                    uint32_t* tile_order, uint32_t* super, hipStream_t s, bool debug);
size_t sort_tmp_bytes(uint32_t n);
int sort_pairs(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
               void* tmp, uint32_t n, int end_bit, hipStream_t s, bool debug);
// 32-bit keys; vals_in may be nullptr (values = input positions); n_dev (optional) = actual count on the
// device, n then being a capacity
int sort_pairs32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                 void* tmp, uint32_t n, const uint32_t* n_dev, int end_bit, hipStream_t s, bool debug);

}  // namespace hgs
