// C ABI of libhgs.so (declared in include/hgs.h): argument validation, workspace
// carving, stage sequencing.  No torch types; the Python host (or any FFI) owns memory.
#include "common.h"

#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

namespace hgs {

// ---- optional per-stage timing (hipEvents on the caller's stream) ---------------
enum Stage { ST_PREPROCESS_FWD = 0, ST_SCAN, ST_DUPLICATE, ST_SORT, ST_RANGES, ST_SORT_DEPTH, ST_RENDER_FWD,
             ST_MEMSET_BWD, ST_RENDER_BWD, ST_PREPROCESS_BWD, ST_SH_BATCHED, ST_COUNT };
static const char* kStageNames[ST_COUNT] = {"preprocess_fwd", "scan", "duplicate_keys", "tile_sort", "tile_ranges",
                                            "tile_depth_sort", "render_fwd", "memset_bwd", "render_bwd",
                                            "preprocess_bwd", "sh_bwd_batched"};
struct Pending { int stage; hipEvent_t a, b; };
static uint32_t g_timing = 0;      // bit 0: every stage; bit (1 + stage): that stage only
static std::mutex g_tmu;
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_pool;
static double g_ms[ST_COUNT];
static uint32_t g_calls[ST_COUNT];

struct StageTimer {
  int stage; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; bool on;
  StageTimer(int st, hipStream_t stream) : stage(st), s(stream), on((g_timing & (1u | (2u << st))) != 0) {
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_tmu);
    for (hipEvent_t* e : {&a, &b}) {
      if (!g_pool.empty()) { *e = g_pool.back(); g_pool.pop_back(); }
      else if (hipEventCreate(e) != hipSuccess) { on = false; return; }
    }
    (void)hipEventRecord(a, s);
  }
  ~StageTimer() {
    if (!on) return;
    (void)hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(g_tmu);
    g_pending.push_back({stage, a, b});
  }
};
#define HGS_TIMED(stage, stream, expr) [&]() { hgs::StageTimer _t(stage, stream); return (expr); }()

static thread_local char g_err[512] = "";

// Per calling thread: 4 bytes of pinned, device-mapped host memory (the scan kernel stores the instance count there --
// no copy command in the stream; where the mapping is refused, a D2H copy into it: a copy into pageable memory would
// block the host until it has run, which would defeat enqueue-ahead in hgs_raster_fwd) and ONE reusable event per device.  Both are released by a
// pthread key destructor when the thread ends (autograd worker threads come and go); a thread that is still alive
// at process exit leaves them to the driver's teardown, which is the only safe order.
constexpr int kMaxDevices = 16;
struct ThreadHost {
  uint32_t* pinned_L = nullptr;
  hipEvent_t ev[kMaxDevices] = {};     // one per device: an event is recorded on streams of the device it was created on
};
static pthread_key_t g_host_key;
static pthread_once_t g_host_once = PTHREAD_ONCE_INIT;
static void thread_host_free(void* p) {
  ThreadHost* h = static_cast<ThreadHost*>(p);
  if (!h) return;
  for (int i = 0; i < kMaxDevices; ++i)
    if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  if (h->pinned_L) (void)hipHostFree(h->pinned_L);
  delete h;
}
static void thread_host_key_init() { (void)pthread_key_create(&g_host_key, thread_host_free); }
// call after hipSetDevice(device): the event of that device is created on first use
static ThreadHost* thread_host(int device) {
  if (device < 0 || device >= kMaxDevices) return nullptr;
  (void)pthread_once(&g_host_once, thread_host_key_init);
  ThreadHost* h = static_cast<ThreadHost*>(pthread_getspecific(g_host_key));
  if (!h) {
    h = new ThreadHost();
    if (hipHostMalloc(reinterpret_cast<void**>(&h->pinned_L), sizeof(uint32_t), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
      thread_host_free(h);
      return nullptr;
    }
    (void)pthread_setspecific(g_host_key, h);
  }
  if (!h->ev[device] && hipEventCreateWithFlags(&h->ev[device], hipEventDisableTiming) != hipSuccess) {
    h->ev[device] = nullptr;
    return nullptr;
  }
  return h;
}

static bool blocking_waits() {
  static const bool b = getenv("HGS_BLOCKING_WAIT") != nullptr;
  return b;
}
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
  __asm__ __volatile__("yield");
#else
  sched_yield();
#endif
}
template <typename Query>
static bool poll_done(Query query, hipError_t* result) {
  if (blocking_waits()) return false;
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned i = 0;; ++i) {
    const hipError_t q = query();
    if (q != hipErrorNotReady) { *result = q; return true; }
    (void)hipGetLastError();                       // "not ready" is not an error to keep
    cpu_relax();
    if ((i & 255u) == 255u) {
      timespec t;
      clock_gettime(CLOCK_MONOTONIC, &t);
      if ((t.tv_sec - t0.tv_sec) * 1000000000LL + (t.tv_nsec - t0.tv_nsec) > 200000000LL) return false;
    }
  }
}
hipError_t wait_stream(hipStream_t s) {
  hipError_t r;
  if (poll_done([&] { return hipStreamQuery(s); }, &r)) return r;
  return hipStreamSynchronize(s);
}
hipError_t wait_event(hipEvent_t e) {
  hipError_t r;
  if (poll_done([&] { return hipEventQuery(e); }, &r)) return r;
  return hipEventSynchronize(e);
}

// 1024-thread workgroups of the chained scan that are certainly resident together = compute units of the device -- as far
// as this process can tell: the attribute does not see a CU mask (HSA_CU_MASK, ROC_GLOBAL_CU_MASK), so with one of them
// in the environment the answer is 1 (the scans then run one chunk per launch: slow and correct).  The scan launch itself is the fallback route since round 5 (preprocess.hip: superblock totals).
int scan_resident_workgroups() {
  static std::atomic<int> cus[kMaxDevices];
  static const bool masked = getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK") || getenv("HSA_CU_MASK_SKIP_INIT");
  if (masked) return 1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 1;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 1;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
bool scan_split_forced() {
  static const bool b = getenv("HGS_SCAN_SPLIT") != nullptr;
  return b;
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

size_t GeomWs::bytes(int32_t P) {
  const size_t p = (size_t)(P > 0 ? P : 1);
  const size_t nblk = (p + kPreBlock - 1) / kPreBlock;
  return align_up(p * kRecFloats * 4) + align_up(p * 4) + align_up(p * 8) + 3 * align_up(p * 4) +
         align_up((nblk + 1) * 4) + align_up((nblk + 1) * kBands * 4) + align_up(p * kJacStride * 4) +
         align_up((size_t)(1 + kBands) * scan_chunks(nblk) * 8) + kAlign;
}
GeomWs GeomWs::carve_from(void* base, int32_t P) {
  const size_t p = (size_t)(P > 0 ? P : 1);
  const size_t nblk = (p + kPreBlock - 1) / kPreBlock;
  char* c = static_cast<char*>(base);
  GeomWs g;
  g.records = carve<float>(c, p * kRecFloats);
  g.depths = carve<float>(c, p);
  g.rects = carve<uint32_t>(c, p * 2);
  g.tiles_touched = carve<uint32_t>(c, p);
  g.offsets = carve<uint32_t>(c, p);
  g.flags = carve<uint32_t>(c, p);
  g.block_sums = carve<uint32_t>(c, nblk + 1);
  g.block_band = carve<uint32_t>(c, (nblk + 1) * kBands);
  g.shjac = carve<float>(c, p * kJacStride);
  g.scan_chain = carve<unsigned long long>(c, (size_t)(1 + kBands) * scan_chunks(nblk));
  return g;
}

size_t BinWs::bytes(uint32_t L, int32_t T) {
  const size_t l = L ? L : 1;
  const size_t tmp_sort = sort_tmp_bytes(L ? L : 1), tmp_bin = tile_bin_tmp_bytes(L, T);
  return 4 * align_up(l * 4) + align_up((size_t)T * 8) + align_up(((size_t)T * 3 + 3) * 4) + align_up(((size_t)T + 8) * 4) +
         (tmp_sort > tmp_bin ? tmp_sort : tmp_bin) + kAlign;
}
BinWs BinWs::carve_from(void* base, uint32_t L, int32_t T) {
  const size_t l = L ? L : 1;
  char* c = static_cast<char*>(base);
  BinWs b;
  b.keys_in = carve<uint32_t>(c, l);
  b.vals_in = carve<uint32_t>(c, l);
  b.keys_out = carve<uint32_t>(c, l);
  b.vals_out = carve<uint32_t>(c, l);
  b.ranges = carve<uint32_t>(c, (size_t)T * 2);
  b.big_tiles = carve<uint32_t>(c, (size_t)T * 3 + 3);
  b.tile_order = carve<uint32_t>(c, (size_t)T + 8);    // 8 * ceil(T / 8) entries
  b.sort_tmp = c;
  return b;
}

size_t ImgWs::bytes(int32_t W, int32_t H) { return 2 * align_up((size_t)W * H * 4) + kAlign; }
ImgWs ImgWs::carve_from(void* base, int32_t W, int32_t H) {
  char* c = static_cast<char*>(base);
  ImgWs im;
  im.final_T = carve<float>(c, (size_t)W * H);
  im.n_contrib = carve<uint32_t>(c, (size_t)W * H);
  return im;
}

static int validate(const hgs_raster_args* a) {
  if (!a) { set_error("null args"); return HGS_ERR_INVALID; }
  if (a->P < 0 || a->width <= 0 || a->height <= 0) { set_error("bad sizes P=%d W=%d H=%d", a->P, a->width, a->height); return HGS_ERR_INVALID; }
  if (grid_x(a->width) > 1023 || grid_y(a->height) > 1023) { set_error("image larger than 16368 px per side is not supported"); return HGS_ERR_INVALID; }
  if (!a->bg || !a->viewmatrix || !a->projmatrix || !a->campos) { set_error("bg/viewmatrix/projmatrix/campos must be device pointers"); return HGS_ERR_INVALID; }
  if (a->P > 0) {
    if (!a->means3D || !a->opacities) { set_error("means3D/opacities missing"); return HGS_ERR_INVALID; }
    if ((a->shs != nullptr) == (a->colors_precomp != nullptr)) { set_error("provide exactly one of shs / colors_precomp"); return HGS_ERR_INVALID; }
    const bool sr = a->scales && a->rotations;
    if (sr == (a->cov3D_precomp != nullptr) || ((a->scales != nullptr) != (a->rotations != nullptr))) {
      set_error("provide exactly one of (scales, rotations) / cov3D_precomp");
      return HGS_ERR_INVALID;
    }
    if (a->shs) {
      if (a->sh_degree < 0 || a->sh_degree > 3) { set_error("sh_degree %d not in 0..3", a->sh_degree); return HGS_ERR_INVALID; }
      if (a->M < (a->sh_degree + 1) * (a->sh_degree + 1) || a->M > 16) { set_error("M=%d incompatible with sh_degree=%d (max 16 coefficients)", a->M, a->sh_degree); return HGS_ERR_INVALID; }
    }
    // the SH blocks and the quaternions are moved with 16-byte accesses
    if (((uintptr_t)a->shs | (uintptr_t)a->shs_rest | (uintptr_t)a->rotations) & 15u) { set_error("shs / shs_rest / rotations must be 16-byte aligned"); return HGS_ERR_INVALID; }
    if (a->defer_sh_bwd && a->shs_rest) { set_error("defer_sh_bwd is not available with split SH storage (shs_rest)"); return HGS_ERR_INVALID; }
    if (a->shs_rest && (!a->shs || a->M < 2)) { set_error("shs_rest needs shs (features_dc) and M >= 2"); return HGS_ERR_INVALID; }
    if ((a->activations & HGS_ACT_OPACITY_SIGMOID) && (a->activations & HGS_ACT_OPACITY_ABS)) { set_error("choose one opacity activation"); return HGS_ERR_INVALID; }
    if ((a->activations & (HGS_ACT_SCALE_EXP | HGS_ACT_ROT_NORMALIZE)) && a->cov3D_precomp) { set_error("scale / rotation activations need scales and rotations"); return HGS_ERR_INVALID; }
    if ((a->interpolation_weights != nullptr) != (a->num_node_kids != nullptr)) { set_error("interpolation_weights and num_node_kids must be given together"); return HGS_ERR_INVALID; }
    if ((a->lod_render_indices != nullptr) != (a->lod_parent_indices != nullptr)) { set_error("lod_render_indices and lod_parent_indices must be given together"); return HGS_ERR_INVALID; }
    if (a->lod_render_indices) {
      if (!a->shs || a->shs_rest || !a->scales || !a->rotations || a->activations || !a->interpolation_weights) { set_error("in-kernel LOD interpolation needs shs, scales, rotations, interpolation_weights / num_node_kids and no activations"); return HGS_ERR_INVALID; }
      if (a->lod_n < 0 || a->lod_n > a->P || a->lod_rows < a->P - a->lod_n) { set_error("bad lod_n / lod_rows (P=%d lod_n=%d lod_rows=%d)", a->P, a->lod_n, a->lod_rows); return HGS_ERR_INVALID; }
      if (a->defer_sh_bwd || a->accumulate_grads) { set_error("defer_sh_bwd / accumulate_grads are not available with in-kernel LOD interpolation"); return HGS_ERR_INVALID; }
    }
  }
  return HGS_OK;
}

}  // namespace hgs

using namespace hgs;

extern "C" {

int hgs_abi_version(void) { return HGS_ABI_VERSION; }
const char* hgs_last_error(void) { return g_err; }
int hgs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

int hgs_raster_ws_sizes(int32_t P, int32_t width, int32_t height, uint32_t L, size_t* geom_bytes,
                        size_t* bin_bytes, size_t* img_bytes, size_t* bwd_bytes) {
  if (P < 0 || width <= 0 || height <= 0) { set_error("bad sizes"); return HGS_ERR_INVALID; }
  const int T = grid_x(width) * grid_y(height);
  if (geom_bytes) *geom_bytes = GeomWs::bytes(P);
  if (bin_bytes) *bin_bytes = BinWs::bytes(L, T);
  if (img_bytes) *img_bytes = ImgWs::bytes(width, height);
  if (bwd_bytes) *bwd_bytes = bwd_ws_bytes(L, P);
  return HGS_OK;
}

int hgs_raster_fwd_stage1(const hgs_raster_args* a, void* geom_ws, int32_t* radii, uint32_t* L_out_host,
                          hgs_stream_t stream, int device) {
  int rc = validate(a);
  if (rc) return rc;
  if (!geom_ws || !L_out_host || (a->P > 0 && !radii)) { set_error("null workspace/output"); return HGS_ERR_INVALID; }
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const GeomWs g = GeomWs::carve_from(geom_ws, a->P);
  *L_out_host = 0;
  if (a->P == 0) return HGS_OK;
  if ((rc = HGS_TIMED(ST_PREPROCESS_FWD, s, launch_preprocess_fwd(*a, g, radii, s)))) return rc;
  if ((rc = HGS_TIMED(ST_SCAN, s, launch_scan_block_sums(g, a->P, s, a->debug)))) return rc;
  const int nblk = (a->P + kPreBlock - 1) / kPreBlock;
  HGS_HIP(hipMemcpyAsync(L_out_host, g.block_sums + nblk, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HGS_HIP(wait_stream(s));
  return HGS_OK;
}

// Stage-2 launches.  L is exact when L_dev == nullptr; otherwise it is a capacity and the kernels read the
// actual instance count from device memory.
// super != nullptr (single-call forward only): K1 left raw workgroup sums + superblock totals, K3 finishes the scans and
// is the kernel that produces the instance count -- into *mirror (mapped host word) or, without a mapping, by a copy
// into `stage` -- and `ev` is recorded right behind it.
static int enqueue_stage2(const hgs_raster_args* a, const GeomWs& g, const BinWs& b, const ImgWs& im, uint32_t L,
                          const uint32_t* L_dev, int T, float* out_color, float* out_invdepth, hipStream_t s,
                          uint32_t* super = nullptr, uint32_t* mirror = nullptr, uint32_t* stage = nullptr,
                          hipEvent_t ev = nullptr) {
  int rc;
  const bool bin = L > 0 && tile_bin_supported(T);
  if ((rc = HGS_TIMED(ST_DUPLICATE, s, launch_duplicate_tiles(*a, g, b, L, bin, s, super, mirror)))) return rc;   // also zeroes b.ranges
  if (super) {
    if (!mirror) HGS_HIP(hipMemcpyAsync(stage, L_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HGS_HIP(hipEventRecord(ev, s));
  }
  if (bin) {            // counting pass + scatter pass per tile band; writes the tile ranges too
    const int nblk = (a->P + kPreBlock - 1) / kPreBlock;
    if ((rc = HGS_TIMED(ST_SORT, s, launch_tile_bin(b.keys_in, b.vals_in, b.vals_out, b.sort_tmp, L, g.block_band, nblk, T, b.ranges, b.big_tiles, b.tile_order, super, s, a->debug)))) return rc;
  } else {              // very large tile grids: stable radix sort by tile id, then ranges off the sorted ids
    if (L > 0) {
      if ((rc = HGS_TIMED(ST_SORT, s, sort_pairs32(b.keys_in, b.vals_in, b.keys_out, b.vals_out, b.sort_tmp, L, L_dev, tile_bits(T), s, a->debug)))) return rc;
    }
    if ((rc = HGS_TIMED(ST_RANGES, s, launch_tile_ranges(b, L, L_dev, T, s, a->debug)))) return rc;
  }
  // (the sorted tile-id column is introspection: the binning path only writes it for a debug forward)
  if ((rc = HGS_TIMED(ST_SORT_DEPTH, s, launch_tile_depth_sort(*a, g, b, L, T, bin && a->debug, s)))) return rc;
  if (!bin && (rc = launch_tile_order(b, T, s, a->debug))) return rc;   // (the binning path orders inside its scatter launch)
  return HGS_TIMED(ST_RENDER_FWD, s, launch_render_fwd(*a, g, b, im, out_color, out_invdepth, s));
}

int hgs_raster_fwd_stage2(const hgs_raster_args* a, void* geom_ws, void* bin_ws, void* img_ws, uint32_t L,
                          float* out_color, float* out_invdepth, hgs_stream_t stream, int device) {
  int rc = validate(a);
  if (rc) return rc;
  if (!geom_ws || !bin_ws || !img_ws || !out_color) { set_error("null workspace/output"); return HGS_ERR_INVALID; }
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int T = grid_x(a->width) * grid_y(a->height);
  const GeomWs g = GeomWs::carve_from(geom_ws, a->P);
  const BinWs b = BinWs::carve_from(bin_ws, L, T);
  const ImgWs im = ImgWs::carve_from(img_ws, a->width, a->height);
  return enqueue_stage2(a, g, b, im, L, nullptr, T, out_color, out_invdepth, s);
}

int hgs_raster_fwd(const hgs_raster_args* a, void* geom_ws, void* bin_ws, void* img_ws, uint32_t L_cap,
                   int32_t* radii, float* out_color, float* out_invdepth, uint32_t* L_out_host,
                   hgs_stream_t stream, int device) {
  int rc = validate(a);
  if (rc) return rc;
  if (!geom_ws || !bin_ws || !img_ws || !out_color || !L_out_host || (a->P > 0 && !radii)) {
    set_error("null workspace/output");
    return HGS_ERR_INVALID;
  }
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int T = grid_x(a->width) * grid_y(a->height);
  const GeomWs g = GeomWs::carve_from(geom_ws, a->P);
  const BinWs b = BinWs::carve_from(bin_ws, L_cap, T);
  const ImgWs im = ImgWs::carve_from(img_ws, a->width, a->height);
  *L_out_host = 0;
  if (a->P == 0) return enqueue_stage2(a, g, b, im, 0, nullptr, T, out_color, out_invdepth, s);
  ThreadHost* th = thread_host(device);
  if (!th) { set_error("cannot allocate pinned host memory / event (device %d)", device); return HGS_ERR_NOMEM; }
  hipEvent_t ev = th->ev[device];
  uint32_t* stage = th->pinned_L;
  // the scan kernel stores the count into the mapped host word itself (visible to the host once the event after the
  // kernel has fired: a kernel's writes are released to system scope when it ends)
  static const bool by_copy = getenv("HGS_COUNT_BY_COPY") != nullptr;     // diagnostic: the copy command instead
  void* mirror = nullptr;
  if (by_copy || hipHostGetDevicePointer(&mirror, stage, 0) != hipSuccess) { (void)hipGetLastError(); mirror = nullptr; }
  // No scan launch on the usual path: K1 adds its workgroup sums to zeroed superblock totals and K3 finishes the scans
  // (preprocess.hip, binning.hip).  The scan launch remains for tile grids that take the radix path, for more than
  // kSuper * kMaxSuper workgroups (16.7 M rows) and when no zeroed block is to be had (HGS_SCAN_LAUNCH=1 forces it).
  const int nblk = (a->P + kPreBlock - 1) / kPreBlock;
  uint32_t* super = nullptr;
  if (L_cap > 0 && tile_bin_supported(T) && nblk <= kSuper * kMaxSuper) super = super_block_acquire(s);
  if ((rc = HGS_TIMED(ST_PREPROCESS_FWD, s, launch_preprocess_fwd(*a, g, radii, s, super, super ? k3_heavy_threshold(L_cap, a->P) : 0u)))) {
    if (super) super_block_mark_dirty(super);
    return rc;
  }
  const uint32_t* L_dev = g.block_sums + nblk;
  hipError_t e = hipSuccess;
  if (!super) {
    if ((rc = HGS_TIMED(ST_SCAN, s, launch_scan_block_sums(g, a->P, s, a->debug, static_cast<uint32_t*>(mirror))))) return rc;
    if (!mirror) HGS_HIP(hipMemcpyAsync(stage, L_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    e = hipEventRecord(ev, s);
  }
  // everything else is enqueued before the host looks at L: the GPU never waits for the host
  if (e == hipSuccess) rc = enqueue_stage2(a, g, b, im, L_cap, L_dev, T, out_color, out_invdepth, s, super,
                                           static_cast<uint32_t*>(mirror), stage, ev);
  if (rc || e != hipSuccess) {
    // the superblock totals may not have been consumed and cleared: zero them before the next use
    if (super) super_block_mark_dirty(super);
    if (rc) return rc;
  }
  if (e == hipSuccess) e = wait_event(ev);
  if (e != hipSuccess) { set_error("hgs_raster_fwd: %s", hipGetErrorString(e)); return HGS_ERR_HIP; }
  *L_out_host = *stage;
  if (rc) return rc;
  if (*L_out_host > L_cap) {
    if (super) {
      // the caller finishes on hgs_raster_fwd_stage2 over THIS geometry workspace, whose K3 expects the workgroup sums
      // scanned: scan the raw sums now (the superblock totals are consumed and cleared; this path is the rare one)
      HGS_HIP(hipMemsetAsync(g.scan_chain, 0, (size_t)(1 + kBands) * scan_chunks(nblk) * sizeof(unsigned long long), s));
      if ((rc = launch_scan_block_sums(g, a->P, s, a->debug, nullptr))) return rc;
    }
    set_error("instance count %u exceeds the capacity %u given to hgs_raster_fwd", *L_out_host, L_cap);
    return HGS_ERR_CAPACITY;
  }
  return HGS_OK;
}

int hgs_release_device_state(int device) { return super_block_release(device); }

int hgs_raster_bwd(const hgs_raster_args* a, const void* geom_ws, const void* bin_ws, const void* img_ws,
                   void* bwd_ws, uint32_t L, const float* out_color, const float* out_invdepth,
                   const float* dL_dcolor, const float* dL_dinvdepth, const hgs_raster_grads* grads,
                   hgs_stream_t stream, int device) {
  int rc = validate(a);
  if (rc) return rc;
  if (!geom_ws || !bin_ws || !img_ws || !bwd_ws || !out_color || !dL_dcolor || !grads) { set_error("null workspace/input"); return HGS_ERR_INVALID; }
  if (a->lod_render_indices && !a->prepare_backward) { set_error("the backward of an in-kernel LOD interpolation needs prepare_backward = 1 in the forward and the backward call"); return HGS_ERR_INVALID; }
  if (a->lod_scatter && (!a->lod_render_indices || ((a->M * 3) & 3) != 0 || a->accumulate_grads || a->defer_sh_bwd)) { set_error("lod_scatter needs in-kernel LOD interpolation, 3M %% 4 == 0, no accumulation and no deferred SH backward"); return HGS_ERR_INVALID; }
  if (a->P > 0) {
    if (!grads->dL_dmeans3D || !grads->dL_dmeans2D || !grads->dL_dopacity) { set_error("missing gradient outputs"); return HGS_ERR_INVALID; }
    if ((a->shs && !grads->dL_dshs) || (a->shs_rest && !grads->dL_dshs_rest) || (a->colors_precomp && !grads->dL_dcolors) ||
        (a->scales && (!grads->dL_dscales || !grads->dL_drotations)) || (a->cov3D_precomp && !grads->dL_dcov3D)) {
      set_error("gradient outputs do not match the inputs that were provided");
      return HGS_ERR_INVALID;
    }
  }
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->P == 0) return HGS_OK;
  const int T = grid_x(a->width) * grid_y(a->height);
  const GeomWs g = GeomWs::carve_from(const_cast<void*>(geom_ws), a->P);
  const BinWs b = BinWs::carve_from(const_cast<void*>(bin_ws), L, T);
  const ImgWs im = ImgWs::carve_from(const_cast<void*>(img_ws), a->width, a->height);
  float* inst = static_cast<float*>(bwd_ws);
  float* drgb = bwd_ws_drgb(bwd_ws, L);
  if (L > 0) {     // (the instance scratch needs no clearing: K7 writes every record, sums or zeros)
    if ((rc = HGS_TIMED(ST_RENDER_BWD, s, launch_render_bwd(*a, g, b, im, out_color, out_invdepth, dL_dcolor, dL_dinvdepth, inst, s)))) return rc;
  }
  hgs_raster_grads gr = *grads;
  if (!a->shs) gr.dL_dshs = nullptr;
  if (!a->shs_rest) gr.dL_dshs_rest = nullptr;
  if (!a->colors_precomp) gr.dL_dcolors = nullptr;
  if (!a->scales) { gr.dL_dscales = nullptr; gr.dL_drotations = nullptr; }
  if (!a->cov3D_precomp) gr.dL_dcov3D = nullptr;
  uint32_t* lod_flag = nullptr;
  if (a->lod_render_indices && a->lod_scatter) {
    lod_flag = bwd_ws_lod_flag(bwd_ws, L, a->P);
    HGS_HIP(hipMemsetAsync(lod_flag, 0, sizeof(uint32_t), s));
    if ((rc = launch_lod_monotone(a->lod_parent_indices, a->lod_n, lod_flag, s))) return rc;
  }
  return HGS_TIMED(ST_PREPROCESS_BWD, s, launch_preprocess_bwd(*a, g, inst, drgb, bwd_ws_dmean(bwd_ws, L, a->P), lod_flag, gr, L,
                                                             bwd_ws_work(bwd_ws, L, a->P), bwd_ws_work_counter(bwd_ws, L, a->P), s));
}

int hgs_raster_sh_bwd_batched(const hgs_sh_bwd_view* views, int32_t n_views, int32_t P, int32_t M, int32_t sh_degree,
                              const float* means3D, const float* shs, float* dL_dshs, float* dL_dmeans3D,
                              int32_t accumulate, hgs_stream_t stream, int device) {
  if (n_views <= 0 || P <= 0) return HGS_OK;
  if (!views || n_views > HGS_MAX_DEFERRED_VIEWS) { set_error("1..%d deferred views per call", HGS_MAX_DEFERRED_VIEWS); return HGS_ERR_INVALID; }
  if (!means3D || !shs || !dL_dshs || !dL_dmeans3D) { set_error("null argument"); return HGS_ERR_INVALID; }
  if (sh_degree < 0 || sh_degree > 3 || M < (sh_degree + 1) * (sh_degree + 1) || M > 16) { set_error("M=%d incompatible with sh_degree=%d", M, sh_degree); return HGS_ERR_INVALID; }
  if (((uintptr_t)shs | (uintptr_t)dL_dshs) & 15u) { set_error("shs / dL_dshs must be 16-byte aligned"); return HGS_ERR_INVALID; }
  ShBwdViews v;
  v.n = n_views;
  v.color = 0;
  for (int i = 0; i < HGS_MAX_DEFERRED_VIEWS; ++i) {
    const hgs_sh_bwd_view& w = views[i < n_views ? i : 0];
    if (!w.geom_ws || !w.bwd_ws || !w.campos) { set_error("deferred view %d has a null pointer", i); return HGS_ERR_INVALID; }
    v.mask[i] = GeomWs::carve_from(const_cast<void*>(w.geom_ws), P).tiles_touched;
    v.drgb[i] = bwd_ws_drgb(const_cast<void*>(w.bwd_ws), w.L);
    v.campos[i] = w.campos;
  }
  HGS_HIP(hipSetDevice(device));
  return HGS_TIMED(ST_SH_BATCHED, static_cast<hipStream_t>(stream),
                   launch_sh_bwd_batched(v, P, M, sh_degree, means3D, shs, dL_dshs, dL_dmeans3D, accumulate != 0,
                                         static_cast<hipStream_t>(stream)));
}

static int sh_color_check(const hgs_sh_color_view* views, int32_t n_views, int32_t M, int32_t sh_degree,
                          const float* means3D, const float* shs, bool backward) {
  if (!views || n_views > HGS_MAX_DEFERRED_VIEWS) { set_error("1..%d views per call", HGS_MAX_DEFERRED_VIEWS); return HGS_ERR_INVALID; }
  if (!means3D || !shs) { set_error("null argument"); return HGS_ERR_INVALID; }
  if (sh_degree < 0 || sh_degree > 3 || M < (sh_degree + 1) * (sh_degree + 1) || M > 16) { set_error("M=%d incompatible with sh_degree=%d", M, sh_degree); return HGS_ERR_INVALID; }
  if ((uintptr_t)shs & 15u) { set_error("shs must be 16-byte aligned"); return HGS_ERR_INVALID; }
  for (int i = 0; i < n_views; ++i)
    if (!views[i].campos || !views[i].clamp || (backward ? !views[i].d_rgb : !views[i].rgb)) { set_error("colour view %d has a null pointer", i); return HGS_ERR_INVALID; }
  return HGS_OK;
}

int hgs_sh_colors_batched(const hgs_sh_color_view* views, int32_t n_views, int32_t P, int32_t M, int32_t sh_degree,
                          const float* means3D, const float* shs, hgs_stream_t stream, int device) {
  if (n_views <= 0 || P <= 0) return HGS_OK;
  int rc = sh_color_check(views, n_views, M, sh_degree, means3D, shs, false);
  if (rc) return rc;
  ShFwdViews v;
  v.n = n_views;
  for (int i = 0; i < HGS_MAX_DEFERRED_VIEWS; ++i) {
    const hgs_sh_color_view& w = views[i < n_views ? i : 0];
    v.campos[i] = w.campos;
    v.rgb_out[i] = w.rgb;
    v.clamp_out[i] = w.clamp;
  }
  HGS_HIP(hipSetDevice(device));
  return launch_sh_colors_batched(v, P, M, sh_degree, means3D, shs, static_cast<hipStream_t>(stream));
}

int hgs_sh_colors_batched_bwd(const hgs_sh_color_view* views, int32_t n_views, int32_t P, int32_t M, int32_t sh_degree,
                              const float* means3D, const float* shs, float* dL_dshs, float* dL_dmeans3D,
                              int32_t accumulate, hgs_stream_t stream, int device) {
  if (n_views <= 0 || P <= 0) return HGS_OK;
  if (!dL_dshs || !dL_dmeans3D || ((uintptr_t)dL_dshs & 15u)) { set_error("dL_dshs (16-byte aligned) / dL_dmeans3D missing"); return HGS_ERR_INVALID; }
  int rc = sh_color_check(views, n_views, M, sh_degree, means3D, shs, true);
  if (rc) return rc;
  ShBwdViews v;
  v.n = n_views;
  v.color = 1;
  for (int i = 0; i < HGS_MAX_DEFERRED_VIEWS; ++i) {
    const hgs_sh_color_view& w = views[i < n_views ? i : 0];
    v.mask[i] = w.clamp;
    v.drgb[i] = w.d_rgb;
    v.campos[i] = w.campos;
  }
  HGS_HIP(hipSetDevice(device));
  return HGS_TIMED(ST_SH_BATCHED, static_cast<hipStream_t>(stream),
                   launch_sh_bwd_batched(v, P, M, sh_degree, means3D, shs, dL_dshs, dL_dmeans3D, accumulate != 0,
                                         static_cast<hipStream_t>(stream)));
}

int hgs_raster_views_get(int32_t P, int32_t width, int32_t height, uint32_t L, const void* geom_ws,
                         const void* bin_ws, const void* img_ws, hgs_raster_views* out) {
  if (!out || !geom_ws || !bin_ws || !img_ws) { set_error("null argument"); return HGS_ERR_INVALID; }
  const int T = grid_x(width) * grid_y(height);
  const GeomWs g = GeomWs::carve_from(const_cast<void*>(geom_ws), P);
  const BinWs b = BinWs::carve_from(const_cast<void*>(bin_ws), L, T);
  const ImgWs im = ImgWs::carve_from(const_cast<void*>(img_ws), width, height);
  out->tile_ids_sorted = b.keys_out;
  out->point_list = b.vals_out;
  out->ranges = b.ranges;
  out->tiles_touched = g.tiles_touched;
  out->offsets = g.offsets;
  out->depths = g.depths;
  out->rects = g.rects;
  out->records = g.records;
  out->final_T = im.final_T;
  out->n_contrib = im.n_contrib;
  return HGS_OK;
}

int hgs_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_tmu);
  g_timing = (uint32_t)on;
  return HGS_OK;
}
int hgs_timing_stage_count(void) { return ST_COUNT; }
const char* hgs_timing_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }
int hgs_timing_read(double* ms_out, uint32_t* calls_out, int reset) {
  std::lock_guard<std::mutex> lk(g_tmu);
  for (const Pending& p : g_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      g_ms[p.stage] += ms;
      g_calls[p.stage] += 1;
    }
    g_pool.push_back(p.a);
    g_pool.push_back(p.b);
  }
  g_pending.clear();
  for (int i = 0; i < ST_COUNT; ++i) {
    if (ms_out) ms_out[i] = g_ms[i];
    if (calls_out) calls_out[i] = g_calls[i];
    if (reset) { g_ms[i] = 0.0; g_calls[i] = 0; }
  }
  return HGS_OK;
}

size_t hgs_sort_tmp_bytes(uint32_t n) { return sort_tmp_bytes(n ? n : 1); }

int hgs_sort_pairs(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                   void* tmp, uint32_t n, int end_bit, hgs_stream_t stream, int device) {
  if (n && (!keys_in || !vals_in || !keys_out || !vals_out || !tmp)) { set_error("null argument"); return HGS_ERR_INVALID; }
  HGS_HIP(hipSetDevice(device));
  return sort_pairs(keys_in, vals_in, keys_out, vals_out, tmp, n, end_bit, static_cast<hipStream_t>(stream), false);
}

}  // extern "C"
