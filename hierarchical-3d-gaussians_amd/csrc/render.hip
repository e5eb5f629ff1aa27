// K6 (tile compositing, forward) and K7 (its backward).
//
// Restates per-tile front-to-back alpha compositing with an inverse-depth channel
// (op call sites gaussian_renderer/__init__.py:105-113,269-277;
// backward entered from train_single.py:123 / train_post.py:142).
//
// CDNA4 design (not the CUDA 256-threads-one-pixel-each shape):
//  * ONE WAVE RENDERS A WHOLE 16x16 TILE and every lane owns four pixels -- no workgroup barrier, no inter-wave
//    accumulation.
//  * QUADRANT STREAMS (round 3).  A tile instance of the benchmark scene touches ~25 of the tile's 256 pixels; walking the
//    tile's list wave-uniformly (rounds 1-2: every visited instance costs a pass over a 16x8 half = 128 pixel slots)
//    leaves four lanes in five idle.  Here the four DPP ROWS of the wave (16 lanes each) own the four 8x8 QUADRANTS of
//    the tile and each row walks ITS OWN list: the lane that stages an instance of the 64-instance batch tests K1's
//    alpha >= 1/255 box against the four quadrants, four ballots give four 64-bit masks, and every set bit is filed --
//    by its rank inside its mask -- into that quadrant's list in LDS.  An iteration of the inner loop then serves up
//    to FOUR DIFFERENT (instance, quadrant) pairs at once: every row fetches the record of its own next instance
//    (ds_read with a row-dependent address) and the same instruction stream composites it into the row's 64 pixels.
//    Nothing is wave-uniform any more, so nothing diverges; a batch costs max(list lengths) iterations instead of one
//    pass per (instance, half) -- 0.40 iterations per former half-visit on the benchmark scene, 0.43 on the heavy one
//    (tests/tools/quad_stats.py).
//  * a lane's four pixels share x (rows y0, y0+2, y0+4, y0+6 of its quadrant) and go through the instruction stream as
//    two PACKED pairs (float2): compares, selects, min / max, exp and rcp have no packed form and are halved by that.
//  * alpha >= 1/255  <=>  power2 >= log2(1/255) - log2(opacity): the backward's per-lane candidate predicate (log domain,
//    before the exp).  The exact alpha test still takes every borderline decision.
//  * pixel coordinates are TILE-RELATIVE: the record carries the pixel centre as hi + lo floats from K1's
//    double-precision projection; (hi - tile_origin) + lo is exact to ~1e-6 px at any resolution.
//  * a finished / outside pixel is moved to y = 1e18 (never a candidate again): "done" needs no flag.
//  * backward: BACK-TO-FRONT from the forward's last contributor with
//        T_i = T_{i+1} / (1 - alpha_i),   A_i = alpha_{i+1} q_{i+1} + (1 - alpha_{i+1}) A_{i+1},
//        dL/dalpha_i = (q_i - A_i) T_i - T_final (dL/dC . bg) / (1 - alpha_i),   q = dL/dC . c + dL/dD / z
//    (every term relatively accurate; a front-to-back variant using V - prefix was measured 10-80x
//    less accurate on pixels with capped alphas and was dropped).
//  * the ten per-(instance, quadrant) sums are reduced INSIDE THE ROW -- the natural domain of DPP: 21 bank-masked DPP adds
//    reduce all ten values of all four rows (four different instances) at once (rounds 1-2: ~30 instructions of
//    permlane swaps + DPP per instance) -- and STORED to the pair's own slot in LDS (round 6; rounds 3-5 added them to
//    the instance's row with ds_add_f32, and the LDS float adds turned out to be the single most expensive thing of a
//    visit: ~12 cycles per adding lane, profiles/r06_notes.md).  The staging lanes hand out the slots: a lane's pairs take
//    consecutive slots behind those of the lanes below it (the four ballots' ranks are a prefix sum for free); a batch
//    whose pairs outnumber the slots keeps its back-most instances and leaves the rest to the next one.  At the end of a
//    batch lane i adds up instance i's slots in quadrant order and stores the record to its EMISSION slot (40 bytes);
//    instances the tile never reaches store zeros, so the scratch needs no zero fill by anybody.  K8 sums each Gaussian's
//    contiguous run.  No atomics anywhere; bit-reproducible.
//  * the record of the NEXT iteration is requested before the current one is composited (two iterations per trip of the
//    loop): a row's record address depends on a list entry that is itself in LDS, and at 4 waves per SIMD two dependent
//    LDS round trips per iteration are not hidden by the other waves (K7 0.426 -> 0.385 ms; with the pixels' colour
//    gradients in registers instead of LDS 0.339).
#include <stdlib.h>

#include "common.h"

namespace hgs {
namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTEps = 0.0001f;
constexpr float kBig = 1.0e18f;   // y coordinate of a finished / outside pixel: its power is -inf-ish, never a candidate
// instances staged per batch (at most one per lane); row kBatch of the LDS arrays is a dummy instance that can never be
// a candidate (rows whose list is exhausted fetch it)
constexpr int kFwdBatch = 64;
constexpr int kBwdBatch = 64;

struct TileGeom {
  int tile, tx, ty;
};

// XCD-aware block -> tile map: consecutive blocks land on different XCDs (b % 8), so give
// every XCD a contiguous band of tiles; neighbouring tiles share Gaussians and therefore
// share that XCD's L2.
//
// With `order` (tile_order_kernel, binning.hip): the same bands, but inside its band every XCD takes the tiles by
// descending instance count (heavy tiles first, light tiles fill the tail of the launch).
__device__ __forceinline__ bool block_to_tile(int T, int gx, const uint32_t* __restrict__ order, TileGeom& tg) {
  const int per = (T + 7) >> 3;
  const int b = blockIdx.x;
  const int tile = order ? (int)order[b] : (b & 7) * per + (b >> 3);
  if ((uint32_t)tile >= (uint32_t)T) return false;
  tg.tile = tile;
  tg.ty = tile / gx;
  tg.tx = tile - tg.ty * gx;
  return true;
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }

// exp2(min(x, 0)) = min(exp2(x), 1) as ONE instruction: the [0, 1] output clamp of v_exp_f32 (the compiler folds the
// median into the instruction's clamp bit).  Also turns +inf / NaN inputs into 1 / 0.
__device__ __forceinline__ float exp2_le1(float x) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x), 0.0f, 1.0f); }

// The lists' entries name a staged instance by the BYTE OFFSET of its three (four) float4 in `lrec` (no multiply per
// visit): K6's entries are that offset, K7's carry it in their low half.
__device__ __forceinline__ const float4* record_at(const float4* lrec, uint32_t off) {
  return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lrec) + off);
}
__device__ __forceinline__ const float4* entry_record(const float4* lrec, uint32_t e) { return record_at(lrec, e & 0xffffu); }

// ---- per-PIXEL LOD remap (hgs_raster_args.lod_per_pixel; oracle: raster_oracle.lod_alpha) -------------------------------
// alpha' = w a + (1 - w) (1 - (1 - min(a, 0.99))^(1/k)) for k >= 2 sibling nodes, a itself otherwise (ik = 1 / k, 0 for
// k < 2).  alpha' <= a, so K1's candidate box and skip threshold -- built from the unmapped opacity -- stay conservative.
// d: also d alpha' / d a (zero through the inner cap, as autograd differentiates the clamp).
__device__ __forceinline__ float lod_alpha_remap(float a, float w, float ik, float* d) {
  const float ac = fminf(a, kAlphaMax);
  const float base = 1.0f - ac;
  const float pw = __builtin_amdgcn_exp2f(ik * __builtin_amdgcn_logf(base));          // base^(1/k), base in [0.01, 1]
  const bool on = ik > 0.0f;
  if (d) *d = on ? w + (1.0f - w) * (a <= kAlphaMax ? ik * pw * __builtin_amdgcn_rcpf(base) : 0.0f) : 1.0f;
  return on ? w * a + (1.0f - w) * (1.0f - pw) : a;
}

// ---- wave priority by remaining work (K7) -------------------------------------------------------------------------------
// The SIMD's arbiter serves the OLDEST of its waves first: of the four tiles a SIMD holds, the one dispatched first runs
// at nearly the speed of a wave alone and the youngest gets what is left, so the waves of a launch's last round end one
// after the other and the last of them runs alone (timeline of every wave: profiles/r06_notes.md: per SIMD 228 us with
// four waves resident, 77 us with three, two, one).  s_setprio ranks above age: a wave of the launch's LAST ROUND (the
// last compute units x 4 SIMDs x 4 waves workgroups) drops from the top priority over its last three batches (2, 1,
// 0), so a wave about to finish yields to the ones that still have work and the waves of a SIMD end together (275 us
// with four resident, 24 us with fewer); every other wave runs at the top priority throughout.  Of the rules tried --
// the quarter of its batches or of its list a wave has left, every wave dropping over its last batches, by dispatch
// order -- this one measured best with the slot stores below (K7 0.281 -> 0.266 ms on the metric frame, 0.477 -> 0.456
// on heavy_1m, 0.599 -> 0.593 on trained_like_10m, where the drop in EVERY wave costs 0.617).  K6 gains nothing from
// any such rule (measured) and keeps the hardware's order.
__device__ __forceinline__ void set_priority(int q) {      // q wave-uniform (the instruction takes an immediate)
  if (q >= 3) __builtin_amdgcn_s_setprio(3);
  else if (q == 2) __builtin_amdgcn_s_setprio(2);
  else if (q == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// number of set bits of `m` below this lane
__device__ __forceinline__ uint32_t rank_below(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---- lane <-> pixel map -------------------------------------------------------------------------------------------------
// row (= lane >> 4) = quadrant: bit 0 -> right half, bit 1 -> lower half of the tile.  Inside a quadrant lanes 0-7 own
// the columns of the even rows' first pixel: lane l of the row -> x = l & 7, y0 = (l >> 3) & 1, pixels at y0 + 2 s.
struct LaneGeom {
  int q, lx, ly0;   // quadrant, x inside the tile, y of pixel 0 inside the tile
};
__device__ __forceinline__ LaneGeom lane_geom(int lane) {
  LaneGeom g;
  g.q = lane >> 4;
  g.lx = (lane & 7) + ((g.q & 1) << 3);
  g.ly0 = ((lane >> 3) & 1) + ((g.q >> 1) << 3);
  return g;
}

// What the staging lane decides for its instance: which quadrants can hold a candidate pixel at all.
//  1. K1's alpha >= 1/255 BOX (ext_x, ext_y around the tile-relative centre) against the quadrants' pixel ranges;
//  2. the exact test "max of the exponent over the quadrant's rectangle >= skip threshold": the box of a
//     slanted ellipse reaches quadrants the ellipse itself misses (7 % of the visits on the benchmark scene, 12 % on the
//     heavy one, more with needles).  The exponent is a concave quadratic, so its maximum over a rectangle that does not
//     hold the centre lies on an edge FACING the centre (from any other boundary point the segment towards the centre
//     runs through the rectangle, along it the exponent grows): one 1-D maximisation per facing edge, at most two per
//     quadrant.  Continuous rectangle >= its pixel centres, plus a guard of 0.02 in the base-2 exponent: conservative;
//     the per-pixel candidate and alpha tests still take every decision.  A NaN (degenerate conic) counts as a hit.
struct QuadHit {
  bool q0, q1, q2, q3;
};
// maximum over dy in [ly, hy] of  A2 dx^2 + B2 dx dy + C2 dy^2  at fixed dx (C2 < 0); nb2c = -B2 / (2 C2)
__device__ __forceinline__ float edge_max(float A2, float B2, float C2, float nb2c, float dx, float ly, float hy) {
  const float dy = fminf(fmaxf(nb2c * dx, ly), hy);
  return fmaf(dy, fmaf(C2, dy, B2 * dx), A2 * dx * dx);
}
__device__ __forceinline__ QuadHit quad_hit(float gxt, float gyt, float ex, float ey, float A2, float B2, float C2,
                                            float thr) {
  const bool xl = (gxt - ex <= 7.0f) && (gxt + ex >= 0.0f);
  const bool xr = (gxt - ex <= 15.0f) && (gxt + ex >= 8.0f);
  const bool yt = (gyt - ey <= 7.0f) && (gyt + ey >= 0.0f);
  const bool yb = (gyt - ey <= 15.0f) && (gyt + ey >= 8.0f);
  QuadHit h{xl && yt, xr && yt, xl && yb, xr && yb};
  // pixel - centre ranges of the two column halves and the two row halves
  const float lx[2] = {0.0f - gxt, 8.0f - gxt}, hx[2] = {7.0f - gxt, 15.0f - gxt};
  const float ly[2] = {0.0f - gyt, 8.0f - gyt}, hy[2] = {7.0f - gyt, 15.0f - gyt};
  const float nb2c = -0.5f * B2 * __builtin_amdgcn_rcpf(C2), nb2a = -0.5f * B2 * __builtin_amdgcn_rcpf(A2);
  const float lim = thr - 0.02f;
  bool e[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cx = q & 1, cy = q >> 1;
    const bool in_x = lx[cx] <= 0.0f && hx[cx] >= 0.0f, in_y = ly[cy] <= 0.0f && hy[cy] >= 0.0f;
    // the vertical edge facing the centre (the nearer one) and the horizontal one; meaningless when in_x / in_y
    const float dxe = lx[cx] > 0.0f ? lx[cx] : hx[cx];
    const float dye = ly[cy] > 0.0f ? ly[cy] : hy[cy];
    const float pv = edge_max(A2, B2, C2, nb2c, dxe, ly[cy], hy[cy]);
    const float ph = edge_max(C2, B2, A2, nb2a, dye, lx[cx], hx[cx]);
    const bool miss_v = in_x || (pv < lim), miss_h = in_y || (ph < lim);     // (NaN: not a miss)
    e[q] = (in_x && in_y) || !(miss_v && miss_h);
  }
  h.q0 = h.q0 && e[0]; h.q1 = h.q1 && e[1]; h.q2 = h.q2 && e[2]; h.q3 = h.q3 && e[3];
  return h;
}

// ================================================================================
// forward
// ================================================================================
template <bool DEPTH>
struct FwdPair {
  f2 fly, T, Cr, Cg, Cb, Dd;
  uint32_t last0, last1;
};

template <bool DEPTH, bool LODA = false>
__device__ __forceinline__ void fwd_pair_live(FwdPair<DEPTH>& p, f2 pw, const float4& q1,
                                              const float4& q2, uint32_t idx1, float lw = 0.0f, float lik = 0.0f) {
  // exp2(min(pw, 0)): the conic is positive definite (0.3 px^2 was added to the covariance's diagonal, K1 keeps the
  // rounded conic positive definite), so the exponent is <= 0 up to rounding; clamping replaces the reference lineage's
  // "power > 0 -> skip" test, which in exact arithmetic never fires, by the value the exact exponent would give, and
  // costs nothing (output clamp of the exp instruction)
  const f2 G = {exp2_le1(pw.x), exp2_le1(pw.y)};
  f2 araw = q1.y * G;
  if constexpr (LODA) araw = f2{lod_alpha_remap(araw.x, lw, lik, nullptr), lod_alpha_remap(araw.y, lw, lik, nullptr)};
  const f2 alpha = {fminf(kAlphaMax, araw.x), fminf(kAlphaMax, araw.y)};
  // no per-lane candidate flag: alpha >= 1/255 implies the log-domain candidate test (which has a 1e-3 guard band);
  // the dummy instance has opacity 0
  const bool live0 = alpha.x >= kAlphaMin;
  const bool live1 = alpha.y >= kAlphaMin;
  const f2 Tn = p.T * (1.0f - alpha);
  // ONE compare per pixel for the saturation test: "blend" and "stop" are both derived from its wave mask in scalar
  // registers (written with booleans the compiler emits a second, complementary compare per pixel)
  const uint64_t l0 = __ballot(live0), l1 = __ballot(live1);
  const uint64_t g0 = __ballot(!(Tn.x < kTEps)), g1 = __ballot(!(Tn.y < kTEps));
  const bool blend0 = __builtin_amdgcn_inverse_ballot_w64(l0 & g0), blend1 = __builtin_amdgcn_inverse_ballot_w64(l1 & g1);
  const bool stop0 = __builtin_amdgcn_inverse_ballot_w64(l0 & ~g0), stop1 = __builtin_amdgcn_inverse_ballot_w64(l1 & ~g1);
  // non-blending lanes: weight 0 (the identity of the colour recurrences), T kept -- as selects (two plain
  // instructions per quantity; zeroing alpha and redoing the packed products costs more)
  const f2 aT = alpha * p.T;
  const f2 w = {blend0 ? aT.x : 0.0f, blend1 ? aT.y : 0.0f};
  p.Cr = fma2(w, splat(q1.z), p.Cr);
  p.Cg = fma2(w, splat(q1.w), p.Cg);
  p.Cb = fma2(w, splat(q2.x), p.Cb);
  if (DEPTH) p.Dd = fma2(w, splat(q2.y), p.Dd);
  p.T = f2{blend0 ? Tn.x : p.T.x, blend1 ? Tn.y : p.T.y};
  p.last0 = blend0 ? idx1 : p.last0;
  p.last1 = blend1 ? idx1 : p.last1;
  p.fly.x = stop0 ? kBig : p.fly.x;      // a saturated pixel leaves the tile (see above); whether a whole quadrant
  p.fly.y = stop1 ? kBig : p.fly.y;      // is finished is checked once per batch, not per stop event
}

// (6 waves per SIMD at 78 registers: 8 160 one-wave tiles on 1 024 SIMDs are 1.33 rounds; fewer resident waves cost more
// than whole rounds win, a 64-register build spills -- profiles/r05_occupancy_vs_rounds.txt)
template <bool DEPTH, bool LODA>
__global__ __launch_bounds__(64) void render_fwd_quad_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ bg,
    float* __restrict__ out_color, float* __restrict__ out_invdepth, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ order, const float* __restrict__ lod_w,
    const int32_t* __restrict__ lod_kids) {
  constexpr int kB = kFwdBatch;
  // float4 per staged Gaussian: (gxt,gyt,A2,B2) (C2,o,r,g) (b,1/z,thr,-); LODA: + (weight, 1 / siblings or 0, -, -)
  constexpr int kLds = LODA ? 4 : 3;
  __shared__ float4 lrec[(kB + 1) * kLds];
  // entry `it`: 16 bits per quadrant q = the it-th instance of q's list as the BYTE OFFSET of its record in `lrec`
  // (instance kB: none); three spare entries for the look-ahead
  __shared__ uint2 qlist[kB + 3];
  constexpr uint32_t kRecBytes = kLds * 16;

  TileGeom tg;
  if (!block_to_tile(T, gx, order, tg)) return;
  const int lane = threadIdx.x;
  const LaneGeom lg = lane_geom(lane);
  const int px = tg.tx * kTile + lg.lx;
  const int py0 = tg.ty * kTile + lg.ly0;
  const float flx = (float)lg.lx;
  const float tile_x0 = (float)(tg.tx * kTile), tile_y0 = (float)(tg.ty * kTile);

  bool inside[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) inside[s] = (px < W) && (py0 + 2 * s < H);
  FwdPair<DEPTH> P0, P1;
  P0.fly = f2{inside[0] ? (float)lg.ly0 : kBig, inside[1] ? (float)(lg.ly0 + 2) : kBig};
  P1.fly = f2{inside[2] ? (float)(lg.ly0 + 4) : kBig, inside[3] ? (float)(lg.ly0 + 6) : kBig};
  P0.T = P1.T = splat(1.0f);
  P0.Cr = P0.Cg = P0.Cb = P0.Dd = P1.Cr = P1.Cg = P1.Cb = P1.Dd = splat(0.0f);
  P0.last0 = P0.last1 = P1.last0 = P1.last1 = 0;

  const uint32_t r0 = ranges[tg.tile * 2 + 0], r1 = ranges[tg.tile * 2 + 1];
  // quadrants that still have an unfinished pixel (bits 16 q .. 16 q + 15 of the ballot)
  uint64_t alive = __ballot(inside[0] || inside[1] || inside[2] || inside[3]);
  if (lane == 0) {     // the dummy instance: opacity 0, threshold +inf
    lrec[kB * kLds + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
    lrec[kB * kLds + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    lrec[kB * kLds + 2] = make_float4(0.f, 0.f, __builtin_inff(), 0.f);
    if constexpr (LODA) lrec[kB * kLds + 3] = make_float4(1.f, 0.f, 0.f, 0.f);
  }
  const uint16_t* myq = reinterpret_cast<const uint16_t*>(qlist) + lg.q;

  for (uint32_t base = r0; base < r1 && alive != 0; base += kB) {
    // wave-uniform by construction; readfirstlane tells the compiler so
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)min((uint32_t)kB, r1 - base));
    __syncthreads();
    QuadHit hit{false, false, false, false};
    if ((uint32_t)lane < n) {
      const uint32_t gid = point_list[base + lane];
      const float4* r = records + (size_t)gid * kRecVec;
      float4 a0 = r[0], a2 = r[2];
      const float4 a3 = r[3];
      a0.x = (a0.x - tile_x0) + a3.x;       // tile-relative pixel centre, once per (tile, Gaussian)
      a0.y = (a0.y - tile_y0) + a3.y;
      const float4 a1 = r[1];
      hit = quad_hit(a0.x, a0.y, a2.z, a3.w, a0.z, a0.w, a1.x, a3.z);
      a2.z = a3.z;                           // skip threshold
      lrec[lane * kLds + 0] = a0;
      lrec[lane * kLds + 1] = a1;
      lrec[lane * kLds + 2] = a2;
      if constexpr (LODA) {
        const int kids = lod_kids[gid];
        lrec[lane * kLds + 3] = make_float4(lod_w[gid], kids >= 2 ? 1.0f / (float)kids : 0.0f, 0.f, 0.f);
      }
    }
    // a finished quadrant takes no more instances
    hit.q0 = hit.q0 && (alive & 0xffffull) != 0;
    hit.q1 = hit.q1 && (alive & 0xffff0000ull) != 0;
    hit.q2 = hit.q2 && (alive & 0xffff00000000ull) != 0;
    hit.q3 = hit.q3 && (alive & 0xffff000000000000ull) != 0;
    // one mask per quadrant in scalar registers
    const uint64_t m0 = __ballot(hit.q0), m1 = __ballot(hit.q1), m2 = __ballot(hit.q2), m3 = __ballot(hit.q3);
    // the quadrants' lists, front to back: entry = rank of the instance inside its quadrant's mask.  (DS operations of
    // one wave execute in order: the fill is complete before the byte stores.)
    for (int i = lane; i < kB + 3; i += 64) qlist[i] = make_uint2(0x00010001u * (kB * kRecBytes), 0x00010001u * (kB * kRecBytes));
    uint16_t* ql16 = reinterpret_cast<uint16_t*>(qlist);
    const uint16_t mine = (uint16_t)((uint32_t)lane * kRecBytes);
    if (hit.q0) ql16[rank_below(m0) * 4 + 0] = mine;
    if (hit.q1) ql16[rank_below(m1) * 4 + 1] = mine;
    if (hit.q2) ql16[rank_below(m2) * 4 + 2] = mine;
    if (hit.q3) ql16[rank_below(m3) * 4 + 3] = mine;
    const int nmax = max(max(__builtin_popcountll(m0), __builtin_popcountll(m1)),
                         max(__builtin_popcountll(m2), __builtin_popcountll(m3)));
    __syncthreads();
    // one (instance, quadrant) pair per row of the wave
    auto visit = [&](uint32_t j, const float4& q0, const float4& q1, const float4& q2) {
      float lw = 0.0f, lik = 0.0f;
      if constexpr (LODA) {
        const float4 q3 = record_at(lrec, j)[3];
        lw = q3.x;
        lik = q3.y;
      }
      const float gyt = q0.y;
      const float dx = q0.x - flx;
      const float ax = q0.z * dx * dx;
      const float bx = q0.w * dx;
      // the last contributor is tracked in the entries' unit ((index + 1) x record bytes) and divided once, at the end
      const uint32_t idx1 = (base - r0 + 1) * kRecBytes + j;
      const f2 dy0 = gyt - P0.fly;
      const f2 pw0 = fma2(dy0, fma2(splat(q1.x), dy0, splat(bx)), splat(ax));
      const f2 dy1 = gyt - P1.fly;
      const f2 pw1 = fma2(dy1, fma2(splat(q1.x), dy1, splat(bx)), splat(ax));
      // (no wave-level "nobody is a candidate" exit: with the exact quadrant test it almost never fires, and the three
      // maxima and the compare it needs cost 4 % of the kernel)
      fwd_pair_live<DEPTH, LODA>(P0, pw0, q1, q2, idx1, lw, lik);
      fwd_pair_live<DEPTH, LODA>(P1, pw1, q1, q2, idx1, lw, lik);
    };
    uint32_t jA = myq[0], jB = myq[4];
    float4 A0 = record_at(lrec, jA)[0], A1 = record_at(lrec, jA)[1], A2 = record_at(lrec, jA)[2];
    for (int it = 0; it < nmax; it += 2) {
      const float4 B0 = record_at(lrec, jB)[0], B1 = record_at(lrec, jB)[1], B2 = record_at(lrec, jB)[2];
      const uint32_t jA2 = myq[(it + 2) * 4];
      visit(jA, A0, A1, A2);
      if (it + 1 < nmax) {
        A0 = record_at(lrec, jA2)[0]; A1 = record_at(lrec, jA2)[1]; A2 = record_at(lrec, jA2)[2];
        const uint32_t jB2 = myq[(it + 3) * 4];
        visit(jB, B0, B1, B2);
        jB = jB2;
      }
      jA = jA2;
    }
    alive = __ballot(fminf(fminf(P0.fly.x, P0.fly.y), fminf(P1.fly.x, P1.fly.y)) < kBig);
  }

  const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
  const size_t plane = (size_t)W * H;
  const float Tf[4] = {P0.T.x, P0.T.y, P1.T.x, P1.T.y};
  const float cr[4] = {P0.Cr.x, P0.Cr.y, P1.Cr.x, P1.Cr.y};
  const float cg[4] = {P0.Cg.x, P0.Cg.y, P1.Cg.x, P1.Cg.y};
  const float cb[4] = {P0.Cb.x, P0.Cb.y, P1.Cb.x, P1.Cb.y};
  const float dd[4] = {P0.Dd.x, P0.Dd.y, P1.Dd.x, P1.Dd.y};
  const uint32_t la[4] = {P0.last0 / kRecBytes, P0.last1 / kRecBytes, P1.last0 / kRecBytes, P1.last1 / kRecBytes};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (inside[s]) {
      const size_t pix = (size_t)(py0 + 2 * s) * W + px;
      out_color[pix] = cr[s] + Tf[s] * b0;
      out_color[plane + pix] = cg[s] + Tf[s] * b1;
      out_color[2 * plane + pix] = cb[s] + Tf[s] * b2;
      if (DEPTH) out_invdepth[pix] = dd[s];
      final_T[pix] = Tf[s];
      n_contrib[pix] = la[s];
    }
  }
}

// ================================================================================
// backward
// ================================================================================
struct BwdPair {
  f2 fly, T, A, bgd, gd;   // A: value blended BEHIND the next Gaussian to be visited, per unit T
  f2 c0, c1, c2;           // dL/dC (r, g, b) of the pair's two pixels
  uint32_t nc0, nc1;
};
// Per-lane partial sums of one (instance, quadrant) over the lane's four pixels, as plain floats.  All four pixels of a
// lane share x, so the three sums that carry dx (sum X dx, sum X dx^2, sum X dx dy) are formed from a0 / a1 just before
// the cross-lane reduction instead of being accumulated per pixel pair.
struct BwdSums {
  float a0, a1, a4, s6, s7, s8, s9;   // sum X, sum X dy, sum X dy^2, sum w dL/dC_rgb, sum w dL/dD
};

// FIRST: the sums are assigned, not accumulated (the first pair of an instance: no zero-filled accumulators)
template <bool DEPTH, bool FIRST, bool LODA = false>
__device__ __forceinline__ void bwd_pair_live(BwdPair& p, BwdSums& S, f2 pw, f2 dy, bool c0, bool c1,
                                              const float4& q1, f2 q2, float lw = 0.0f, float lik = 0.0f) {   // q2 = (blue, 1/z)
  // exp2(min(pw, 0)) as in the forward (the exponent of a positive definite conic is <= 0 up to rounding); it also
  // keeps G finite on the non-live lanes
  const f2 G = {exp2_le1(pw.x), exp2_le1(pw.y)};
  f2 araw0 = q1.y * G;
  f2 xfac = araw0;                                   // d alpha / d power = alpha (per pixel LOD: x d alpha' / d alpha)
  if constexpr (LODA) {
    float d0, d1;
    const f2 og = araw0;
    araw0 = f2{lod_alpha_remap(og.x, lw, lik, &d0), lod_alpha_remap(og.y, lw, lik, &d1)};
    xfac = f2{og.x * d0, og.y * d1};
  }
  const bool live0 = c0 && (araw0.x >= kAlphaMin);   // = blended by the forward (alpha = min(0.99, araw) >= 1/255)
  const bool live1 = c1 && (araw0.y >= kAlphaMin);
  // non-live lanes take part with o G = 0: alpha = 0 is the identity for T and for the A recurrence, and every sum is
  // a multiple of o G (dL/dopacity = sum G dL/dalpha = (sum X) / o is formed by K8 from sum X: no sum of its own)
  const f2 araw = {live0 ? araw0.x : 0.0f, live1 ? araw0.y : 0.0f};
  if constexpr (LODA) xfac = f2{live0 ? xfac.x : 0.0f, live1 ? xfac.y : 0.0f};
  const f2 ae = {fminf(kAlphaMax, araw.x), fminf(kAlphaMax, araw.y)};
  const f2 oma = 1.0f - ae;
  const f2 rinv = {__builtin_amdgcn_rcpf(oma.x), __builtin_amdgcn_rcpf(oma.y)};
  const f2 Tcur = p.T * rinv;                                   // transmittance in front of this Gaussian
  const f2 g0 = p.c0, g1 = p.c1, g2 = p.c2;
  f2 q = fma2(g2, splat(q2.x), fma2(g1, splat(q1.w), g0 * q1.z));
  if (DEPTH) q = fma2(p.gd, splat(q2.y), q);
  const f2 qA = q - p.A;
  const f2 dLda = fma2(qA, Tcur, -(p.bgd * rinv));              // finite on non-live lanes, multiplied by 0 below
  const f2 w = ae * Tcur;
  const f2 X = (LODA ? xfac : araw) * dLda;                     // dL/dpower (straight-through 0.99 cap)
  const f2 Xdy = X * dy;
  // dot products over the pair's two pixels with plain instructions (a packed product + a fold of its halves costs more)
  if (FIRST) {
    S.a0 = X.x + X.y;
    S.a1 = Xdy.x + Xdy.y;
    S.a4 = fmaf(Xdy.y, dy.y, Xdy.x * dy.x);
    S.s6 = fmaf(w.y, g0.y, w.x * g0.x);
    S.s7 = fmaf(w.y, g1.y, w.x * g1.x);
    S.s8 = fmaf(w.y, g2.y, w.x * g2.x);
    if (DEPTH) S.s9 = fmaf(w.y, p.gd.y, w.x * p.gd.x);
  } else {
    S.a0 += X.x + X.y;
    S.a1 += Xdy.x + Xdy.y;
    S.a4 = fmaf(Xdy.y, dy.y, fmaf(Xdy.x, dy.x, S.a4));
    S.s6 = fmaf(w.y, g0.y, fmaf(w.x, g0.x, S.s6));
    S.s7 = fmaf(w.y, g1.y, fmaf(w.x, g1.x, S.s7));
    S.s8 = fmaf(w.y, g2.y, fmaf(w.x, g2.x, S.s8));
    if (DEPTH) S.s9 = fmaf(w.y, p.gd.y, fmaf(w.x, p.gd.x, S.s9));
  }
  p.T = Tcur;
  p.A = fma2(ae, qA, p.A);                                      // A_(i-1) = alpha_i q_i + (1 - alpha_i) A_i
}

// Ten registers of per-lane partial sums -> row totals, for all four rows of the wave at once (every row = another
// instance).  The row is halved three times with DPP adds whose bank masks write the halves of DIFFERENT values next to
// each other, so the registers are packed as the lane count shrinks: 10 registers (16 partials each) -> 5 (2 x 8) -> 3
// (4 x 4) -> totals in every lane of a quad.  21 DPP adds, no cross-row traffic, no LDS.
//   ta: quad b of a row (lanes 4 b .. 4 b + 3) holds the total of v{0,2,1,3}[b]
//   tb: ... of v{4,6,5,7}[b]
//   tc: quads 0, 1: v8; quads 2, 3: v9
// (A DPP operand must not be read within two instructions of the vector instruction that wrote it; the order below
// keeps every such pair at least three instructions apart, the leading s_nop covers the inputs.)
__device__ __forceinline__ void row_reduce10(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                             float v7, float v8, float v9, float& ta, float& tb, float& tc) {
  float r01, r23, r45, r67, r89;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %3, %8, %8 row_mirror row_mask:0xf bank_mask:0x3\n\t"       // r01 lanes 0-7 : v0[i] + v0[15-i]
      "v_add_f32_dpp %3, %9, %9 row_mirror row_mask:0xf bank_mask:0xc\n\t"       // r01 lanes 8-15: v1
      "v_add_f32_dpp %4, %10, %10 row_mirror row_mask:0xf bank_mask:0x3\n\t"     // r23: v2 | v3
      "v_add_f32_dpp %4, %11, %11 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %5, %12, %12 row_mirror row_mask:0xf bank_mask:0x3\n\t"     // r45: v4 | v5
      "v_add_f32_dpp %5, %13, %13 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %6, %14, %14 row_mirror row_mask:0xf bank_mask:0x3\n\t"     // r67: v6 | v7
      "v_add_f32_dpp %6, %15, %15 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %7, %16, %16 row_mirror row_mask:0xf bank_mask:0x3\n\t"     // r89: v8 | v9
      "v_add_f32_dpp %7, %17, %17 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      // 8 partials per half row -> 4: quads 0 / 2 from the first register, quads 1 / 3 from the second
      "v_add_f32_dpp %0, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"  // ta quad 0: v0, quad 2: v1
      "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"  // ta quad 1: v2, quad 3: v3
      "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"  // tb quad 0: v4, quad 2: v5
      "v_add_f32_dpp %1, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"  // tb quad 1: v6, quad 3: v7
      "v_add_f32_dpp %2, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"  // tc quads 0,1: v8, quads 2,3: v9
      // 4 partials per quad -> the total in every lane of the quad
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "=&v"(ta), "=&v"(tb), "=&v"(tc), "=&v"(r01), "=&v"(r23), "=&v"(r45), "=&v"(r67), "=&v"(r89)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "v"(v8), "v"(v9));
}

// four waves per SIMD (128 registers, no scratch); five need 96 and spill, the colour gradients in LDS reach 113:
// profiles/r06_k7_occupancy.txt
constexpr int kK7Waves = 4;
template <bool DEPTH, bool LODA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kK7Waves, kK7Waves))) void render_bwd_quad_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ bg,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const uint32_t* __restrict__ offsets, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_dinvdepth, float* __restrict__ inst, const uint32_t* __restrict__ order,
    const float* __restrict__ lod_w, const int32_t* __restrict__ lod_kids, int last_round_waves) {
  constexpr int kB = kBwdBatch;
  // float4 per staged Gaussian: (gxt,gyt,A2,B2) (C2,o,r,g) (b,1/z,thr,-); LODA: + (weight, 1 / siblings or 0, -, -)
  constexpr int kLds = LODA ? 4 : 3;
  __shared__ float4 lrec[(kB + 1) * kLds];
  // entry `it`: 32 bits per quadrant q = the it-th (instance, quadrant) pair of q's list, as the two BYTE OFFSETS the visit
  // needs: of the staged instance's record in `lrec` (low half; instance kB: none) and of the pair's slot in `acc` (high
  // half; slot kSlots: the sink of the rows that have run out of pairs); three spare entries for the look-ahead
  __shared__ uint4 qlist[kB + 3];
  constexpr uint32_t kRecBytes = kLds * 16, kSlotBytes = kInstStride * 4;
  // One slot of ten sums per (instance, quadrant) pair of the batch, each written by exactly one visit: the pairs of an
  // instance are consecutive (quadrant order) and are added up in that order at the end of the batch, so the result does
  // not depend on what else is in the batch (tests/test_properties_gpu.py: appending Gaussians that cannot contribute
  // changes no bit).  64 instances have up to 256 pairs (a hierarchy cut's big nodes reach all four quadrants); 128 slots
  // hold a whole batch of the benchmark scenes (88 pairs on average) and at least 31 instances of any.
  constexpr int kSlots = LODA ? 124 : 128;      // (the per-pixel LOD build stages 64-byte records: 124 slots keep it at 10 KB)
  __shared__ __attribute__((aligned(8))) float acc[(kSlots + 1) * kInstStride];

  TileGeom tg;
  if (!block_to_tile(T, gx, order, tg)) return;
  const int lane = threadIdx.x;
  const LaneGeom lg = lane_geom(lane);
  const int px = tg.tx * kTile + lg.lx;
  const int py0 = tg.ty * kTile + lg.ly0;
  const float flx = (float)lg.lx;
  const float tile_x0 = (float)(tg.tx * kTile), tile_y0 = (float)(tg.ty * kTile);
  const size_t plane = (size_t)W * H;
  const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
  const uint32_t r0 = ranges[tg.tile * 2 + 0], r1 = ranges[tg.tile * 2 + 1];
  const uint32_t total = r1 - r0;
  if (total == 0) return;

  float Tr[4], bgd[4], g0[4], g1[4], g2[4], gd[4];
  uint32_t nc[4];
  uint32_t maxnc = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int py = py0 + 2 * s;
    Tr[s] = 1.0f; bgd[s] = 0.f; g0[s] = g1[s] = g2[s] = gd[s] = 0.f; nc[s] = 0;   // outside the image: n_contrib 0
    if (px < W && py < H) {
      const size_t pix = (size_t)py * W + px;
      g0[s] = dL_dcolor[pix];
      g1[s] = dL_dcolor[plane + pix];
      g2[s] = dL_dcolor[2 * plane + pix];
      if (DEPTH) gd[s] = dL_dinvdepth[pix];
      Tr[s] = final_T[pix];
      bgd[s] = Tr[s] * (g0[s] * b0 + g1[s] * b1 + g2[s] * b2);
      nc[s] = n_contrib[pix];
    }
    maxnc = max(maxnc, nc[s]);
  }
  // last contributor per quadrant (row of 16 lanes) and of the tile: uniform values, loop counters in scalar registers
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) maxnc = max(maxnc, (uint32_t)__shfl_xor((int)maxnc, off, 64));
  const uint32_t qnc0 = (uint32_t)__builtin_amdgcn_readlane((int)maxnc, 0), qnc1 = (uint32_t)__builtin_amdgcn_readlane((int)maxnc, 16);
  const uint32_t qnc2 = (uint32_t)__builtin_amdgcn_readlane((int)maxnc, 32), qnc3 = (uint32_t)__builtin_amdgcn_readlane((int)maxnc, 48);
  maxnc = max(max(qnc0, qnc1), max(qnc2, qnc3));
  BwdPair P0, P1;
  const float flyb = (float)lg.ly0;                              // y of pixel 0; the pairs' fly is set per batch
  P0.T = f2{Tr[0], Tr[1]};     P1.T = f2{Tr[2], Tr[3]};
  P0.bgd = f2{bgd[0], bgd[1]}; P1.bgd = f2{bgd[2], bgd[3]};
  P0.c0 = f2{g0[0], g0[1]}; P0.c1 = f2{g1[0], g1[1]}; P0.c2 = f2{g2[0], g2[1]};
  P1.c0 = f2{g0[2], g0[3]}; P1.c1 = f2{g1[2], g1[3]}; P1.c2 = f2{g2[2], g2[3]};
  P0.gd = f2{gd[0], gd[1]};    P1.gd = f2{gd[2], gd[3]};
  P0.A = P1.A = splat(0.0f);
  P0.nc0 = nc[0]; P0.nc1 = nc[1]; P1.nc0 = nc[2]; P1.nc1 = nc[3];
  if (lane == 0) {     // the dummy instance: opacity 0, threshold +inf
    lrec[kB * kLds + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
    lrec[kB * kLds + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    lrec[kB * kLds + 2] = make_float4(0.f, 0.f, __builtin_inff(), 0.f);
    if constexpr (LODA) lrec[kB * kLds + 3] = make_float4(1.f, 0.f, 0.f, 0.f);
  }
  const uint32_t* myq = reinterpret_cast<const uint32_t*>(qlist) + lg.q;
  // which float of the pair's slot this lane stores after the row reduction (see row_reduce10): only the first lane of
  // every quad takes part
  const int quad = (lane >> 2) & 3;
  const bool writer = (lane & 3) == 0;
  const int ka = ((quad & 1) << 1) | (quad >> 1);          // {0,2,1,3}[quad]
  const int kc = 8 + (quad >> 1);                          // quads 0 / 2 store v8 / v9
  const bool writer_c = writer && (quad & 1) == 0;

  // Batches back to front, from the last contributor of the tile: lane i stages instance bstart + i of the list (bstart
  // may be negative in the front-most batch: those lanes stage nothing).  Instances behind the last contributor (and every
  // instance of a tile whose pixels blended nothing) only get their zero record written (tail loop below).
  const bool last_round = (int)blockIdx.x >= (int)gridDim.x - last_round_waves;
  for (int hi = (int)maxnc; hi > 0;) {
    const int bstart = hi - kB;
    // a wave of the launch's last round: top priority down to its last three batches (2, 1, 0); the others: top priority
    set_priority(last_round ? (hi + kB - 1) / kB - 1 : 3);
    __syncthreads();
    QuadHit hit{false, false, false, false};
    uint32_t my_off = 0, my_rect = 0;
    const bool staged_lane = bstart + lane >= 0;
    if (staged_lane) {
      const uint32_t gid = point_list[r0 + (uint32_t)(bstart + lane)];
      const float4* r = records + (size_t)gid * kRecVec;
      float4 a0 = r[0], a2 = r[2];
      const float4 a3 = r[3];
      a0.x = (a0.x - tile_x0) + a3.x;           // tile-relative pixel centre, once per (tile, Gaussian)
      a0.y = (a0.y - tile_y0) + a3.y;
      const float4 a1 = r[1];
      // (behind the last contributor of every pixel of a quadrant: nothing to do there)
      const uint32_t rel = (uint32_t)(bstart + lane);
      hit = quad_hit(a0.x, a0.y, a2.z, a3.w, a0.z, a0.w, a1.x, a3.z);
      hit.q0 = hit.q0 && rel < qnc0; hit.q1 = hit.q1 && rel < qnc1;
      hit.q2 = hit.q2 && rel < qnc2; hit.q3 = hit.q3 && rel < qnc3;
      my_off = offsets[gid];
      my_rect = __float_as_uint(a2.w);
      a2.z = a3.z;                               // skip threshold
      lrec[lane * kLds + 0] = a0;
      lrec[lane * kLds + 1] = a1;
      lrec[lane * kLds + 2] = a2;
      if constexpr (LODA) {
        const int kids = lod_kids[gid];
        lrec[lane * kLds + 3] = make_float4(lod_w[gid], kids >= 2 ? 1.0f / (float)kids : 0.0f, 0.f, 0.f);
      }
    }
    uint64_t m0 = __ballot(hit.q0), m1 = __ballot(hit.q1), m2 = __ballot(hit.q2), m3 = __ballot(hit.q3);
    int c0n = __builtin_popcountll(m0), c1n = __builtin_popcountll(m1);
    int c2n = __builtin_popcountll(m2), c3n = __builtin_popcountll(m3);
    uint32_t k0 = rank_below(m0), k1 = rank_below(m1), k2 = rank_below(m2), k3 = rank_below(m3);
    // The pairs of the lanes below this one = this lane's first slot.  More pairs than slots (wave-uniform, rare outside
    // scenes of very large Gaussians): the batch keeps the longest run of its BACK-most lanes whose pairs fit (at most
    // four per lane: never fewer than 31 lanes) and the next batch starts at the first lane left out.
    int first = 0;                                           // first lane of the batch that is processed
    if (c0n + c1n + c2n + c3n > kSlots) {
      const uint32_t from_here = (uint32_t)(c0n + c1n + c2n + c3n) - (k0 + k1 + k2 + k3);      // pairs of lanes >= this one
      const uint64_t keep = __ballot(from_here <= (uint32_t)kSlots);                           // lanes first .. 63
      first = __builtin_ctzll(keep);
      const uint64_t low = ~keep;
      const int d0 = __builtin_popcountll(m0 & low), d1 = __builtin_popcountll(m1 & low);
      const int d2 = __builtin_popcountll(m2 & low), d3 = __builtin_popcountll(m3 & low);
      m0 &= keep; m1 &= keep; m2 &= keep; m3 &= keep;
      c0n -= d0; c1n -= d1; c2n -= d2; c3n -= d3;
      k0 -= (uint32_t)d0; k1 -= (uint32_t)d1; k2 -= (uint32_t)d2; k3 -= (uint32_t)d3;      // (lanes below `first`: unused)
      const bool kept = lane >= first;
      hit.q0 = hit.q0 && kept; hit.q1 = hit.q1 && kept; hit.q2 = hit.q2 && kept; hit.q3 = hit.q3 && kept;
    }
    const uint32_t slot0 = k0 + k1 + k2 + k3;
    const uint32_t npairs = (uint32_t)hit.q0 + (uint32_t)hit.q1 + (uint32_t)hit.q2 + (uint32_t)hit.q3;
    // the quadrants' lists, BACK to front: entry = number of the mask's set bits ABOVE the instance.  (DS operations of
    // one wave execute in order: the fill is complete before the entries' stores.)
    {
      const uint32_t none = (uint32_t)kB * kRecBytes | ((uint32_t)kSlots * kSlotBytes << 16);
      for (int i = lane; i < kB + 3; i += 64) qlist[i] = make_uint4(none, none, none, none);
    }
    uint32_t* ql32 = reinterpret_cast<uint32_t*>(qlist);
    {
      uint32_t en = (uint32_t)lane * kRecBytes | (slot0 * kSlotBytes << 16);
      if (hit.q0) { ql32[(c0n - 1 - (int)k0) * 4 + 0] = en; en += kSlotBytes << 16; }
      if (hit.q1) { ql32[(c1n - 1 - (int)k1) * 4 + 1] = en; en += kSlotBytes << 16; }
      if (hit.q2) { ql32[(c2n - 1 - (int)k2) * 4 + 2] = en; en += kSlotBytes << 16; }
      if (hit.q3) { ql32[(c3n - 1 - (int)k3) * 4 + 3] = en; }
    }
    const int nmax = max(max(c0n, c1n), max(c2n, c3n));
    __syncthreads();
    // "The forward blended this Gaussian into the pixel" <=> rel < n_contrib.  A pixel whose last contributor lies in a
    // LATER batch passes for every instance of this one; a pixel whose last contributor lies in an EARLIER batch (or
    // outside the image: n_contrib 0) fails for every instance and is parked at y = kBig for the batch; only a batch
    // that holds some pixel's last contributor needs the per-instance compare at all (one batch in five on the benchmark;
    // a second instantiation of the loop without it measured slower: profiles/r03_optimisation_ladders.md).
    const uint32_t bs = (uint32_t)max(bstart + first, 0);    // the front-most instance this batch processes
    P0.fly = f2{P0.nc0 > bs ? flyb : kBig, P0.nc1 > bs ? flyb + 2.0f : kBig};
    P1.fly = f2{P1.nc0 > bs ? flyb + 4.0f : kBig, P1.nc1 > bs ? flyb + 6.0f : kBig};
    // "rel < n_contrib" in the units of an entry's low half (staged instance j = rel - bstart as a record offset)
    auto limit = [&](uint32_t nc) { return (uint32_t)min(max((int)nc - bstart, 0), kB + 1) * kRecBytes; };
    const uint32_t lim0 = limit(P0.nc0), lim1 = limit(P0.nc1), lim2 = limit(P1.nc0), lim3 = limit(P1.nc1);

    // one (instance, quadrant) pair per row of the wave
    auto visit = [&](uint32_t e, const float4& q0, const float4& q1, const float4& q2v, auto&& pin_lookahead) {
      const uint32_t jo = e & 0xffffu;                // the staged instance's record offset; e >> 16: the pair's slot offset
      const float gyt = q0.y;
      const float dx = q0.x - flx;
      const float ax = q0.z * dx * dx;
      const float bx = q0.w * dx;
      const f2 dy0 = gyt - P0.fly;
      const f2 pw0 = fma2(dy0, fma2(splat(q1.x), dy0, splat(bx)), splat(ax));
      const f2 dy1 = gyt - P1.fly;
      const f2 pw1 = fma2(dy1, fma2(splat(q1.x), dy1, splat(bx)), splat(ax));
      // per-pixel predicate "the forward blended this Gaussian into the pixel": rel < n_contrib here, alpha >= 1/255 in
      // bwd_pair_live (exactly the forward's test; a parked pixel and the dummy instance have alpha 0).
      const bool c0 = jo < lim0, c1 = jo < lim1;
      const bool c2 = jo < lim2, c3 = jo < lim3;
      const f2 q2 = f2{q2v.x, q2v.y};
      float lw = 0.0f, lik = 0.0f;
      if constexpr (LODA) {
        const float4 q3 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lrec) + jo)[3];
        lw = q3.x;
        lik = q3.y;
      }
      BwdSums S;
      if (!DEPTH) S.s9 = 0.0f;
      bwd_pair_live<DEPTH, true, LODA>(P0, S, pw0, dy0, c0, c1, q1, q2, lw, lik);
      bwd_pair_live<DEPTH, false, LODA>(P1, S, pw1, dy1, c2, c3, q1, q2, lw, lik);
      const float s0 = dx * S.a0, s3 = dx * S.a1;     // sum X dx, sum X dx dy
      const float s2 = dx * s0;                        // sum X dx^2
      float ta, tb, tc;
      row_reduce10(s0, S.a1, s2, s3, S.a4, S.a0, S.s6, S.s7, S.s8, S.s9, ta, tb, tc);
      // The look-ahead's LDS reads are made to land HERE, in front of the branch around the stores: behind a branch the
      // compiler no longer knows how many LDS operations are in flight and waits for all of them at the next use of a
      // look-ahead register -- the first instruction of the next visit.
      pin_lookahead();
      // the pair's own slot (rows that have run out of pairs: the sink)
      if (writer) {
        float* row = reinterpret_cast<float*>(reinterpret_cast<char*>(acc) + (e >> 16));
        row[ka] = ta;
        row[4 + ka] = tb;
        if (writer_c) row[kc] = tc;
      }
    };
    {
      // Two iterations per trip, the record of the NEXT iteration and the list entry of the one after it requested
      // before the current one is composited: every row's record address depends on a list entry that is itself in
      // LDS, and at 4 waves per SIMD two dependent LDS round trips per iteration are not hidden by the other waves.
      uint32_t jA = myq[0], jB = myq[4];
      float4 A0 = entry_record(lrec, jA)[0], A1 = entry_record(lrec, jA)[1], A2 = entry_record(lrec, jA)[2];
      // (nothing of the loop's first trip is left pending at its head either)
      asm volatile("" : : "v"(jB), "v"(A0.x), "v"(A0.y), "v"(A0.z), "v"(A0.w), "v"(A1.x), "v"(A1.y), "v"(A1.z), "v"(A1.w), "v"(A2.x),
                   "v"(A2.y));
      for (int it = 0; it < nmax; it += 2) {
        const float4 B0 = entry_record(lrec, jB)[0], B1 = entry_record(lrec, jB)[1], B2 = entry_record(lrec, jB)[2];
        const uint32_t jA2 = myq[(it + 2) * 4];
        visit(jA, A0, A1, A2, [&] {
          asm volatile("" : : "v"(B0.x), "v"(B0.y), "v"(B0.z), "v"(B0.w), "v"(B1.x), "v"(B1.y), "v"(B1.z), "v"(B1.w), "v"(B2.x),
                       "v"(B2.y), "v"(jA2));
        });
        if (it + 1 < nmax) {
          A0 = entry_record(lrec, jA2)[0]; A1 = entry_record(lrec, jA2)[1]; A2 = entry_record(lrec, jA2)[2];
          const uint32_t jB2 = myq[(it + 3) * 4];
          visit(jB, B0, B1, B2, [&] {
            asm volatile("" : : "v"(A0.x), "v"(A0.y), "v"(A0.z), "v"(A0.w), "v"(A1.x), "v"(A1.y), "v"(A1.z), "v"(A1.w), "v"(A2.x),
                         "v"(A2.y), "v"(jB2));
          });
          jB = jB2;
        }
        jA = jA2;
      }
    }
    __syncthreads();
    // lane i adds up instance i's pairs (quadrant order) and stores the record to its emission slot: every processed
    // instance is written, reached or not
    if (staged_lane && lane >= first) {
      const uint32_t minx = my_rect & 1023u, miny = (my_rect >> 10) & 1023u, rw = my_rect >> 20;
      const uint32_t e = my_off + ((uint32_t)tg.ty - miny) * rw + ((uint32_t)tg.tx - minx);
      f2* dst = reinterpret_cast<f2*>(inst + (size_t)e * kInstStride);
      const f2* row = reinterpret_cast<const f2*>(acc + slot0 * kInstStride);
      f2 sum[kInstStride / 2];
#pragma unroll
      for (int k = 0; k < kInstStride / 2; ++k) sum[k] = npairs > 0 ? row[k] : splat(0.0f);
#pragma unroll
      for (int t = 1; t < 4; ++t) {
        if ((uint32_t)t < npairs) {
#pragma unroll
          for (int k = 0; k < kInstStride / 2; ++k) sum[k] += row[t * (kInstStride / 2) + k];
        }
      }
#pragma unroll
      for (int k = 0; k < kInstStride / 2; ++k) dst[k] = sum[k];
    }
    hi = bstart + first;
  }
  // instances behind the last contributor of every pixel: zero records
  for (int i = (int)maxnc + lane; i < (int)total; i += 64) {
    const uint32_t gid = point_list[r0 + i];
    const uint32_t rb = __float_as_uint(reinterpret_cast<const float*>(records + (size_t)gid * kRecVec + 2)[3]);
    const uint32_t minx = rb & 1023u, miny = (rb >> 10) & 1023u, rw = rb >> 20;
    const uint32_t e = offsets[gid] + ((uint32_t)tg.ty - miny) * rw + ((uint32_t)tg.tx - minx);
    f2* dst = reinterpret_cast<f2*>(inst + (size_t)e * kInstStride);
#pragma unroll
    for (int k = 0; k < kInstStride / 2; ++k) dst[k] = splat(0.0f);
  }
}

}  // namespace

int launch_render_fwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      float* out_color, float* out_invdepth, hipStream_t s) {
  const int gxx = grid_x(a.width), T = gxx * grid_y(a.height);
  const int nblk = ((T + 7) / 8) * 8;
  const bool depth = a.do_depth && out_invdepth;
  const bool loda = a.lod_per_pixel && a.interpolation_weights && a.num_node_kids;
  auto kern = loda ? (depth ? render_fwd_quad_kernel<true, true> : render_fwd_quad_kernel<false, true>)
                   : (depth ? render_fwd_quad_kernel<true, false> : render_fwd_quad_kernel<false, false>);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, s, b.ranges, b.vals_out,
                     reinterpret_cast<const float4*>(g.records), a.width, a.height, gxx, T, a.bg, out_color,
                     out_invdepth, im.final_T, im.n_contrib, b.tile_order, a.interpolation_weights, a.num_node_kids);
  HGS_LAUNCH_CHECK("render_fwd_quad", s, a.debug);
  return HGS_OK;
}

int launch_render_bwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      const float* out_color, const float* out_invdepth, const float* dL_dcolor,
                      const float* dL_dinvdepth, float* inst_grads, hipStream_t s) {
  (void)out_color;
  const int gxx = grid_x(a.width), T = gxx * grid_y(a.height);
  const int nblk = ((T + 7) / 8) * 8;
  const bool depth = a.do_depth && out_invdepth && dL_dinvdepth;
  const bool loda = a.lod_per_pixel && a.interpolation_weights && a.num_node_kids;
  auto kern = loda ? (depth ? render_bwd_quad_kernel<true, true> : render_bwd_quad_kernel<false, true>)
                   : (depth ? render_bwd_quad_kernel<true, false> : render_bwd_quad_kernel<false, false>);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, s, b.ranges, b.vals_out,
                     reinterpret_cast<const float4*>(g.records), a.width, a.height, gxx, T, a.bg, im.final_T,
                     im.n_contrib, g.offsets, dL_dcolor, dL_dinvdepth, inst_grads, b.tile_order, a.interpolation_weights,
                     a.num_node_kids, scan_resident_workgroups() * 4 * kK7Waves);      // wave slots of the device
  HGS_LAUNCH_CHECK("render_bwd_quad", s, a.debug);
  return HGS_OK;
}

}  // namespace hgs
