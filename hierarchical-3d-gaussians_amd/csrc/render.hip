// K6 (tile compositing, forward) and K7 (its backward).
//
// Restates per-tile front-to-back alpha compositing with an inverse-depth channel
// (op call sites gaussian_renderer/__init__.py:105-113,269-277;
// backward entered from train_single.py:123 / train_post.py:142).
//
// CDNA4 design (not the CUDA 256-threads-one-pixel-each shape):
//  * a 16x16 tile is cut into four 16x4 STRIPS; a lane owns four pixels (same x, y = y0 + 4*s) and ONE WAVE RENDERS
//    A WHOLE TILE: the per-Gaussian record is read from LDS once per tile, no workgroup barrier is needed, and the
//    backward's cross-lane reduction runs once per (tile, Gaussian).  Measured 0.53 / 0.37 / 0.35 ms forward for
//    4 / 2 / 1 waves per tile before anything else was tuned (round 1).
//  * the two strips of a PAIR (rows 0-7 resp. 8-15 of the tile) go through one instruction stream as float2 values;
//    compares, selects, min / max, exp and rcp have no packed form and are halved by that (packed FMAs themselves cost
//    as much as two plain ones on this chip: scripts/microbench/valu_issue.hip).
//  * who is visited: K1 stores, per Gaussian, the half extents of the box around its alpha >= 1/255 region; the lane
//    that stages a Gaussian into LDS tests that box against the tile's two halves, two ballots turn the answers
//    into 64-bit masks in scalar registers, and the loop walks their set bits.  A Gaussian whose box misses the tile
//    costs no vector instruction, one that reaches a single half never computes the other half's exponents.
//  * inside a visited half: alpha >= 1/255  <=>  power2 >= log2(1/255) - log2(opacity).  If no pixel of the half is above
//    that threshold (minus a 1e-3 guard band) the half is left after 5 vector instructions, before the exp; the exact
//    alpha test still takes every borderline decision, so results do not depend on either pre-test.
//  * pixel coordinates are TILE-RELATIVE: the record carries the pixel centre as hi + lo floats from K1's
//    double-precision projection; (hi - tile_origin) + lo is exact to ~1e-6 px at any resolution.
//  * a finished / outside pixel is moved to y = 1e18 (never a candidate again): "done" needs no flag.
//  * backward: BACK-TO-FRONT from the forward's last contributor with
//        T_i = T_{i+1} / (1 - alpha_i),   A_i = alpha_{i+1} q_{i+1} + (1 - alpha_{i+1}) A_{i+1},
//        dL/dalpha_i = (q_i - A_i) T_i - T_final (dL/dC . bg) / (1 - alpha_i),   q = dL/dC . c + dL/dD / z
//    (every term relatively accurate; a front-to-back variant using V - prefix was measured 10-80x
//    less accurate on pixels with capped alphas and was dropped).
//  * the 10 per-(tile,Gaussian) partial sums are reduced over the 4 strips in registers, over the
//    wave with permlane swaps + bank-masked DPP adds (three registers of row partials share one reduction), and ten
//    lanes store the instance record to its EMISSION slot with one instruction; K8 sums each Gaussian's
//    contiguous run.  No LDS accumulator, no atomics of any kind, bit-reproducible.
#include "common.h"

namespace hgs {
namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTEps = 0.0001f;
constexpr float kBig = 1.0e18f;   // y coordinate of a finished / outside pixel: its power is -inf-ish, never a candidate

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over each row of 16 lanes; every lane of the row ends with the row total
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}
// Three registers of 16-lane partial sums (every row of 16 lanes = one value) -> ONE register in which every quad of
// a row holds a total: lanes 0-3 = sum of a's row, lanes 8-11 = b's, lanes 4-7 of rows 1 / 3 = c's summed over the row
// PAIR (0,1) / (2,3).  The row is halved with bank-masked DPP adds that write their results next to each other
// instead of into separate registers: 8 DPP adds where three full row reductions and the row-pair add take 13.
// (s_nop: a DPP operand must not be read within two instructions of the vector instruction that wrote it, and the
// compiler does not see into the asm block.)
__device__ __forceinline__ float row_sum16_x3(float a, float b, float c) {
  float ab, abc, c1;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %2, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n\t"        // c1: lanes i, 15-i = c[i] + c[15-i]
      "v_add_f32_dpp %0, %3, %3 row_mirror row_mask:0xf bank_mask:0x3\n\t"        // lanes 0-7 : a[i] + a[15-i]
      "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"        // lanes 8-15: b[i] + b[15-i]
      "s_nop 0\n\t"
      "v_add_f32_dpp %1, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"   // lanes 4-7, 12-15: c, four partials
      "v_add_f32_dpp %1, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"   // lanes 0-3: a, lanes 8-11: b
      "s_nop 1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      // c's values (s8 / s9) occupy two rows each (0,1 / 2,3): lane 15 of rows 0 / 2 (a copy of c's row total) is
      // added into the c quad (lanes 4-7) of rows 1 / 3
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0x2"
      : "=&v"(ab), "=&v"(abc), "=&v"(c1)
      : "v"(a), "v"(b), "v"(c));
  return abc;
}

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

// ================================================================================
// The kernels: one wave per tile, the four strips processed as two PACKED pairs.
//
// Both compositing kernels are vector-issue-bound (profiles/pmc_valu.json: the vector ALUs are busy ~88 % of a launch)
// and, at 6 - 8 waves per SIMD, sensitive to the dependent chain of an instance as well: the levers are instruction
// count, cheap instruction forms (plain instead of packed where nothing is gained by packing, output modifiers,
// mask algebra in scalar registers) and as few LDS round trips per instance as possible.  Non-live lanes are handled
// by zeroing alpha (an alpha = 0 Gaussian is the identity for every recurrence used here), not by select-updating the
// state.  The ISA of the two inner loops was read after every change (DESIGN.md, K6/K7 ladder).
// ================================================================================
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }

// exp2(min(x, 0)) = min(exp2(x), 1) as ONE instruction: the [0, 1] output clamp of v_exp_f32 (the compiler folds the
// median into the instruction's clamp bit).  Also turns +inf / NaN inputs into 1 / 0.
__device__ __forceinline__ float exp2_le1(float x) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x), 0.0f, 1.0f); }

template <bool DEPTH>
struct FwdPair {
  f2 fly, T, Cr, Cg, Cb, Dd;
  uint32_t last0, last1;
};

template <bool DEPTH>
__device__ __forceinline__ void fwd_pair_live(FwdPair<DEPTH>& p, f2 pw, const float4& q1,
                                              const float4& q2, uint32_t idx1) {
  // exp2(min(pw, 0)): the conic is positive definite (0.3 px^2 was added to the covariance's diagonal, K1 keeps the
  // rounded conic positive definite), so the exponent is <= 0 up to rounding; clamping replaces the reference lineage's
  // "power > 0 -> skip" test, which in exact arithmetic never fires, by the value the exact exponent would give, and
  // costs nothing (output clamp of the exp instruction)
  const f2 G = {exp2_le1(pw.x), exp2_le1(pw.y)};
  const f2 araw = q1.y * G;
  const f2 alpha = {fminf(kAlphaMax, araw.x), fminf(kAlphaMax, araw.y)};
  // no per-lane candidate flag: alpha >= 1/255 implies the log-domain candidate test (which has a 1e-3 guard band)
  const bool live0 = alpha.x >= kAlphaMin;
  const bool live1 = alpha.y >= kAlphaMin;
  const f2 Tn = p.T * (1.0f - alpha);
  // ONE compare per strip for the saturation test: "blend" and "stop" are both derived from its wave mask in scalar
  // registers (written with booleans the compiler emits a second, complementary compare per strip)
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
  p.fly.x = stop0 ? kBig : p.fly.x;      // a saturated pixel leaves the tile (see above); whether the WHOLE tile
  p.fly.y = stop1 ? kBig : p.fly.y;      // is finished is checked once per batch, not per stop event
}

template <bool DEPTH>
__global__ __launch_bounds__(64) void render_fwd_packed_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ bg,
    float* __restrict__ out_color, float* __restrict__ out_invdepth, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float4* __restrict__ zero_ws, uint32_t zero_vecs,
    const uint32_t* __restrict__ order) {
  constexpr int BATCH = 64;
  constexpr int kLds = 3;   // float4 per staged Gaussian: (gxt,gyt,A2,B2) (C2,o,r,g) (b,1/z,thr,-)
  __shared__ float4 lrec[BATCH * kLds];

  TileGeom tg;
  if (!block_to_tile(T, gx, order, tg)) return;
  const int lane = threadIdx.x;
  const int lx = lane & 15, ly0 = lane >> 4;
  const int px = tg.tx * kTile + lx;
  const int py0 = tg.ty * kTile + ly0;
  const float flx = (float)lx;
  const float tile_x0 = (float)(tg.tx * kTile), tile_y0 = (float)(tg.ty * kTile);

  bool inside[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) inside[s] = (px < W) && (py0 + 4 * s < H);
  FwdPair<DEPTH> P0, P1;
  P0.fly = f2{inside[0] ? (float)ly0 : kBig, inside[1] ? (float)(ly0 + 4) : kBig};
  P1.fly = f2{inside[2] ? (float)(ly0 + 8) : kBig, inside[3] ? (float)(ly0 + 12) : kBig};
  P0.T = P1.T = splat(1.0f);
  P0.Cr = P0.Cg = P0.Cb = P0.Dd = P1.Cr = P1.Cg = P1.Cb = P1.Dd = splat(0.0f);
  P0.last0 = P0.last1 = P1.last0 = P1.last1 = 0;

  const uint32_t r0 = ranges[tg.tile * 2 + 0], r1 = ranges[tg.tile * 2 + 1];
  bool wave_done = (__ballot(inside[0] || inside[1] || inside[2] || inside[3]) == 0);

  for (uint32_t base = r0; base < r1 && !wave_done; base += BATCH) {
    // wave-uniform by construction; readfirstlane tells the compiler so
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)min((uint32_t)BATCH, r1 - base));
    __syncthreads();
    // staging lane = one Gaussian of the batch: besides copying its record it decides which HALF of the tile (rows
    // 0-7 = pair 0, rows 8-15 = pair 1) the Gaussian's alpha >= 1/255 box (K1: ext_x, ext_y) can reach at all
    bool half0 = false, half1 = false;
    if ((uint32_t)lane < n) {
      const uint32_t gid = point_list[base + lane];
      const float4* r = records + (size_t)gid * kRecVec;
      float4 a0 = r[0], a2 = r[2];
      const float4 a3 = r[3];
      a0.x = (a0.x - tile_x0) + a3.x;       // tile-relative pixel centre, once per (tile, Gaussian)
      a0.y = (a0.y - tile_y0) + a3.y;
      const float ex = a2.z, ey = a3.w;
      const bool xok = (a0.x - ex <= 15.0f) && (a0.x + ex >= 0.0f);
      half0 = xok && (a0.y - ey <= 7.0f) && (a0.y + ey >= 0.0f);
      half1 = xok && (a0.y - ey <= 15.0f) && (a0.y + ey >= 8.0f);
      a2.z = a3.z;                           // skip threshold
      lrec[lane * kLds + 0] = a0;
      lrec[lane * kLds + 1] = r[1];
      lrec[lane * kLds + 2] = a2;
    }
    const uint64_t m0 = __ballot(half0), m1 = __ballot(half1);   // scalar registers: one bit per Gaussian of the batch
    __syncthreads();
    // only Gaussians whose box reaches the tile are visited (22 % of the instances of the benchmark scene do not:
    // their 3-sigma rectangle touches the tile, their alpha >= 1/255 region does not), and only for the half they reach
    for (uint64_t todo = m0 | m1; todo != 0; todo &= todo - 1) {
      const uint32_t j = (uint32_t)__builtin_ctzll(todo);
      const float4 q0 = lrec[j * kLds + 0];
      const float4 q1 = lrec[j * kLds + 1];
      const float4 q2 = lrec[j * kLds + 2];
      const float gyt = q0.y;
      const float dx = q0.x - flx;
      const float ax = q0.z * dx * dx;
      const float bx = q0.w * dx;
      const float thr = q2.z;
      const uint32_t idx1 = base - r0 + j + 1;
      // per flagged half: exponents of its two strips, then the exact wave-wide candidate test ("max of the pair >=
      // thr": one compare per ballot); a finished / outside pixel sits at y = kBig and is never a candidate
      if ((m0 >> j) & 1) {
        const f2 dy0 = gyt - P0.fly;
        const f2 pw0 = fma2(dy0, fma2(splat(q1.x), dy0, splat(bx)), splat(ax));
        if (__ballot(fmaxf(pw0.x, pw0.y) >= thr) != 0) fwd_pair_live<DEPTH>(P0, pw0, q1, q2, idx1);
      }
      if ((m1 >> j) & 1) {
        const f2 dy1 = gyt - P1.fly;
        const f2 pw1 = fma2(dy1, fma2(splat(q1.x), dy1, splat(bx)), splat(ax));
        if (__ballot(fmaxf(pw1.x, pw1.y) >= thr) != 0) fwd_pair_live<DEPTH>(P1, pw1, q1, q2, idx1);
      }
    }
    // every pixel saturated?  (finished pixels sit at y = kBig: the rest of a batch costs them only the
    // no-candidate path above)
    wave_done = __ballot(fminf(fminf(P0.fly.x, P0.fly.y), fminf(P1.fly.x, P1.fly.y)) < kBig) == 0;
  }

  const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
  const size_t plane = (size_t)W * H;
  const float Tf[4] = {P0.T.x, P0.T.y, P1.T.x, P1.T.y};
  const float cr[4] = {P0.Cr.x, P0.Cr.y, P1.Cr.x, P1.Cr.y};
  const float cg[4] = {P0.Cg.x, P0.Cg.y, P1.Cg.x, P1.Cg.y};
  const float cb[4] = {P0.Cb.x, P0.Cb.y, P1.Cb.x, P1.Cb.y};
  const float dd[4] = {P0.Dd.x, P0.Dd.y, P1.Dd.x, P1.Dd.y};
  const uint32_t la[4] = {P0.last0, P0.last1, P1.last0, P1.last1};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (inside[s]) {
      const size_t pix = (size_t)(py0 + 4 * s) * W + px;
      out_color[pix] = cr[s] + Tf[s] * b0;
      out_color[plane + pix] = cg[s] + Tf[s] * b1;
      out_color[2 * plane + pix] = cb[s] + Tf[s] * b2;
      if (DEPTH) out_invdepth[pix] = dd[s];
      final_T[pix] = Tf[s];
      n_contrib[pix] = la[s];
    }
  }
  // Side job (hgs_raster_args.bwd_ws_prezero): every tile clears its share of the backward's instance scratch.
  // This kernel is ALU-bound with HBM mostly idle, so the stores ride along for free; they are issued last so that
  // nothing in this wave waits for them.
  if (zero_ws) {
    const uint32_t per = (zero_vecs + (uint32_t)T - 1) / (uint32_t)T;
    const uint32_t z0 = (uint32_t)tg.tile * per;
    const uint32_t z1 = min(z0 + per, zero_vecs);
    for (uint32_t i = z0 + (uint32_t)lane; i < z1; i += 64u) zero_ws[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ---- packed backward ------------------------------------------------------------------
struct BwdPair {
  f2 fly, T, A, bgd, gd;   // A: value blended BEHIND the next Gaussian to be visited, per unit T
  const f2* pix;           // LDS: this lane's dL/dC (r, g, b) of the pair's two pixels at [0], [64], [128] (see the kernel)
  uint32_t nc0, nc1;
};
// Per-lane partial sums of one (tile, Gaussian) over the lane's four pixels, as plain floats.  All four pixels of a lane
// share x, so the three sums that carry dx (sum X dx, sum X dx^2, sum X dx dy) are formed from a0 / a1 just before the
// cross-lane reduction instead of being accumulated per pixel pair.
struct BwdSums {
  float a0, a1, a4, s5, s6, s7, s8, s9;   // sum X, sum X dy, sum X dy^2, sum G dL/dalpha, sum w dL/dC_rgb, sum w dL/dD
};

// FIRST: the sums are assigned, not accumulated (the first visited half of an instance: no zero-filled accumulators)
template <bool DEPTH, bool FIRST>
__device__ __forceinline__ void bwd_pair_live(BwdPair& p, BwdSums& S, f2 pw, f2 dy, float dx, bool c0, bool c1,
                                              const float4& q1, f2 q2) {   // q2 = (blue, 1/z)
  // exp2(min(pw, 0)) as in the forward (the exponent of a positive definite conic is <= 0 up to rounding); it also
  // keeps G finite on the non-live lanes, whose contributions are multiplied by an exact 0 below
  const f2 G = {exp2_le1(pw.x), exp2_le1(pw.y)};
  const f2 araw = q1.y * G;
  const f2 alpha = {fminf(kAlphaMax, araw.x), fminf(kAlphaMax, araw.y)};
  const bool live0 = c0 && (alpha.x >= kAlphaMin);   // = blended by the forward
  const bool live1 = c1 && (alpha.y >= kAlphaMin);
  // (no wave-level "nobody live" exit: the candidate test already is the alpha test up to its 1e-3 guard band)
  // non-live lanes take part with alpha = 0: identity for T, for the A recurrence and for every sum
  const f2 ae = {live0 ? alpha.x : 0.0f, live1 ? alpha.y : 0.0f};
  const f2 oma = 1.0f - ae;
  const f2 rinv = {__builtin_amdgcn_rcpf(oma.x), __builtin_amdgcn_rcpf(oma.y)};
  const f2 Tcur = p.T * rinv;                                   // transmittance in front of this Gaussian
  const f2 g0 = p.pix[0], g1 = p.pix[64], g2 = p.pix[128];
  f2 q = fma2(g2, splat(q2.x), fma2(g1, splat(q1.w), g0 * q1.z));
  if (DEPTH) q = fma2(p.gd, splat(q2.y), q);
  const f2 qA = q - p.A;
  const f2 dfull = fma2(qA, Tcur, -(p.bgd * rinv));
  const f2 dLda = {live0 ? dfull.x : 0.0f, live1 ? dfull.y : 0.0f};
  const f2 w = ae * Tcur;
  const f2 X = araw * dLda;                                     // dL/dpower (straight-through 0.99 cap)
  const f2 Xdy = X * dy;
  // dot products over the pair's two pixels with plain instructions (a packed product + a fold of its halves costs more)
  if (FIRST) {
    S.a0 = X.x + X.y;
    S.a1 = Xdy.x + Xdy.y;
    S.a4 = fmaf(Xdy.y, dy.y, Xdy.x * dy.x);
    S.s5 = fmaf(G.y, dLda.y, G.x * dLda.x);
    S.s6 = fmaf(w.y, g0.y, w.x * g0.x);
    S.s7 = fmaf(w.y, g1.y, w.x * g1.x);
    S.s8 = fmaf(w.y, g2.y, w.x * g2.x);
    if (DEPTH) S.s9 = fmaf(w.y, p.gd.y, w.x * p.gd.x);
  } else {
    S.a0 += X.x + X.y;
    S.a1 += Xdy.x + Xdy.y;
    S.a4 = fmaf(Xdy.y, dy.y, fmaf(Xdy.x, dy.x, S.a4));
    S.s5 = fmaf(G.y, dLda.y, fmaf(G.x, dLda.x, S.s5));
    S.s6 = fmaf(w.y, g0.y, fmaf(w.x, g0.x, S.s6));
    S.s7 = fmaf(w.y, g1.y, fmaf(w.x, g1.x, S.s7));
    S.s8 = fmaf(w.y, g2.y, fmaf(w.x, g2.x, S.s8));
    if (DEPTH) S.s9 = fmaf(w.y, p.gd.y, fmaf(w.x, p.gd.x, S.s9));
  }
  p.T = Tcur;
  p.A = fma2(ae, qA, p.A);                                      // A_(i-1) = alpha_i q_i + (1 - alpha_i) A_i
}

__device__ __forceinline__ float swap32_add(float a, float b) {   // [a.lo+a.hi | b.lo+b.hi]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float a, float b) {   // rows: [a.r0+a.r1, b.r0+b.r1, a.r2+a.r3, b.r2+b.r3]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <bool DEPTH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) void render_bwd_packed_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ bg,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const uint32_t* __restrict__ offsets, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_dinvdepth, float* __restrict__ inst, const uint32_t* __restrict__ order) {
  constexpr int BATCH = 64;
  constexpr int kLds = 3;   // float4 per staged Gaussian: (gxt,gyt,A2,B2) (C2,o,r,g) (b,1/z,emission offset,rect)
  __shared__ float4 lrec[BATCH * kLds];
  __shared__ float lthr[BATCH];          // skip threshold
  // The colour gradients of a lane's four pixels are constants that only the live path reads: they sit in LDS ([pair][r,
  // g, b][lane] as float2; written and read by the SAME lane, so no barrier), not in 12 registers -- the kernel then
  // fits 80 registers = 6 waves per SIMD (6.3 KB of LDS per wave: 24 waves per CU; same time alone, 2.5 % more
  // frames/s when the other stream's streaming kernels run beside it)
  __shared__ f2 lpix[2 * 3 * 64];

  TileGeom tg;
  if (!block_to_tile(T, gx, order, tg)) return;
  const int lane = threadIdx.x;
  const int lx = lane & 15, ly0 = lane >> 4;
  const int px = tg.tx * kTile + lx;
  const int py0 = tg.ty * kTile + ly0;
  const float flx = (float)lx;
  const float tile_x0 = (float)(tg.tx * kTile), tile_y0 = (float)(tg.ty * kTile);
  const size_t plane = (size_t)W * H;
  const float b0 = bg[0], b1 = bg[1], b2 = bg[2];

  float fly[4], Tr[4], bgd[4], g0[4], g1[4], g2[4], gd[4];
  uint32_t nc[4];
  uint32_t maxnc = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int py = py0 + 4 * s;
    fly[s] = kBig; Tr[s] = 1.0f; bgd[s] = 0.f; g0[s] = g1[s] = g2[s] = gd[s] = 0.f; nc[s] = 0;
    if (px < W && py < H) {
      const size_t pix = (size_t)py * W + px;
      fly[s] = (float)(ly0 + 4 * s);
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
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) maxnc = max(maxnc, (uint32_t)__shfl_xor((int)maxnc, off, 64));
  maxnc = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxnc);   // uniform: loop counters live in scalar registers
  if (maxnc == 0) return;
  BwdPair P0, P1;
  P0.fly = f2{fly[0], fly[1]}; P1.fly = f2{fly[2], fly[3]};
  P0.T = f2{Tr[0], Tr[1]};     P1.T = f2{Tr[2], Tr[3]};
  P0.bgd = f2{bgd[0], bgd[1]}; P1.bgd = f2{bgd[2], bgd[3]};
  P0.pix = lpix + lane; P1.pix = lpix + 3 * 64 + lane;
  lpix[0 * 64 + lane] = f2{g0[0], g0[1]}; lpix[3 * 64 + lane] = f2{g0[2], g0[3]};
  lpix[1 * 64 + lane] = f2{g1[0], g1[1]}; lpix[4 * 64 + lane] = f2{g1[2], g1[3]};
  lpix[2 * 64 + lane] = f2{g2[0], g2[1]}; lpix[5 * 64 + lane] = f2{g2[2], g2[3]};
  P0.gd = f2{gd[0], gd[1]};    P1.gd = f2{gd[2], gd[3]};
  P0.A = P1.A = splat(0.0f);
  P0.nc0 = nc[0]; P0.nc1 = nc[1]; P1.nc0 = nc[2]; P1.nc1 = nc[3];
  const uint32_t r0 = ranges[tg.tile * 2 + 0];
  // which float of the 12-float instance record this lane stores after the reduction (-1: none).  Rows carry the
  // values {0,2,1,3}[row] of each group of four (permlane swap order); within a row, quad 0 holds s0..s3's, quad 2
  // s4..s7's, quad 1 s8 (row 1) / s9 (row 3) -- see row_sum16_x3
  const int row = lane >> 4, quad = (lane >> 2) & 3;
  const int slot = ((row & 1) << 1) | (row >> 1);
  int store_k = -1;
  if ((lane & 3) == 0) {
    if (quad == 0) store_k = slot;
    else if (quad == 2) store_k = 4 + slot;
    else if (quad == 1 && (row & 1)) store_k = 8 + (row >> 1);
  }
  for (int bstart = (int)((maxnc - 1) / BATCH) * BATCH; bstart >= 0; bstart -= BATCH) {
    const int n = min(BATCH, (int)maxnc - bstart);
    __syncthreads();
    bool half0 = false, half1 = false;          // see the forward kernel: which half of the tile the Gaussian can reach
    if (lane < n) {
      const uint32_t gid = point_list[r0 + bstart + lane];
      const float4* r = records + (size_t)gid * kRecVec;
      float4 a0 = r[0], a2 = r[2];
      const float4 a3 = r[3];
      a0.x = (a0.x - tile_x0) + a3.x;           // tile-relative pixel centre, once per (tile, Gaussian)
      a0.y = (a0.y - tile_y0) + a3.y;
      const float ex = a2.z, ey = a3.w;
      const bool xok = (a0.x - ex <= 15.0f) && (a0.x + ex >= 0.0f);
      half0 = xok && (a0.y - ey <= 7.0f) && (a0.y + ey >= 0.0f);
      half1 = xok && (a0.y - ey <= 15.0f) && (a0.y + ey >= 8.0f);
      a2.z = __uint_as_float(offsets[gid]);     // emission offset of this Gaussian's instance run
      lrec[lane * kLds + 0] = a0;
      lrec[lane * kLds + 1] = r[1];
      lrec[lane * kLds + 2] = a2;
      lthr[lane] = a3.z;
    }
    const uint64_t m0 = __ballot(half0), m1 = __ballot(half1);
    __syncthreads();
    // back to front over the Gaussians whose box reaches the tile
    for (uint64_t todo = m0 | m1; todo != 0;) {
      const int j = 63 - __builtin_clzll(todo);
      todo &= ~(1ull << j);
      const uint32_t rel = (uint32_t)(bstart + j);
      const float4 q0 = lrec[j * kLds + 0];
      const float4 q1 = lrec[j * kLds + 1];
      const float gyt = q0.y;
      const float dx = q0.x - flx;
      const float ax = q0.z * dx * dx;
      const float bx = q0.w * dx;
      const float thr = lthr[j];
      // per flagged half: exponents, per-strip candidate predicates (log-domain alpha test AND "the forward blended
      // this Gaussian into the pixel", i.e. rel < n_contrib) and the wave-uniform "half has a candidate" from the
      // ballots of the plain compares combined in scalar registers
      // (read only under b0 / b1, i.e. only when assigned: left uninitialised on purpose -- a zero initialiser costs
      // four vector moves per instance on the path of a half that is not visited)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
#pragma clang diagnostic ignored "-Wconditional-uninitialized"
      f2 dy0, dy1, pw0, pw1;
      bool c0 = false, c1 = false, c2 = false, c3 = false, b0 = false, b1 = false;
      if ((m0 >> j) & 1) {
        dy0 = gyt - P0.fly;
        pw0 = fma2(dy0, fma2(splat(q1.x), dy0, splat(bx)), splat(ax));
        c0 = (pw0.x >= thr) && (rel < P0.nc0);
        c1 = (pw0.y >= thr) && (rel < P0.nc1);
        b0 = ((__ballot(pw0.x >= thr) & __ballot(rel < P0.nc0)) | (__ballot(pw0.y >= thr) & __ballot(rel < P0.nc1))) != 0;
      }
      if ((m1 >> j) & 1) {
        dy1 = gyt - P1.fly;
        pw1 = fma2(dy1, fma2(splat(q1.x), dy1, splat(bx)), splat(ax));
        c2 = (pw1.x >= thr) && (rel < P1.nc0);
        c3 = (pw1.y >= thr) && (rel < P1.nc1);
        b1 = ((__ballot(pw1.x >= thr) & __ballot(rel < P1.nc0)) | (__ballot(pw1.y >= thr) & __ballot(rel < P1.nc1))) != 0;
      }
      if (!(b0 || b1)) continue;
      // the record's third float4 in two halves: (blue, 1/z) now, for the live path; (emission offset, rectangle) next
      // to the reduction that hides its latency -- read as one float4 here, the two scalars it feeds into the store's
      // address make the compiler wait for the whole read before the live path starts
      const f2 q2 = *reinterpret_cast<const f2*>(&lrec[j * kLds + 2]);
      BwdSums S;
      if (!DEPTH) S.s9 = 0.0f;
      if (b0) {
        bwd_pair_live<DEPTH, true>(P0, S, pw0, dy0, dx, c0, c1, q1, q2);
        if (b1) bwd_pair_live<DEPTH, false>(P1, S, pw1, dy1, dx, c2, c3, q1, q2);
      } else {
        bwd_pair_live<DEPTH, true>(P1, S, pw1, dy1, dx, c2, c3, q1, q2);
      }
      {
        // 10 sums x 64 lanes -> 10 floats of the instance record: halve the lane
        // count twice with permlane32 / permlane16 swaps (two values share a register afterwards: rows of the two
        // registers = (s0,s2,s1,s3) and (s4,s6,s5,s7); s8 / s9 keep two rows each), then ONE packed row reduction of the
        // three registers (row_sum16_x3) that also joins the row pairs of s8 / s9.
        const uint2 slot = *reinterpret_cast<const uint2*>(reinterpret_cast<const float*>(&lrec[j * kLds + 2]) + 2);
        const float s0 = dx * S.a0, s3 = dx * S.a1;     // sum X dx, sum X dx dy
        const float s2 = dx * s0;                        // sum X dx^2
        const float u0 = swap32_add(s0, S.a1);
        const float u1 = swap32_add(s2, s3);
        const float u2 = swap32_add(S.a4, S.s5);
        const float u3 = swap32_add(S.s6, S.s7);
        const float u4 = swap32_add(S.s8, S.s9);
        const float v = row_sum16_x3(swap16_add(u0, u1), swap16_add(u2, u3), u4);
        __builtin_amdgcn_sched_barrier(0);              // (keep the scalar address arithmetic behind the reduction)
        if (store_k >= 0) {
          const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot.x);
          const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot.y);
          const uint32_t minx = rb & 1023u, miny = (rb >> 10) & 1023u, rw = rb >> 20;
          const uint32_t e = off + ((uint32_t)tg.ty - miny) * rw + ((uint32_t)tg.tx - minx);
          inst[(size_t)e * kInstStride + store_k] = v;     // ONE store: ten lanes, 40 of the record's 48 bytes
        }
      }
    }
  }
}

}  // namespace

int launch_render_fwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      float* out_color, float* out_invdepth, float* zero_ws, size_t zero_floats, hipStream_t s) {
  if (zero_ws && (zero_floats >> 2) > 0xfff00000ull) {   // the side job counts float4s in 32 bits
    HGS_HIP(hipMemsetAsync(zero_ws, 0, zero_floats * sizeof(float), s));
    zero_ws = nullptr;
  }
  const int gxx = grid_x(a.width), T = gxx * grid_y(a.height);
  const int nblk = ((T + 7) / 8) * 8;
  const bool depth = a.do_depth && out_invdepth;
  auto kern = depth ? render_fwd_packed_kernel<true> : render_fwd_packed_kernel<false>;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, s, b.ranges, b.vals_out,
                     reinterpret_cast<const float4*>(g.records), a.width, a.height, gxx, T, a.bg, out_color,
                     out_invdepth, im.final_T, im.n_contrib, reinterpret_cast<float4*>(zero_ws),
                     (uint32_t)(zero_floats >> 2), b.tile_order);
  HGS_LAUNCH_CHECK("render_fwd_packed", s, a.debug);
  return HGS_OK;
}

int launch_render_bwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      const float* out_color, const float* out_invdepth, const float* dL_dcolor,
                      const float* dL_dinvdepth, float* inst_grads, hipStream_t s) {
  (void)out_color;
  const int gxx = grid_x(a.width), T = gxx * grid_y(a.height);
  const int nblk = ((T + 7) / 8) * 8;
  const bool depth = a.do_depth && out_invdepth && dL_dinvdepth;
  auto kern = depth ? render_bwd_packed_kernel<true> : render_bwd_packed_kernel<false>;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, s, b.ranges, b.vals_out,
                     reinterpret_cast<const float4*>(g.records), a.width, a.height, gxx, T, a.bg, im.final_T,
                     im.n_contrib, g.offsets, dL_dcolor, dL_dinvdepth, inst_grads, b.tile_order);
  HGS_LAUNCH_CHECK("render_bwd_packed", s, a.debug);
  return HGS_OK;
}

}  // namespace hgs
