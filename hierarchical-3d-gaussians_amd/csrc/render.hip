// K6 (tile compositing, forward) and K7 (its backward).
//
// Restates per-tile front-to-back alpha compositing with an inverse-depth channel
// (op call sites gaussian_renderer/__init__.py:105-113,269-277;
// backward entered from train_single.py:123 / train_post.py:142).
//
// CDNA4 design (not the CUDA 256-threads-one-pixel-each shape):
//  * a 16x16 tile is cut into four 16x4 STRIPS; a wave64 owns S strips, i.e. a lane owns S
//    pixels (same x, y = y0 + 4*s).  Default S=4: ONE WAVE RENDERS A WHOLE TILE.  The
//    per-Gaussian record is read from LDS once per tile instead of once per strip (LDS
//    broadcast bandwidth, shared by the CU's four SIMDs, is the binding resource at one pixel
//    per lane: measured 0.53 / 0.37 / 0.35 ms forward for S = 1 / 2 / 4 at 1080p, 1 M
//    Gaussians), no workgroup barrier is needed, and the backward's cross-lane reduction runs
//    once per (tile, Gaussian).
//  * dead-pair skipping: alpha >= 1/255  <=>  power2 >= log2(1/255) - log2(opacity).  A strip
//    whose 64 pixels are all below that threshold (minus a 1e-3 guard band) is skipped after
//    5 VALU instructions, before the exp; the exact alpha test still takes every borderline
//    decision, so results do not depend on the pre-test.
//  * pixel coordinates are TILE-RELATIVE: the record carries the pixel centre as hi + lo
//    floats from K1's double-precision projection; (hi - tile_origin) + lo is exact to ~1e-6 px
//    at any resolution (absolute float32 coordinates carry 6e-5 px of error at x ~ 1900).
//  * "done" lives in the sign of T (T < 0 <=> saturated, |T| = final transmittance).
//  * backward: BACK-TO-FRONT from the forward's last contributor with
//        T_i = T_{i+1} / (1 - alpha_i),   A_i = alpha_{i+1} q_{i+1} + (1 - alpha_{i+1}) A_{i+1},
//        dL/dalpha_i = (q_i - A_i) T_i - T_final (dL/dC . bg) / (1 - alpha_i),   q = dL/dC . c + dL/dD / z
//    (every term relatively accurate; a front-to-back variant using V - prefix was measured 10-80x
//    less accurate on pixels with capped alphas and was dropped).
//  * the 10 per-(tile,Gaussian) partial sums are reduced over the 4 strips in registers, over the
//    wave with DPP row reductions + row_bcast, and lane 63 stores the 48-byte instance record to
//    its EMISSION slot; K8 sums each Gaussian's contiguous run.  No LDS accumulator, no atomics
//    of any kind, bit-reproducible.
#include "common.h"

namespace hgs {
namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTEps = 0.0001f;
constexpr float kLog2AlphaMin = -7.994353436858858f;   // log2(1/255)
constexpr float kSkipGuard = 1.0e-3f;

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
// wave total in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = row_sum16(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));  // row_bcast:15
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));  // row_bcast:31
  return v;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct TileGeom {
  int tile, tx, ty;
};

// XCD-aware block -> tile map: consecutive blocks land on different XCDs (b % 8), so give
// every XCD a contiguous band of tiles; neighbouring tiles share Gaussians and therefore
// share that XCD's L2.
__device__ __forceinline__ bool block_to_tile(int T, int gx, TileGeom& tg) {
  const int per = (T + 7) >> 3;
  const int b = blockIdx.x;
  const int tile = (b & 7) * per + (b >> 3);
  if (tile >= T) return false;
  tg.tile = tile;
  tg.ty = tile / gx;
  tg.tx = tile - tg.ty * gx;
  return true;
}

// --------------------------------------------------------------------------------
// forward
// --------------------------------------------------------------------------------
template <int S, bool DEPTH>
__global__ __launch_bounds__(64 * (4 / S)) void render_fwd_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ bg,
    float* __restrict__ out_color, float* __restrict__ out_invdepth, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib) {
  constexpr int NW = 4 / S;
  constexpr int BATCH = 64 * NW;
  __shared__ float4 lrec[BATCH * kRecVec];

  TileGeom tg;
  if (!block_to_tile(T, gx, tg)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = lane & 15;
  const int ly0 = wave * (4 * S) + (lane >> 4);
  const int px = tg.tx * kTile + lx;
  const int py0 = tg.ty * kTile + ly0;
  const float flx = (float)lx;
  const float tile_x0 = (float)(tg.tx * kTile), tile_y0 = (float)(tg.ty * kTile);

  float fly[S], Tr[S], Cr[S], Cg[S], Cb[S], Dd[S];
  uint32_t last[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    fly[s] = (float)(ly0 + 4 * s);
    Tr[s] = (px < W && py0 + 4 * s < H) ? 1.0f : -1.0f;
    Cr[s] = 0.f; Cg[s] = 0.f; Cb[s] = 0.f; Dd[s] = 0.f;
    last[s] = 0;
  }
  const uint32_t r0 = ranges[tg.tile * 2 + 0], r1 = ranges[tg.tile * 2 + 1];
  bool wave_done;
  {
    bool any = false;
#pragma unroll
    for (int s = 0; s < S; ++s) any = any || (Tr[s] > 0.0f);
    wave_done = (__ballot(any) == 0);
  }

  for (uint32_t base = r0; base < r1; base += BATCH) {
    if (NW == 1) {
      if (wave_done) break;
    } else {
      if (__syncthreads_and(wave_done)) break;
    }
    const uint32_t n = min((uint32_t)BATCH, r1 - base);
    if ((uint32_t)tid < n) {
      const uint32_t gid = point_list[base + tid];
      const float4* r = records + (size_t)gid * kRecVec;
      lrec[tid * kRecVec + 0] = r[0];
      lrec[tid * kRecVec + 1] = r[1];
      lrec[tid * kRecVec + 2] = r[2];
      lrec[tid * kRecVec + 3] = r[3];
    }
    __syncthreads();
    if (!wave_done) {
      for (uint32_t j = 0; j < n; ++j) {
        const float4 q0 = lrec[j * kRecVec + 0];
        const float4 q1 = lrec[j * kRecVec + 1];
        const float4 q2 = lrec[j * kRecVec + 2];
        const float4 q3 = lrec[j * kRecVec + 3];
        const float gxt = (q0.x - tile_x0) + q3.x;      // tile-relative pixel centre
        const float gyt = (q0.y - tile_y0) + q3.y;
        const float dx = gxt - flx;
        const float ax = q0.z * dx * dx;
        const float bx = q0.w * dx;
        const float thr = (kLog2AlphaMin - kSkipGuard) - __builtin_amdgcn_logf(q1.y);
        bool any = false;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float dy = gyt - fly[s];
          const float pw = ax + dy * (q1.x * dy + bx);
          const bool cand = (pw >= thr) && (Tr[s] > 0.0f);
          if (__ballot(cand) != 0) {
            const float G = fast_exp2(pw);
            const float araw = q1.y * G;
            const float alpha = fminf(kAlphaMax, araw);
            const bool live = cand && (pw <= 0.0f) && (alpha >= kAlphaMin);
            const float Tn = Tr[s] * (1.0f - alpha);
            const bool stop = live && (Tn < kTEps);
            const bool blend = live && !stop;
            const float w = blend ? alpha * Tr[s] : 0.0f;
            Cr[s] += w * q1.z;
            Cg[s] += w * q1.w;
            Cb[s] += w * q2.x;
            if (DEPTH) Dd[s] += w * q2.y;
            Tr[s] = stop ? -Tr[s] : (blend ? Tn : Tr[s]);
            last[s] = blend ? (base - r0 + j + 1) : last[s];
          }
          any = any || (Tr[s] > 0.0f);
        }
        if (__ballot(any) == 0) {
          wave_done = true;
          break;
        }
      }
    }
  }

  const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
  const size_t plane = (size_t)W * H;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int py = py0 + 4 * s;
    if (px < W && py < H) {
      const size_t pix = (size_t)py * W + px;
      const float Tf = fabsf(Tr[s]);
      out_color[pix] = Cr[s] + Tf * b0;
      out_color[plane + pix] = Cg[s] + Tf * b1;
      out_color[2 * plane + pix] = Cb[s] + Tf * b2;
      if (DEPTH) out_invdepth[pix] = Dd[s];
      final_T[pix] = Tf;
      n_contrib[pix] = last[s];
    }
  }
}

// --------------------------------------------------------------------------------
// backward (back-to-front, one wave per tile)
// --------------------------------------------------------------------------------
template <bool DEPTH>
__global__ __launch_bounds__(64) void render_bwd_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ bg,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const uint32_t* __restrict__ offsets, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_dinvdepth, float4* __restrict__ inst) {
  constexpr int S = 4;
  constexpr int BATCH = 64;
  __shared__ float4 lrec[BATCH * kRecVec];

  TileGeom tg;
  if (!block_to_tile(T, gx, tg)) return;
  const int lane = threadIdx.x;
  const int lx = lane & 15, ly0 = lane >> 4;
  const int px = tg.tx * kTile + lx;
  const int py0 = tg.ty * kTile + ly0;
  const float flx = (float)lx;
  const float tile_x0 = (float)(tg.tx * kTile), tile_y0 = (float)(tg.ty * kTile);
  const size_t plane = (size_t)W * H;
  const float b0 = bg[0], b1 = bg[1], b2 = bg[2];

  float fly[S], Tr[S], Aq[S], la[S], lq[S], bgd[S], g0[S], g1[S], g2[S], gd[S];
  uint32_t nc[S];
  uint32_t maxnc = 0;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int py = py0 + 4 * s;
    fly[s] = (float)(ly0 + 4 * s);
    Tr[s] = 1.0f; Aq[s] = 0.f; la[s] = 0.f; lq[s] = 0.f; bgd[s] = 0.f;
    g0[s] = g1[s] = g2[s] = gd[s] = 0.f;
    nc[s] = 0;
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
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) maxnc = max(maxnc, (uint32_t)__shfl_xor((int)maxnc, off, 64));
  if (maxnc == 0) return;
  const uint32_t r0 = ranges[tg.tile * 2 + 0];

  for (int bstart = (int)((maxnc - 1) / BATCH) * BATCH; bstart >= 0; bstart -= BATCH) {
    const int n = min(BATCH, (int)maxnc - bstart);
    __syncthreads();
    if (lane < n) {
      const uint32_t gid = point_list[r0 + bstart + lane];
      const float4* r = records + (size_t)gid * kRecVec;
      float4 r2 = r[2];
      r2.z = __uint_as_float(offsets[gid]);     // emission offset of this Gaussian's instance run
      lrec[lane * kRecVec + 0] = r[0];
      lrec[lane * kRecVec + 1] = r[1];
      lrec[lane * kRecVec + 2] = r2;
      lrec[lane * kRecVec + 3] = r[3];
    }
    __syncthreads();
    for (int j = n - 1; j >= 0; --j) {
      const uint32_t rel = (uint32_t)(bstart + j);
      const float4 q0 = lrec[j * kRecVec + 0];
      const float4 q1 = lrec[j * kRecVec + 1];
      const float4 q2 = lrec[j * kRecVec + 2];
      const float4 q3 = lrec[j * kRecVec + 3];
      const float gxt = (q0.x - tile_x0) + q3.x;
      const float gyt = (q0.y - tile_y0) + q3.y;
      const float dx = gxt - flx;
      const float ax = q0.z * dx * dx;
      const float bx = q0.w * dx;
      const float thr = (kLog2AlphaMin - kSkipGuard) - __builtin_amdgcn_logf(q1.y);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f, s8 = 0.f, s9 = 0.f;
      uint64_t any_blend = 0;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float dy = gyt - fly[s];
        const float pw = ax + dy * (q1.x * dy + bx);
        const bool cand = (pw >= thr) && (rel < nc[s]);
        if (__ballot(cand) != 0) {
          const float G = fast_exp2(pw);
          const float araw = q1.y * G;
          const float alpha = fminf(kAlphaMax, araw);
          const bool live = cand && (pw <= 0.0f) && (alpha >= kAlphaMin);   // blended by the forward
          const uint64_t lm = __ballot(live);
          if (lm != 0) {
            any_blend |= lm;
            const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
            const float Tcur = Tr[s] * rinv;                       // transmittance in front of this Gaussian
            const float An = Aq[s] + la[s] * (lq[s] - Aq[s]);      // value blended behind it, per unit T
            float q = g0[s] * q1.z + g1[s] * q1.w + g2[s] * q2.x;
            if (DEPTH) q += gd[s] * q2.y;
            const float dLda = live ? ((q - An) * Tcur - bgd[s] * rinv) : 0.0f;
            const float w = live ? alpha * Tcur : 0.0f;
            const float X = araw * dLda;                           // dL/dpower (straight-through 0.99 cap)
            const float Xdx = X * dx, Xdy = X * dy;
            s0 += Xdx;
            s1 += Xdy;
            s2 += Xdx * dx;
            s3 += Xdx * dy;
            s4 += Xdy * dy;
            s5 += G * dLda;
            s6 += w * g0[s];
            s7 += w * g1[s];
            s8 += w * g2[s];
            if (DEPTH) s9 += w * gd[s];
            Tr[s] = live ? Tcur : Tr[s];
            Aq[s] = live ? An : Aq[s];
            la[s] = live ? alpha : la[s];
            lq[s] = live ? q : lq[s];
          }
        }
      }
      if (any_blend != 0) {   // wave-uniform
        s0 = wave_sum_to_lane63(s0); s1 = wave_sum_to_lane63(s1); s2 = wave_sum_to_lane63(s2);
        s3 = wave_sum_to_lane63(s3); s4 = wave_sum_to_lane63(s4); s5 = wave_sum_to_lane63(s5);
        s6 = wave_sum_to_lane63(s6); s7 = wave_sum_to_lane63(s7); s8 = wave_sum_to_lane63(s8);
        if (DEPTH) s9 = wave_sum_to_lane63(s9);
        if (lane == 63) {
          const uint32_t off = __float_as_uint(q2.z);
          const uint32_t rb = __float_as_uint(q2.w);
          const uint32_t minx = rb & 1023u, miny = (rb >> 10) & 1023u, rw = rb >> 20;
          const uint32_t e = off + ((uint32_t)tg.ty - miny) * rw + ((uint32_t)tg.tx - minx);
          float4* dst = inst + (size_t)e * 3;
          dst[0] = make_float4(s0, s1, s2, s3);
          dst[1] = make_float4(s4, s5, s6, s7);
          dst[2] = make_float4(s8, s9, 0.f, 0.f);
        }
      }
    }
  }
}

}  // namespace

template <int S>
static int launch_fwd_s(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                        float* out_color, float* out_invdepth, hipStream_t s) {
  const int gxx = grid_x(a.width), T = gxx * grid_y(a.height);
  const int nblk = ((T + 7) / 8) * 8;
  const dim3 block(64 * (4 / S));
  const float4* rec = reinterpret_cast<const float4*>(g.records);
  const bool depth = a.do_depth && out_invdepth;
  auto kern = depth ? render_fwd_kernel<S, true> : render_fwd_kernel<S, false>;
  hipLaunchKernelGGL(kern, dim3(nblk), block, 0, s, b.ranges, b.vals_out, rec, a.width, a.height, gxx, T, a.bg,
                     out_color, out_invdepth, im.final_T, im.n_contrib);
  HGS_LAUNCH_CHECK("render_fwd", s, a.debug);
  return HGS_OK;
}

// variant: 0 (default) = one wave per tile (S=4); 1 / 2 = four / two waves per tile (S=1 / S=2),
// kept for A/B profiling of the strip layout.  The backward has a single implementation.
int launch_render_fwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      float* out_color, float* out_invdepth, hipStream_t s) {
  switch (a.variant) {
    case 1: return launch_fwd_s<1>(a, g, b, im, out_color, out_invdepth, s);
    case 2: return launch_fwd_s<2>(a, g, b, im, out_color, out_invdepth, s);
    default: return launch_fwd_s<4>(a, g, b, im, out_color, out_invdepth, s);
  }
}

int launch_render_bwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      const float* out_color, const float* out_invdepth, const float* dL_dcolor,
                      const float* dL_dinvdepth, float* inst_grads, hipStream_t s) {
  (void)out_color;
  const int gxx = grid_x(a.width), T = gxx * grid_y(a.height);
  const int nblk = ((T + 7) / 8) * 8;
  const bool depth = a.do_depth && out_invdepth && dL_dinvdepth;
  auto kern = depth ? render_bwd_kernel<true> : render_bwd_kernel<false>;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, s, b.ranges, b.vals_out,
                     reinterpret_cast<const float4*>(g.records), a.width, a.height, gxx, T, a.bg, im.final_T,
                     im.n_contrib, g.offsets, dL_dcolor, dL_dinvdepth, reinterpret_cast<float4*>(inst_grads));
  HGS_LAUNCH_CHECK("render_bwd", s, a.debug);
  return HGS_OK;
}

}  // namespace hgs
