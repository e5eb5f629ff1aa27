// K6 (tile compositing, forward) and K7 (its backward).
//
// Restates per-tile front-to-back alpha compositing with an inverse-depth channel
// (op call sites gaussian_renderer/__init__.py:105-113,269-277;
// backward entered from train_single.py:123 / train_post.py:142).
//
// CDNA4 design (not the CUDA 256-threads-one-pixel-each shape):
//  * a 16x16 tile is cut into four 16x4 STRIPS; a wave64 owns S strips, i.e. a lane
//    owns S pixels (same x, y = y0 + 4*s).  With S=4 one wave renders a whole tile: the
//    per-Gaussian record is fetched from LDS once per tile instead of once per strip
//    (LDS broadcast bandwidth is the binding resource at one pixel per lane), and the
//    backward's cross-lane reduction runs once per (tile, Gaussian).
//  * dead-pair skipping with wave ballots: a strip whose 64 pixels all fail the
//    alpha >= 1/255 test (about 90 % of tile x Gaussian pairs in the benchmark scene)
//    costs ~12 VALU instructions and no blending work.
//  * the backward walks FRONT-TO-BACK like the forward: with V = dL/dC . C_out + dL/dD * D_out
//    known per pixel, dL/dalpha_i = T_i q_i - (V - sum_{j<=i} q_j w_j) / (1 - alpha_i), so
//    transmittance is recomputed by the same multiplications as the forward (no division
//    chain, no n_contrib replay) and early termination is identical.
//  * per-(tile,Gaussian) partial sums are reduced across the wave with DPP row
//    reductions and stored ONCE per instance into an emission-ordered scratch buffer;
//    the preprocess backward then sums each Gaussian's contiguous run.  No global
//    atomics, deterministic for S=4.
#include "common.h"

namespace hgs {
namespace {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTEps = 0.0001f;

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
  __shared__ float4 lrec[BATCH * 3];

  TileGeom tg;
  if (!block_to_tile(T, gx, tg)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = tg.tx * kTile + (lane & 15);
  const int py0 = tg.ty * kTile + wave * (4 * S) + (lane >> 4);
  const float fpx = (float)px;

  float fpy[S], Tr[S], Cr[S], Cg[S], Cb[S], Dd[S];
  bool done[S];
  uint32_t last[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int py = py0 + 4 * s;
    fpy[s] = (float)py;
    Tr[s] = 1.0f; Cr[s] = 0.f; Cg[s] = 0.f; Cb[s] = 0.f; Dd[s] = 0.f;
    done[s] = !(px < W && py < H);
    last[s] = 0;
  }
  const uint32_t r0 = ranges[tg.tile * 2 + 0], r1 = ranges[tg.tile * 2 + 1];
  bool wave_done = false;
  {
    bool all = true;
#pragma unroll
    for (int s = 0; s < S; ++s) all = all && done[s];
    wave_done = (__ballot(!all) == 0);
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
      const float4* r = records + (size_t)gid * 3;
      lrec[tid * 3 + 0] = r[0];
      lrec[tid * 3 + 1] = r[1];
      lrec[tid * 3 + 2] = r[2];
    }
    __syncthreads();
    if (!wave_done) {
      for (uint32_t j = 0; j < n; ++j) {
        const float4 q0 = lrec[j * 3 + 0];
        const float4 q1 = lrec[j * 3 + 1];
        const float4 q2 = lrec[j * 3 + 2];
        const float dx = q0.x - fpx;
        const float ax = q0.z * dx * dx;
        const float bx = q0.w * dx;
        bool all = true;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float dy = q0.y - fpy[s];
          const float pw = ax + dy * (q1.x * dy + bx);
          const float G = fast_exp2(pw);
          const float alpha = fminf(kAlphaMax, q1.y * G);
          bool live = (pw <= 0.0f) && (alpha >= kAlphaMin) && !done[s];
          if (__ballot(live) != 0) {
            const float Tn = Tr[s] * (1.0f - alpha);
            const bool stop = live && (Tn < kTEps);
            done[s] = done[s] || stop;
            live = live && !stop;
            const float w = live ? alpha * Tr[s] : 0.0f;
            Cr[s] += w * q1.z;
            Cg[s] += w * q1.w;
            Cb[s] += w * q2.x;
            if (DEPTH) Dd[s] += w * q2.y;
            Tr[s] = live ? Tn : Tr[s];
            last[s] = live ? (base - r0 + j + 1) : last[s];
          }
          all = all && done[s];
        }
        if (__ballot(!all) == 0) {
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
      out_color[pix] = Cr[s] + Tr[s] * b0;
      out_color[plane + pix] = Cg[s] + Tr[s] * b1;
      out_color[2 * plane + pix] = Cb[s] + Tr[s] * b2;
      if (DEPTH) out_invdepth[pix] = Dd[s];
      final_T[pix] = Tr[s];
      n_contrib[pix] = last[s];
    }
  }
}

// --------------------------------------------------------------------------------
// backward
// --------------------------------------------------------------------------------
template <int S, bool DEPTH>
__global__ __launch_bounds__(64 * (4 / S)) void render_bwd_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ out_color,
    const float* __restrict__ out_invdepth, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_dinvdepth, float4* __restrict__ inst) {
  constexpr int NW = 4 / S;
  constexpr int BATCH = 64 * NW;
  __shared__ float4 lrec[BATCH * 3];
  __shared__ float lacc[BATCH * kInstStride];

  TileGeom tg;
  if (!block_to_tile(T, gx, tg)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = tg.tx * kTile + (lane & 15);
  const int py0 = tg.ty * kTile + wave * (4 * S) + (lane >> 4);
  const float fpx = (float)px;
  const size_t plane = (size_t)W * H;

  float fpy[S], Tr[S], Pacc[S], V[S], g0[S], g1[S], g2[S], gd[S];
  bool done[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int py = py0 + 4 * s;
    fpy[s] = (float)py;
    Tr[s] = 1.0f;
    Pacc[s] = 0.f;
    const bool inside = (px < W && py < H);
    done[s] = !inside;
    g0[s] = g1[s] = g2[s] = gd[s] = 0.f;
    V[s] = 0.f;
    if (inside) {
      const size_t pix = (size_t)py * W + px;
      g0[s] = dL_dcolor[pix];
      g1[s] = dL_dcolor[plane + pix];
      g2[s] = dL_dcolor[2 * plane + pix];
      V[s] = g0[s] * out_color[pix] + g1[s] * out_color[plane + pix] + g2[s] * out_color[2 * plane + pix];
      if (DEPTH) {
        gd[s] = dL_dinvdepth[pix];
        V[s] += gd[s] * out_invdepth[pix];
      }
    }
  }
  const uint32_t r0 = ranges[tg.tile * 2 + 0], r1 = ranges[tg.tile * 2 + 1];
  bool wave_done = false;
  {
    bool all = true;
#pragma unroll
    for (int s = 0; s < S; ++s) all = all && done[s];
    wave_done = (__ballot(!all) == 0);
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
      const float4* r = records + (size_t)gid * 3;
      lrec[tid * 3 + 0] = r[0];
      lrec[tid * 3 + 1] = r[1];
      lrec[tid * 3 + 2] = r[2];
    }
    {
      float4* z = reinterpret_cast<float4*>(lacc) + tid * 3;
      z[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      z[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (!wave_done) {
      for (uint32_t j = 0; j < n; ++j) {
        const float4 q0 = lrec[j * 3 + 0];
        const float4 q1 = lrec[j * 3 + 1];
        const float4 q2 = lrec[j * 3 + 2];
        const float dx = q0.x - fpx;
        const float ax = q0.z * dx * dx;
        const float bx = q0.w * dx;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f, s8 = 0.f, s9 = 0.f;
        uint64_t any_live = 0;   // wave-uniform (SGPR) mask of lanes that blended this Gaussian
        bool all = true;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float dy = q0.y - fpy[s];
          const float pw = ax + dy * (q1.x * dy + bx);
          const float G = fast_exp2(pw);
          const float araw = q1.y * G;
          const float alpha = fminf(kAlphaMax, araw);
          bool live = (pw <= 0.0f) && (alpha >= kAlphaMin) && !done[s];
          if (__ballot(live) != 0) {
            const float oma = 1.0f - alpha;
            const float Tn = Tr[s] * oma;
            const bool stop = live && (Tn < kTEps);
            done[s] = done[s] || stop;
            live = live && !stop;
            const uint64_t lm = __ballot(live);
            if (lm != 0) {
              any_live |= lm;
              const float w = live ? alpha * Tr[s] : 0.0f;
              float q = g0[s] * q1.z + g1[s] * q1.w + g2[s] * q2.x;
              if (DEPTH) q += gd[s] * q2.y;
              Pacc[s] += q * w;
              const float dLda = live ? (Tr[s] * q - (V[s] - Pacc[s]) * __builtin_amdgcn_rcpf(oma)) : 0.0f;
              const float X = live ? araw * dLda : 0.0f;    // dL/dpower (straight-through 0.99 cap)
              const float Xdx = X * dx, Xdy = X * dy;
              s0 += Xdx;
              s1 += Xdy;
              s2 += Xdx * dx;
              s3 += Xdx * dy;
              s4 += Xdy * dy;
              s5 += live ? G * dLda : 0.0f;
              s6 += w * g0[s];
              s7 += w * g1[s];
              s8 += w * g2[s];
              if (DEPTH) s9 += w * gd[s];
              Tr[s] = live ? Tn : Tr[s];
            }
          }
          all = all && done[s];
        }
        if (any_live != 0) {   // wave-uniform
          s0 = row_sum16(s0); s1 = row_sum16(s1); s2 = row_sum16(s2); s3 = row_sum16(s3); s4 = row_sum16(s4);
          s5 = row_sum16(s5); s6 = row_sum16(s6); s7 = row_sum16(s7); s8 = row_sum16(s8);
          if (DEPTH) s9 = row_sum16(s9);
          if ((lane & 15) == 0) {
            float* acc = lacc + j * kInstStride;
            atomicAdd(acc + 0, s0); atomicAdd(acc + 1, s1); atomicAdd(acc + 2, s2); atomicAdd(acc + 3, s3);
            atomicAdd(acc + 4, s4); atomicAdd(acc + 5, s5); atomicAdd(acc + 6, s6); atomicAdd(acc + 7, s7);
            atomicAdd(acc + 8, s8);
            if (DEPTH) atomicAdd(acc + 9, s9);
            if (lane == 0) acc[11] = 1.0f;
          }
        }
        if (__ballot(!all) == 0) {
          wave_done = true;
          break;
        }
      }
    }
    __syncthreads();
    if ((uint32_t)tid < n) {
      const float4* a4 = reinterpret_cast<const float4*>(lacc) + tid * 3;
      const float4 v2 = a4[2];
      if (v2.w != 0.0f) {
        const float4 my2 = lrec[tid * 3 + 2];
        const uint32_t off = __float_as_uint(my2.z);
        const uint32_t rb = __float_as_uint(my2.w);
        const uint32_t minx = rb & 1023u, miny = (rb >> 10) & 1023u, rw = rb >> 20;
        const uint32_t e = off + ((uint32_t)tg.ty - miny) * rw + ((uint32_t)tg.tx - minx);
        float4* dst = inst + (size_t)e * 3;
        dst[0] = a4[0];
        dst[1] = a4[1];
        dst[2] = make_float4(v2.x, v2.y, 0.f, 0.f);
      }
    }
  }
}


// ================================================================================
// v2 kernels: same arithmetic and the same discrete decisions as the kernels above, cheaper
// dead path.
//  * "done" is carried in the sign of T (T < 0 <=> pixel saturated; |T| is the final
//    transmittance), so liveness is one compare instead of a separate flag register;
//  * log-domain pre-test: alpha >= 1/255  <=>  power2 >= log2(1/255) - log2(opacity); a strip is
//    skipped before the exp when every lane is below that threshold minus a 1e-3 guard band, so
//    the exact test (identical to v1) still takes every borderline decision.
// ================================================================================
constexpr float kLog2AlphaMin = -7.994353436858858f;   // log2(1/255)
constexpr float kSkipGuard = 1.0e-3f;

template <int S, bool DEPTH>
__global__ __launch_bounds__(64 * (4 / S)) void render_fwd2_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ bg,
    float* __restrict__ out_color, float* __restrict__ out_invdepth, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib) {
  constexpr int NW = 4 / S;
  constexpr int BATCH = 64 * NW;
  __shared__ float4 lrec[BATCH * 3];

  TileGeom tg;
  if (!block_to_tile(T, gx, tg)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = tg.tx * kTile + (lane & 15);
  const int py0 = tg.ty * kTile + wave * (4 * S) + (lane >> 4);
  const float fpx = (float)px;

  float fpy[S], Tr[S], Cr[S], Cg[S], Cb[S], Dd[S];
  uint32_t last[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int py = py0 + 4 * s;
    fpy[s] = (float)py;
    Tr[s] = (px < W && py < H) ? 1.0f : -1.0f;
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
      const float4* r = records + (size_t)gid * 3;
      lrec[tid * 3 + 0] = r[0];
      lrec[tid * 3 + 1] = r[1];
      lrec[tid * 3 + 2] = r[2];
    }
    __syncthreads();
    if (!wave_done) {
      for (uint32_t j = 0; j < n; ++j) {
        const float4 q0 = lrec[j * 3 + 0];
        const float4 q1 = lrec[j * 3 + 1];
        const float4 q2 = lrec[j * 3 + 2];
        const float dx = q0.x - fpx;
        const float ax = q0.z * dx * dx;
        const float bx = q0.w * dx;
        const float thr = (kLog2AlphaMin - kSkipGuard) - __builtin_amdgcn_logf(q1.y);
        bool any = false;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float dy = q0.y - fpy[s];
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

template <int S, bool DEPTH>
__global__ __launch_bounds__(64 * (4 / S)) void render_bwd2_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ out_color,
    const float* __restrict__ out_invdepth, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_dinvdepth, float4* __restrict__ inst) {
  constexpr int NW = 4 / S;
  constexpr int BATCH = 64 * NW;
  __shared__ float4 lrec[BATCH * 3];
  __shared__ float lacc[BATCH * kInstStride];

  TileGeom tg;
  if (!block_to_tile(T, gx, tg)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = tg.tx * kTile + (lane & 15);
  const int py0 = tg.ty * kTile + wave * (4 * S) + (lane >> 4);
  const float fpx = (float)px;
  const size_t plane = (size_t)W * H;

  float fpy[S], Tr[S], Rr[S], g0[S], g1[S], g2[S], gd[S];   // Rr = V - prefix (remaining "behind" value)
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int py = py0 + 4 * s;
    fpy[s] = (float)py;
    const bool inside = (px < W && py < H);
    Tr[s] = inside ? 1.0f : -1.0f;
    g0[s] = g1[s] = g2[s] = gd[s] = 0.f;
    Rr[s] = 0.f;
    if (inside) {
      const size_t pix = (size_t)py * W + px;
      g0[s] = dL_dcolor[pix];
      g1[s] = dL_dcolor[plane + pix];
      g2[s] = dL_dcolor[2 * plane + pix];
      Rr[s] = g0[s] * out_color[pix] + g1[s] * out_color[plane + pix] + g2[s] * out_color[2 * plane + pix];
      if (DEPTH) {
        gd[s] = dL_dinvdepth[pix];
        Rr[s] += gd[s] * out_invdepth[pix];
      }
    }
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
      const float4* r = records + (size_t)gid * 3;
      lrec[tid * 3 + 0] = r[0];
      lrec[tid * 3 + 1] = r[1];
      lrec[tid * 3 + 2] = r[2];
    }
    {
      float4* z = reinterpret_cast<float4*>(lacc) + tid * 3;
      z[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      z[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (!wave_done) {
      for (uint32_t j = 0; j < n; ++j) {
        const float4 q0 = lrec[j * 3 + 0];
        const float4 q1 = lrec[j * 3 + 1];
        const float4 q2 = lrec[j * 3 + 2];
        const float dx = q0.x - fpx;
        const float ax = q0.z * dx * dx;
        const float bx = q0.w * dx;
        const float thr = (kLog2AlphaMin - kSkipGuard) - __builtin_amdgcn_logf(q1.y);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f, s8 = 0.f, s9 = 0.f;
        uint64_t any_blend = 0;
        bool any = false;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float dy = q0.y - fpy[s];
          const float pw = ax + dy * (q1.x * dy + bx);
          const bool cand = (pw >= thr) && (Tr[s] > 0.0f);
          if (__ballot(cand) != 0) {
            const float G = fast_exp2(pw);
            const float araw = q1.y * G;
            const float alpha = fminf(kAlphaMax, araw);
            const bool live = cand && (pw <= 0.0f) && (alpha >= kAlphaMin);
            const float oma = 1.0f - alpha;
            const float Tn = Tr[s] * oma;
            const bool stop = live && (Tn < kTEps);
            const bool blend = live && !stop;
            const uint64_t bm = __ballot(blend);
            if (bm != 0) {
              any_blend |= bm;
              const float w = blend ? alpha * Tr[s] : 0.0f;
              float q = g0[s] * q1.z + g1[s] * q1.w + g2[s] * q2.x;
              if (DEPTH) q += gd[s] * q2.y;
              Rr[s] -= q * w;
              const float dLda = blend ? (Tr[s] * q - Rr[s] * __builtin_amdgcn_rcpf(oma)) : 0.0f;
              const float X = araw * dLda;             // dL/dpower (straight-through 0.99 cap); 0 when !blend
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
            }
            Tr[s] = stop ? -Tr[s] : (blend ? Tn : Tr[s]);
          }
          any = any || (Tr[s] > 0.0f);
        }
        if (any_blend != 0) {   // wave-uniform
          s0 = row_sum16(s0); s1 = row_sum16(s1); s2 = row_sum16(s2); s3 = row_sum16(s3); s4 = row_sum16(s4);
          s5 = row_sum16(s5); s6 = row_sum16(s6); s7 = row_sum16(s7); s8 = row_sum16(s8);
          if (DEPTH) s9 = row_sum16(s9);
          if ((lane & 15) == 0) {
            float* acc = lacc + j * kInstStride;
            atomicAdd(acc + 0, s0); atomicAdd(acc + 1, s1); atomicAdd(acc + 2, s2); atomicAdd(acc + 3, s3);
            atomicAdd(acc + 4, s4); atomicAdd(acc + 5, s5); atomicAdd(acc + 6, s6); atomicAdd(acc + 7, s7);
            atomicAdd(acc + 8, s8);
            if (DEPTH) atomicAdd(acc + 9, s9);
            if (lane == 0) acc[11] = 1.0f;
          }
        }
        if (__ballot(any) == 0) {
          wave_done = true;
          break;
        }
      }
    }
    __syncthreads();
    if ((uint32_t)tid < n) {
      const float4* a4 = reinterpret_cast<const float4*>(lacc) + tid * 3;
      const float4 v2 = a4[2];
      if (v2.w != 0.0f) {
        const float4 my2 = lrec[tid * 3 + 2];
        const uint32_t off = __float_as_uint(my2.z);
        const uint32_t rb = __float_as_uint(my2.w);
        const uint32_t minx = rb & 1023u, miny = (rb >> 10) & 1023u, rw = rb >> 20;
        const uint32_t e = off + ((uint32_t)tg.ty - miny) * rw + ((uint32_t)tg.tx - minx);
        float4* dst = inst + (size_t)e * 3;
        dst[0] = a4[0];
        dst[1] = a4[1];
        dst[2] = make_float4(v2.x, v2.y, 0.f, 0.f);
      }
    }
  }
}


// ================================================================================
// Default backward: BACK-TO-FRONT, one wave per tile (4 pixels per lane).
//
// Numerics: the front-to-back form above needs R_i = V - prefix_i, whose absolute error is
// eps*|V|; divided by (1 - alpha_i) >= 0.01 that is up to 1e-5*|V| per capped pixel (measured:
// max error 5e-5 of the stack maximum with 5-10 % capped alphas vs 6e-7 back-to-front).  Walking
// back from the forward's last contributor with T_i = T_{i+1} / (1 - alpha_i) and the scalar
// recurrence A_i = alpha_{i+1} q_{i+1} + (1 - alpha_{i+1}) A_{i+1} keeps every term relatively
// accurate:  dL/dalpha_i = (q_i - A_i) T_i - T_final (dL/dC . bg) / (1 - alpha_i).
//
// Reduction: 10 partial sums per (tile, Gaussian): per-lane over the 4 strips for free, DPP row
// reductions + row_bcast to lane 63, which stores the 48-byte instance record straight to its
// emission slot.  No LDS accumulator, no atomics of any kind, bit-reproducible.
// ================================================================================
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = row_sum16(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));  // row_bcast:15
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));  // row_bcast:31
  return v;
}

template <bool DEPTH>
__global__ __launch_bounds__(64) void render_bwd_b2f_kernel(
    const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ records, int W, int H, int gx, int T, const float* __restrict__ bg,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_dinvdepth, float4* __restrict__ inst) {
  constexpr int S = 4;
  constexpr int BATCH = 64;
  __shared__ float4 lrec[BATCH * 3];

  TileGeom tg;
  if (!block_to_tile(T, gx, tg)) return;
  const int lane = threadIdx.x;
  const int px = tg.tx * kTile + (lane & 15);
  const int py0 = tg.ty * kTile + (lane >> 4);
  const float fpx = (float)px;
  const size_t plane = (size_t)W * H;
  const float b0 = bg[0], b1 = bg[1], b2 = bg[2];

  float fpy[S], Tr[S], Aq[S], la[S], lq[S], bgd[S], g0[S], g1[S], g2[S], gd[S];
  uint32_t nc[S];
  uint32_t maxnc = 0;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int py = py0 + 4 * s;
    fpy[s] = (float)py;
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
      const float4* r = records + (size_t)gid * 3;
      lrec[lane * 3 + 0] = r[0];
      lrec[lane * 3 + 1] = r[1];
      lrec[lane * 3 + 2] = r[2];
    }
    __syncthreads();
    for (int j = n - 1; j >= 0; --j) {
      const uint32_t rel = (uint32_t)(bstart + j);
      const float4 q0 = lrec[j * 3 + 0];
      const float4 q1 = lrec[j * 3 + 1];
      const float4 q2 = lrec[j * 3 + 2];
      const float dx = q0.x - fpx;
      const float ax = q0.z * dx * dx;
      const float bx = q0.w * dx;
      const float thr = (kLog2AlphaMin - kSkipGuard) - __builtin_amdgcn_logf(q1.y);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f, s8 = 0.f, s9 = 0.f;
      uint64_t any_blend = 0;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float dy = q0.y - fpy[s];
        const float pw = ax + dy * (q1.x * dy + bx);
        const bool cand = (pw >= thr) && (rel < nc[s]);
        if (__ballot(cand) != 0) {
          const float G = fast_exp2(pw);
          const float araw = q1.y * G;
          const float alpha = fminf(kAlphaMax, araw);
          const bool live = cand && (pw <= 0.0f) && (alpha >= kAlphaMin);
          const uint64_t lm = __ballot(live);
          if (lm != 0) {
            any_blend |= lm;
            const float rinv = __builtin_amdgcn_rcpf(1.0f - alpha);
            const float Tcur = Tr[s] * rinv;                       // transmittance in front of this Gaussian
            const float An = Aq[s] + la[s] * (lq[s] - Aq[s]);      // blended value behind it (per unit T)
            float q = g0[s] * q1.z + g1[s] * q1.w + g2[s] * q2.x;
            if (DEPTH) q += gd[s] * q2.y;
            const float dLda = live ? ((q - An) * Tcur - bgd[s] * rinv) : 0.0f;
            const float w = live ? alpha * Tcur : 0.0f;
            const float X = araw * dLda;                           // dL/dpower, straight-through 0.99 cap
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

template <int S, bool V2 = false>
static int launch_fwd_s(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                        float* out_color, float* out_invdepth, hipStream_t s) {
  const int gxx = grid_x(a.width), T = gxx * grid_y(a.height);
  const int nblk = ((T + 7) / 8) * 8;
  const dim3 block(64 * (4 / S));
  const float4* rec = reinterpret_cast<const float4*>(g.records);
  const bool depth = a.do_depth && out_invdepth;
  auto kern = V2 ? (depth ? render_fwd2_kernel<S, true> : render_fwd2_kernel<S, false>)
                 : (depth ? render_fwd_kernel<S, true> : render_fwd_kernel<S, false>);
  hipLaunchKernelGGL(kern, dim3(nblk), block, 0, s, b.ranges, b.vals_out, rec, a.width, a.height, gxx, T, a.bg,
                     out_color, out_invdepth, im.final_T, im.n_contrib);
  HGS_LAUNCH_CHECK("render_fwd", s, a.debug);
  return HGS_OK;
}

int launch_render_fwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      float* out_color, float* out_invdepth, hipStream_t s) {
  switch (a.variant) {
    case 1: return launch_fwd_s<1>(a, g, b, im, out_color, out_invdepth, s);
    case 2: return launch_fwd_s<2>(a, g, b, im, out_color, out_invdepth, s);
    case 3: return launch_fwd_s<4>(a, g, b, im, out_color, out_invdepth, s);
    case 5: return launch_fwd_s<1, true>(a, g, b, im, out_color, out_invdepth, s);
    case 6: return launch_fwd_s<2, true>(a, g, b, im, out_color, out_invdepth, s);
    default: return launch_fwd_s<4, true>(a, g, b, im, out_color, out_invdepth, s);
  }
}

template <int S, bool V2 = false>
static int launch_bwd_s(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const float* out_color,
                        const float* out_invdepth, const float* dL_dcolor, const float* dL_dinvdepth,
                        float* inst_grads, hipStream_t s) {
  const int gxx = grid_x(a.width), T = gxx * grid_y(a.height);
  const int nblk = ((T + 7) / 8) * 8;
  const dim3 block(64 * (4 / S));
  const float4* rec = reinterpret_cast<const float4*>(g.records);
  float4* inst = reinterpret_cast<float4*>(inst_grads);
  const bool depth = a.do_depth && out_invdepth && dL_dinvdepth;
  auto kern = V2 ? (depth ? render_bwd2_kernel<S, true> : render_bwd2_kernel<S, false>)
                 : (depth ? render_bwd_kernel<S, true> : render_bwd_kernel<S, false>);
  hipLaunchKernelGGL(kern, dim3(nblk), block, 0, s, b.ranges, b.vals_out, rec, a.width, a.height, gxx, T, out_color,
                     out_invdepth, dL_dcolor, dL_dinvdepth, inst);
  HGS_LAUNCH_CHECK("render_bwd", s, a.debug);
  return HGS_OK;
}

int launch_render_bwd(const hgs_raster_args& a, const GeomWs& g, const BinWs& b, const ImgWs& im,
                      const float* out_color, const float* out_invdepth, const float* dL_dcolor,
                      const float* dL_dinvdepth, float* inst_grads, hipStream_t s) {
  if (a.variant == 0 || a.variant == 4) {
    const int gxx = grid_x(a.width), T = gxx * grid_y(a.height);
    const int nblk = ((T + 7) / 8) * 8;
    const bool depth = a.do_depth && out_invdepth && dL_dinvdepth;
    auto kern = depth ? render_bwd_b2f_kernel<true> : render_bwd_b2f_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, s, b.ranges, b.vals_out,
                       reinterpret_cast<const float4*>(g.records), a.width, a.height, gxx, T, a.bg, im.final_T,
                       im.n_contrib, dL_dcolor, dL_dinvdepth, reinterpret_cast<float4*>(inst_grads));
    HGS_LAUNCH_CHECK("render_bwd_b2f", s, a.debug);
    return HGS_OK;
  }
  switch (a.variant) {
    case 1: return launch_bwd_s<1>(a, g, b, out_color, out_invdepth, dL_dcolor, dL_dinvdepth, inst_grads, s);
    case 2: return launch_bwd_s<2>(a, g, b, out_color, out_invdepth, dL_dcolor, dL_dinvdepth, inst_grads, s);
    case 3: return launch_bwd_s<4>(a, g, b, out_color, out_invdepth, dL_dcolor, dL_dinvdepth, inst_grads, s);
    case 5: return launch_bwd_s<1, true>(a, g, b, out_color, out_invdepth, dL_dcolor, dL_dinvdepth, inst_grads, s);
    case 6: return launch_bwd_s<2, true>(a, g, b, out_color, out_invdepth, dL_dcolor, dL_dinvdepth, inst_grads, s);
    default: return launch_bwd_s<4, true>(a, g, b, out_color, out_invdepth, dL_dcolor, dL_dinvdepth, inst_grads, s);  // 7: f2b v2, S=4
  }
}

}  // namespace hgs
