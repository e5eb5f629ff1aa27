// K1 (preprocess forward) and K8 (preprocess backward): one lane per Gaussian.
//
// Restates the per-Gaussian stage of the op called at
// gaussian_renderer/__init__.py:105-113 (reference source absent; algorithm per
// SURVEY.md App. A 1-6, 9-10).  HBM-bound streaming kernels: 236 B in per Gaussian
// at SH degree 3 (fwd), 236 B in + 248 B out (bwd).
#include "gaussian_math.h"

#include <stdlib.h>

#include <mutex>

namespace hgs {

namespace {

struct CamLds {
  float vm[16];
  float pm[16];
  float cam[3];
};

__device__ __forceinline__ void load_camera(const hgs_raster_args& a, CamLds& c) {
  // 35 floats, wave-uniform addresses -> scalar loads.
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    c.vm[i] = a.viewmatrix[i];
    c.pm[i] = a.projmatrix[i];
  }
  c.cam[0] = a.campos[0];
  c.cam[1] = a.campos[1];
  c.cam[2] = a.campos[2];
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void load_sh(const float* __restrict__ shs, int idx, int M, float sh[48]) {
  const int n = M * 3;
  const float* src = shs + (size_t)idx * n;
  if ((n & 3) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i * 4 < n) {
        const float4 v = s4[i];
        sh[i * 4 + 0] = v.x;
        sh[i * 4 + 1] = v.y;
        sh[i * 4 + 2] = v.z;
        sh[i * 4 + 3] = v.w;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 48; ++i)
      if (i < n) sh[i] = src[i];
  }
}

// 16-byte stores of a Gaussian's M*3 SH gradients (12 x dwordx4 per lane at M = 16)
__device__ __forceinline__ void store_sh(float* __restrict__ dshs, int idx, int M, const float v[48]) {
  const int n = M * 3;
  float* dst = dshs + (size_t)idx * n;
  if ((n & 3) == 0) {
    float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll
    for (int i = 0; i < 12; ++i)
      if (i * 4 < n) d4[i] = make_float4(v[i * 4 + 0], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < 48; ++i)
      if (i < n) dst[i] = v[i];
  }
}

// ---- cooperative (coalesced) access to the [P, M, 3] SH arrays --------------------------------------
// One lane per Gaussian wants 12*M contiguous bytes (192 B at M = 16): issued per lane, every dwordx4
// instruction touches 64 different cache lines (measured 2.6 TB/s).  Instead the workgroup streams its
// 256 Gaussians' coefficients as one contiguous block (lane i <-> 16 bytes i) through LDS, rows padded
// by 4 floats so that the per-lane ds_read_b128 / ds_write_b128 are bank-conflict free.
__device__ __forceinline__ int sh_row_stride(int n) { return n + 4; }   // floats

__device__ __forceinline__ void coop_load_sh(const float* __restrict__ shs, int block_first, int P, int n,
                                             float* lds) {
  const int count = min(kPreBlock, P - block_first);           // Gaussians in this workgroup
  const int vecs = count * n / 4;
  const float4* src = reinterpret_cast<const float4*>(shs + (size_t)block_first * n);
  const int stride = sh_row_stride(n);
  for (int v = threadIdx.x; v < vecs; v += kPreBlock) {
    const int e = v * 4;
    const int gsn = e / n, off = e - gsn * n;
    *reinterpret_cast<float4*>(lds + gsn * stride + off) = src[v];
  }
}

// The same block load split in two: ISSUE the workgroup's coalesced 16-byte loads into registers (up to 12 per lane),
// COMMIT them to the padded LDS rows later -- whatever the kernel computes in between runs under the loads' latency.
__device__ __forceinline__ void coop_issue_sh(const float* __restrict__ shs, int block_first, int P, int n,
                                              float4 (&reg)[12]) {
  const int vecs = min(kPreBlock, P - block_first) * n / 4;
  const float4* src = reinterpret_cast<const float4*>(shs + (size_t)block_first * n);
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int v = (int)threadIdx.x + k * kPreBlock;
    reg[k] = v < vecs ? src[v] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void coop_commit_sh(const float4 (&reg)[12], int block_first, int P, int n, float* lds) {
  const int vecs = min(kPreBlock, P - block_first) * n / 4;
  const int stride = sh_row_stride(n);
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int v = (int)threadIdx.x + k * kPreBlock;
    if (v < vecs) {
      const int e = v * 4;
      const int gsn = n == 48 ? e / 48 : e / n, off = e - gsn * n;
      *reinterpret_cast<float4*>(lds + gsn * stride + off) = reg[k];
    }
  }
}

template <bool ACC>
__device__ __forceinline__ void coop_store_sh(float* __restrict__ dst_all, int block_first, int P, int n,
                                              const float* lds) {
  const int count = min(kPreBlock, P - block_first);
  const int vecs = count * n / 4;
  float4* dst = reinterpret_cast<float4*>(dst_all + (size_t)block_first * n);
  const int stride = sh_row_stride(n);
  if (ACC) {
    // accumulate: reads of the old gradients are issued four at a time ahead of the adds and stores -- a
    // load/add/store chain per 16 bytes would serialise on memory latency, twelve in flight spill registers
    constexpr int kGroup = 4;
    for (int base = 0; base < vecs; base += kGroup * kPreBlock) {
      float4 old[kGroup];
#pragma unroll
      for (int it = 0; it < kGroup; ++it) {
        const int v = base + it * kPreBlock + (int)threadIdx.x;
        old[it] = v < vecs ? dst[v] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < kGroup; ++it) {
        const int v = base + it * kPreBlock + (int)threadIdx.x;
        if (v < vecs) {
          const int e = v * 4;
          const int gsn = e / n, off = e - gsn * n;
          const float4 val = *reinterpret_cast<const float4*>(lds + gsn * stride + off);
          dst[v] = make_float4(val.x + old[it].x, val.y + old[it].y, val.z + old[it].z, val.w + old[it].w);
        }
      }
    }
  } else {
    for (int v = threadIdx.x; v < vecs; v += kPreBlock) {
      const int e = v * 4;
      const int gsn = e / n, off = e - gsn * n;
      dst[v] = *reinterpret_cast<const float4*>(lds + gsn * stride + off);
    }
  }
}

__device__ __forceinline__ void lds_row_read(const float* lds, int n, float v[48]) {
  const float* row = lds + threadIdx.x * sh_row_stride(n);
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    if (i * 4 < n) {
      const float4 t = *reinterpret_cast<const float4*>(row + i * 4);
      v[i * 4 + 0] = t.x; v[i * 4 + 1] = t.y; v[i * 4 + 2] = t.z; v[i * 4 + 3] = t.w;
    }
  }
}
__device__ __forceinline__ void lds_row_write(float* lds, int n, const float v[48]) {
  float* row = lds + threadIdx.x * sh_row_stride(n);
#pragma unroll
  for (int i = 0; i < 12; ++i)
    if (i * 4 < n) *reinterpret_cast<float4*>(row + i * 4) = make_float4(v[i * 4 + 0], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]);
}

// ---- split SH storage (hgs_raster_args.shs_rest): features_dc [P,1,3] and features_rest [P,M-1,3] ----------
// Same LDS row layout as above ([dc | rest], stride 3M + 4), filled from / drained to two contiguous global
// blocks.  3 and 45 floats per Gaussian are not multiples of 4, so the 16-byte global accesses straddle rows and
// the LDS side is done per float; a workgroup's block starts 16-byte aligned (256 Gaussians x 12 / 180 bytes).
__device__ __forceinline__ void coop_load_seg(const float* __restrict__ src_all, int block_first, int P, int nseg,
                                              int col0, int stride, float* lds) {
  const int count = min(kPreBlock, P - block_first);
  const int total = count * nseg;
  const float* src = src_all + (size_t)block_first * nseg;
  const int vecs = total >> 2;
  for (int v = threadIdx.x; v < vecs; v += kPreBlock) {
    const float4 t = reinterpret_cast<const float4*>(src)[v];
    const float tv[4] = {t.x, t.y, t.z, t.w};
    int gsn = (v * 4) / nseg, off = v * 4 - gsn * nseg;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      lds[gsn * stride + col0 + off] = tv[c];
      if (++off == nseg) { off = 0; ++gsn; }
    }
  }
  for (int e = vecs * 4 + threadIdx.x; e < total; e += kPreBlock) {
    const int gsn = e / nseg, off = e - gsn * nseg;
    lds[gsn * stride + col0 + off] = src[e];
  }
}

template <bool ACC>
__device__ __forceinline__ void coop_store_seg(float* __restrict__ dst_all, int block_first, int P, int nseg,
                                               int col0, int stride, const float* lds) {
  const int count = min(kPreBlock, P - block_first);
  const int total = count * nseg;
  float* dst = dst_all + (size_t)block_first * nseg;
  const int vecs = total >> 2;
  for (int v = threadIdx.x; v < vecs; v += kPreBlock) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ACC) o = reinterpret_cast<const float4*>(dst)[v];
    float tv[4];
    int gsn = (v * 4) / nseg, off = v * 4 - gsn * nseg;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      tv[c] = lds[gsn * stride + col0 + off];
      if (++off == nseg) { off = 0; ++gsn; }
    }
    reinterpret_cast<float4*>(dst)[v] = make_float4(tv[0] + o.x, tv[1] + o.y, tv[2] + o.z, tv[3] + o.w);
  }
  for (int e = vecs * 4 + threadIdx.x; e < total; e += kPreBlock) {
    const int gsn = e / nseg, off = e - gsn * nseg;
    const float v = lds[gsn * stride + col0 + off];
    dst[e] = ACC ? dst[e] + v : v;
  }
}

// per-lane access to the split layout (mostly-culled workgroups of K1)
__device__ __forceinline__ void load_sh_split(const float* __restrict__ dc, const float* __restrict__ rest, int idx,
                                              int M, float sh[48]) {
  sh[0] = dc[(size_t)idx * 3 + 0]; sh[1] = dc[(size_t)idx * 3 + 1]; sh[2] = dc[(size_t)idx * 3 + 2];
  const int nr = (M - 1) * 3;
  const float* src = rest + (size_t)idx * nr;
#pragma unroll
  for (int i = 0; i < 45; ++i)
    if (i < nr) sh[3 + i] = src[i];
}

// In-kernel LOD interpolation: the workgroup's rows are gathered (node row, parent row) and lerped on their way into
// LDS, one lane per 16-byte chunk so that every 192-byte row is read as contiguous segments.  (node, parent, weight) of
// row i sit in the pad floats of its LDS slot (written by lane i before the barrier that precedes this call).
__device__ __forceinline__ void coop_gather_sh(const float* __restrict__ shs, int block_first, int P, int n, float* lds) {
  const int count = min(kPreBlock, P - block_first);
  const int cpr = n / 4;                                        // 16-byte chunks per row
  const int stride = sh_row_stride(n);
  const float4* src = reinterpret_cast<const float4*>(shs);
  for (int v = threadIdx.x; v < count * cpr; v += kPreBlock) {
    const int row = v / cpr, c = v - row * cpr;
    const float* pad = lds + row * stride + n;
    const size_t r = (size_t)__float_as_uint(pad[0]), p = (size_t)__float_as_uint(pad[1]);
    const float w = pad[2], u = 1.0f - w;
    const float4 x = src[r * cpr + c], y = src[p * cpr + c];     // (weight 1: p = r, see lod_row_gather)
    *reinterpret_cast<float4*>(lds + row * stride + c * 4) =
        make_float4(lod_lerp(x.x, y.x, w, u), lod_lerp(x.y, y.y, w, u), lod_lerp(x.z, y.z, w, u), lod_lerp(x.w, y.w, w, u));
  }
}
__device__ __forceinline__ void load_sh_lod(const hgs_raster_args& a, int idx, float sh[48]) {
  const LodRow l = lod_row_gather<true>(a, idx);
  const int n = a.M * 3;
  const float* x = a.shs + l.r * n;
  const float* y = a.shs + l.p * n;
#pragma unroll
  for (int i = 0; i < 48; ++i)
    if (i < n) sh[i] = lod_lerp(x[i], y[i], l.w, l.u);
}


// ---- M = 16, plain [P, 16, 3] layout: the coefficient block straight into LDS ----------------------------------------------
// K1 at M = 16 used to stage the workgroup's whole block (256 rows x 52 floats = 53 KB) through LDS and to hold its
// twelve 16-byte loads per lane in registers across the double-precision chain: 53 KB and 144 registers both capped the
// kernel at 3 waves per SIMD.  `global_load_lds_dwordx4` writes the rows into LDS without passing through registers
// (nothing to hold across the chain); the image is UNPADDED (the DMA's destination is the wave's base + 16 * lane: no
// room for pad), so the bank swizzle is applied on the SOURCE address of the DMA and on the read, never on the
// destination (cdna_hip_programming.md rule 21).  (Round 5 also built the block as two half rows -- every line requested
// twice -- and as two row groups, and K1 as a geometry kernel + a colour kernel on a second stream: all measured slower,
// profiles/r05_k1_layouts.txt, r05_k1_split_ab.txt; removed in round 6.)
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

// rgb and (JAC) d(rgb)/d(direction) sums over the four coefficients [K0, K0 + 4); same order of additions as the
// one-block loop (k ascending), so the two routes give the same bits
template <bool JAC, int K0>
__device__ __forceinline__ void sh48_accumulate4(int deg, float dx, float dy, float dz, const float sh[12], float rgb[3],
                                                 float J[9]) {
  const int nb = (deg + 1) * (deg + 1);
  float b[16];
  sh_basis(deg, dx, dy, dz, b);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (K0 + k < nb) {                                     // (explicit FMAs: the same bits in every instantiation)
      rgb[0] = __builtin_fmaf(b[K0 + k], sh[k * 3 + 0], rgb[0]);
      rgb[1] = __builtin_fmaf(b[K0 + k], sh[k * 3 + 1], rgb[1]);
      rgb[2] = __builtin_fmaf(b[K0 + k], sh[k * 3 + 2], rgb[2]);
    }
  }
  if constexpr (JAC) {
    float dbx[16], dby[16], dbz[16];
    sh_basis_grad(deg, dx, dy, dz, dbx, dby, dbz);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (K0 + k < nb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          J[0 + c] = __builtin_fmaf(dbx[K0 + k], sh[k * 3 + c], J[0 + c]);
          J[3 + c] = __builtin_fmaf(dby[K0 + k], sh[k * 3 + c], J[3 + c]);
          J[6 + c] = __builtin_fmaf(dbz[K0 + k], sh[k * 3 + c], J[6 + c]);
        }
      }
    }
  }
}
// the same from global memory (mostly-culled workgroups: visible lanes only), 48 bytes at a time
template <bool JAC>
__device__ __forceinline__ void sh48_row_from_global(const float* __restrict__ shs, int idx, int deg, float dx, float dy,
                                                     float dz, float rgb[3], float J[9]) {
  const float4* src = reinterpret_cast<const float4*>(shs + (size_t)idx * 48);
  const int nb = (deg + 1) * (deg + 1);
  float sh[12];
#define HGS_SH48_Q(Q)                                                                         \
  if (Q * 4 < nb) {                                                                           \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                           \
      const float4 t = src[Q * 3 + c];                                                        \
      sh[c * 4 + 0] = t.x; sh[c * 4 + 1] = t.y; sh[c * 4 + 2] = t.z; sh[c * 4 + 3] = t.w;     \
    }                                                                                         \
    sh48_accumulate4<JAC, Q * 4>(deg, dx, dy, dz, sh, rgb, J);                                \
  }
  HGS_SH48_Q(0) HGS_SH48_Q(1) HGS_SH48_Q(2) HGS_SH48_Q(3)
#undef HGS_SH48_Q
}

// All 256 rows at once (48 KB: three workgroups per compute unit -- enough, because the DMA keeps 48 KB per workgroup in
// flight without a register).  A row = 12 chunks of 16 bytes at row * 192; row r keeps chunk c in slot (c + f(r)) mod 12,
// f(r) = bits 2..3 of r: the 16 rows of a ds_read_b128 lane group then cover all 16 bank groups (192 r mod 256 takes four
// values).  Only the chunks the active degree needs are fetched.
constexpr int kK1ImageBytes = kPreBlock * 192;

__device__ __forceinline__ void sh48_issue_rows(const float* __restrict__ shs, int block_first, int P, int nchunks,
                                                float* lds) {
  const int count = min(kPreBlock, P - block_first);
  const char* src = reinterpret_cast<const char*>(shs + (size_t)block_first * 48);
  const int wave_base = (int)(threadIdx.x & ~63u);
  constexpr int kPer = 12;                                      // DMA instructions per lane
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int slot = k * kPreBlock + (int)threadIdx.x;          // linear 16-byte slot of the LDS image
    const int row_l = slot / 12, cs = slot - row_l * 12;
    int c = cs - ((row_l >> 2) & 3);                             // the logical chunk this slot keeps
    c = c < 0 ? c + 12 : c;
    const int row = row_l;
    char* dst = reinterpret_cast<char*>(lds) + (size_t)(k * kPreBlock + wave_base) * 16;   // wave-uniform
    if (row < count && c < nchunks)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)row * 192 + c * 16), (lptr_t)dst, 16, 0, 0);
  }
}
// row `row_l` of the image, coefficients [4 Q, 4 Q + 4): 12 floats
template <int Q>
__device__ __forceinline__ void sh48_read_row_quarter(const float* lds, int row_l, float v[12]) {
  const int f = (row_l >> 2) & 3;
  const float* row = lds + row_l * 48;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int sl = Q * 3 + c + f;
    sl = sl >= 12 ? sl - 12 : sl;
    const float4 t = *reinterpret_cast<const float4*>(row + sl * 4);
    v[c * 4 + 0] = t.x; v[c * 4 + 1] = t.y; v[c * 4 + 2] = t.z; v[c * 4 + 3] = t.w;
  }
}
template <bool JAC>
__device__ __forceinline__ void sh48_row_from_lds(const float* lds, int row_l, bool vis, int deg, float dx, float dy,
                                                  float dz, float rgb[3], float J[9]) {
  const int nb = (deg + 1) * (deg + 1);
  float sh[12];
  sh48_read_row_quarter<0>(lds, row_l, sh);
  if (vis) sh48_accumulate4<JAC, 0>(deg, dx, dy, dz, sh, rgb, J);
  if (nb > 4) {                                                  // (uniform)
    sh48_read_row_quarter<1>(lds, row_l, sh);
    if (vis) sh48_accumulate4<JAC, 4>(deg, dx, dy, dz, sh, rgb, J);
    if (nb > 8) {
      sh48_read_row_quarter<2>(lds, row_l, sh);
      if (vis) sh48_accumulate4<JAC, 8>(deg, dx, dy, dz, sh, rgb, J);
      sh48_read_row_quarter<3>(lds, row_l, sh);
      if (vis) sh48_accumulate4<JAC, 12>(deg, dx, dy, dz, sh, rgb, J);
    }
  }
}

// ---- the per-workgroup sums of K1 and their scan -----------------------------------------------------------------------
// Every workgroup of K1 leaves nine sums (its instances, and its instances per tile band); K3 needs their exclusive scans
// over the workgroups.  Rounds 1-4 ran a scan launch between K1 and K3 (6.6 us + a kernel boundary for 140 KB).  With
// `super` given, that launch is gone: K1 also ADDS its sums (fire-and-forget atomics, nobody waits for an answer) to the
// totals of its SUPERBLOCK of kSuper consecutive workgroups, and workgroup b of K3 puts its nine prefixes together
// itself -- the superblock totals before b's superblock plus the raw sums of the workgroups before b inside it: two
// 64-lane loads and two wave reductions per array (binning.hip).  Integer adds commute, the consumer runs behind a
// kernel boundary: no hand-off inside a launch, no assumption about dispatch order, residency or placement.
// The superblock totals must be ZERO when K1 starts; they live in a block owned by the library per (device, stream),
// zeroed when it is created and zeroed again by the counting kernel that follows K3 (tile_bin.hip) -- a workspace carved
// from the caller's memory is fresh on every call and would need a memset launch in front of K1, the launch this is
// there to save.
// instances per tile band of one rectangle, added to the workgroup's LDS counters.  Almost every rectangle lies inside ONE
// band (first and last tile in the same band: seven compares each, no division); one that straddles boundaries counts
// (# of its tiles with id < x) at the boundaries x = b * per it spans, in closed form.
__device__ __forceinline__ void k1_band_count(const Proj& pr, uint32_t touched, int gx, int gy, uint32_t* band_cnt) {
  const int T = gx * gy, per = band_tiles(T), w = pr.maxx - pr.minx;
  const int t_first = pr.miny * gx + pr.minx, t_last = (pr.maxy - 1) * gx + pr.maxx - 1;
  int b_first = 0, b_last = 0;
#pragma unroll
  for (int b = 1; b < kBands; ++b) {
    b_first += (t_first >= b * per) ? 1 : 0;
    b_last += (t_last >= b * per) ? 1 : 0;
  }
  if (b_first == b_last) {
    atomicAdd(&band_cnt[b_first], touched);
  } else {
    int prev = 0;
    for (int b = b_first; b <= b_last; ++b) {
      const int x = min((b + 1) * per, T);
      const int xr = x / gx, xc = x - xr * gx;
      int cnt = (min(max(xr, pr.miny), pr.maxy) - pr.miny) * w;
      if (xr >= pr.miny && xr < pr.maxy) cnt += min(max(xc - pr.minx, 0), w);
      if (cnt > prev) atomicAdd(&band_cnt[b], (uint32_t)(cnt - prev));
      prev = cnt;
    }
  }
}

// The continuous quantities of a visible Gaussian's 2D record (double-precision chain, gaussian_math.h).
struct K1Rec {
  float opac, thr, ext_x, ext_y, invz;
  float gx_hi, gy_hi, gx_lo, gy_lo, A2, B2, C2;
};
__device__ __forceinline__ void k1_continuous(const hgs_raster_args& a, const CamLds& cam, const Proj& pr,
                                              const float p[3], const float sc_act[3], const float q_act[4],
                                              float opac, K1Rec& o) {
  o.opac = opac;
  ProjD pd;
  if (a.cov3D_precomp) {
#pragma unroll
    for (int i = 0; i < 6; ++i) pd.c3[i] = (double)pr.c3[i];
  } else {
    cov3d_from_scale_rot_d(sc_act, a.scale_modifier, q_act, pd);
  }
  project_gaussian_d(p, cam.vm, cam.pm, a.width, a.height, a.tanfovx, a.tanfovy, pr.clampx, pr.clampy, pd);
  // pixel centre as hi + lo floats: the render kernels make it tile-relative before use
  o.gx_hi = (float)pd.px; o.gy_hi = (float)pd.py;
  o.gx_lo = (float)(pd.px - (double)o.gx_hi); o.gy_lo = (float)(pd.py - (double)o.gy_hi);
  // conic pre-scaled to a base-2 exponent: power2 = A2*dx^2 + C2*dy^2 + B2*dx*dy
  const double kLog2e = 1.4426950408889634;
  o.A2 = (float)(-0.5 * kLog2e * pd.conA);
  o.B2 = (float)(-kLog2e * pd.conB);
  o.C2 = (float)(-0.5 * kLog2e * pd.conC);
  // The compositing kernels clamp the exponent at 0 instead of testing its sign ("power > 0 -> skip" never fires
  // for a positive definite conic).  Rounding the three coefficients to float32 must therefore not make an extremely
  // elongated conic indefinite (relative determinant below ~1e-7: sigma of thousands of pixels): if it does, the
  // mixed term is pulled back inside by one part in a million.
  {
    const double lim = 4.0 * (double)o.A2 * (double)o.C2;
    if (!((double)o.B2 * (double)o.B2 < lim)) o.B2 = (float)copysign(sqrt(fmax(lim, 0.0)) * (1.0 - 1.0e-6), (double)o.B2);
  }
  // log-domain skip threshold of the compositing kernels: alpha >= 1/255 <=> power2 >= log2(1/255) - log2(o);
  // 1e-3 guard band so the exact alpha test keeps every borderline decision (opacity <= 0 -> +inf / NaN:
  // never a candidate, exactly like the exact test)
  o.thr = (-7.994353436858858f - 1.0e-3f) - __builtin_amdgcn_logf(opac);
  // Half extents (pixels) of the axis-aligned box around the region where alpha can reach 1/255:
  //   d^T conic d <= T2 = 2 (ln(255 o) + guard)  =>  |dx| <= sqrt(T2 Sigma'_xx), |dy| <= sqrt(T2 Sigma'_yy)
  // (the extent of an ellipse along an axis is set by the COVARIANCE's diagonal: no determinant, no cancellation, so
  // float32 is enough).  T2 = -2 ln 2 * thr: the same logarithm as the skip threshold.  The compositing kernels
  // use the box to decide, once per (tile, Gaussian), which quadrants can be touched at all; it is inflated (the 1e-3
  // guard in the base-2 exponent, 1e-4 relative, 5e-3 px) so that it contains every pixel the exact alpha test could
  // accept.  -1: no pixel ever (o <= ~1/255).
  o.ext_x = -1.0f; o.ext_y = -1.0f;
  {
    const float T2 = -1.3862943611198906f * o.thr;
    const float sxx = (float)pd.a, syy = (float)pd.c;
    if (T2 > 0.0f) {
      const bool sane = sxx < 3.0e38f && syy < 3.0e38f;      // (NaN / inf covariance: never skip on the box)
      o.ext_x = sane ? sqrtf(T2 * sxx) * 1.0001f + 5.0e-3f : 1.0e9f;
      o.ext_y = sane ? sqrtf(T2 * syy) * 1.0001f + 5.0e-3f : 1.0e9f;
    }
  }
  o.invz = (float)pd.itz;
}

// How the M = 16 K1 stores its 64-byte records and 48-byte Jacobian rows: through the wave's own part of the LDS image
// (free once the wave has read its rows; a lane's row of NV 16-byte pieces goes to LDS as it will lie in memory, then
// every store instruction of the wave writes 1 KB of consecutive bytes).  Straight from the lane -- NV instructions
// that each touch 64 different cache lines -- a row-per-lane store pattern is bound by the rate at which the memory
// pipeline takes partial lines, not by bandwidth (8 us more; what each part of K1 costs: profiles/r05_k1_anatomy.txt).
// v[NV] of lane l = the NV consecutive float4 of row (row0 + l) of a [rows, NV] float4 array at dst; rows whose bit in
// `mask` is clear are not written.  wave_lds: >= 64 * NV * 16 bytes private to the calling wave.
template <int NV>
__device__ __forceinline__ void wave_store_rows(float4* __restrict__ dst, size_t row0, const float4 (&v)[NV],
                                                unsigned long long mask, float4* wave_lds) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < NV; ++q) wave_lds[lane * NV + q] = v[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = j * 64 + lane;                 // float4 index inside the wave's block of rows
    const int r = i / NV;
    const float4 t = wave_lds[i];
    if ((mask >> r) & 1ull) dst[row0 * NV + i] = t;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();               // (the next use of wave_lds overwrites it)
}

// The 36-byte Jacobian rows of the wave's 64 Gaussians the same way: nine words per lane at a stride of nine (odd: no
// bank conflict), read back as the 144 consecutive 16-byte pieces they are in memory.  A wave with no visible row
// stores nothing; otherwise all 64 rows are written (the rows of culled Gaussians are never read).
__device__ __forceinline__ void wave_store_jac(float* __restrict__ dst, size_t row0, const float (&J)[9],
                                               unsigned long long mask, float4* wave_lds) {
  const int lane = threadIdx.x & 63;
  float* wl = reinterpret_cast<float*>(wave_lds);
#pragma unroll
  for (int k = 0; k < 9; ++k) wl[lane * 9 + k] = J[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float4* d4 = reinterpret_cast<float4*>(dst + row0 * kJacStride);      // row0 is a multiple of 64: 16-byte aligned
  if (mask) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = j * 64 + lane;
      if (i < 64 * 9 / 4) d4[i] = wave_lds[i];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();               // (the next use of wave_lds overwrites it)
}

template <bool JAC, bool LOD, bool DEFER, bool H48>
                                // JAC: also store d(rgb)/d(direction) for the backward; LOD: in-kernel LOD
                                // interpolation; DEFER: the plain [P, M, 3] coefficient block is loaded into registers
                                // ahead of the double-precision chain (their own instantiations: the extra state
                                // would cost every other caller of K1 its occupancy); H48: plain [P, 16, 3] block
                                // straight into LDS by DMA (above; DEFER and LOD do not apply)
__device__ __forceinline__ void preprocess_fwd_body(const hgs_raster_args& a, const GeomWs& g,
                                                    int32_t* __restrict__ radii, uint32_t* __restrict__ super, uint32_t heavy_thr) {
  static_assert(!(H48 && (LOD || DEFER)), "the DMA route is the plain layout's");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* lds_sh = reinterpret_cast<float*>(smem_raw);
  __shared__ uint32_t wave_tot[kPreBlock / 64];
  __shared__ uint32_t band_cnt[kBands];          // this workgroup's instances per tile band (binning: band streams)
  if (threadIdx.x < kBands) band_cnt[threadIdx.x] = 0u;
  if (!super && blockIdx.x == 0)        // the scan launch that follows publishes its chunk totals here (common.h)
    for (int t = threadIdx.x; t < (1 + kBands) * scan_chunks(gridDim.x); t += kPreBlock) g.scan_chain[t] = 0ull;
  __syncthreads();
  const int idx = blockIdx.x * kPreBlock + threadIdx.x;
  const int gx = (a.width + kTile - 1) / kTile;
  const int gy = (a.height + kTile - 1) / kTile;
  CamLds cam;
  load_camera(a, cam);

  uint32_t touched = 0;
  Proj pr;
  pr.visible = false;
  float p[3] = {0.f, 0.f, 0.f};
  float sc_act[3] = {0.f, 0.f, 0.f}, q_act[4] = {1.f, 0.f, 0.f, 0.f};   // activated scale / rotation
  float opac = 0.f;
  constexpr bool lod = LOD;
  const int shn = a.M * 3;
  if (idx < a.P) {
    const LodRow lr = lod_row_gather<LOD>(a, idx);     // (its node / parent / weight also steer the cooperative SH gather)
    load_mean<LOD>(a, lr, p);
    if (lod && a.shs && (shn & 3) == 0) {
      // the cooperative SH gather below needs every row's (node row, parent row, weight): they ride in the four pad
      // floats at the end of the row's LDS slot
      float* pad = lds_sh + threadIdx.x * sh_row_stride(shn) + shn;
      pad[0] = __uint_as_float((uint32_t)lr.r); pad[1] = __uint_as_float((uint32_t)lr.p); pad[2] = lr.w;
    }
    if constexpr (H48) {
      // every ordinary load of the kernel is issued up here: while the LDS DMA below is in flight the compiler waits
      // for ALL outstanding memory operations at the first use of any load's result
      opac = load_opacity<LOD>(a, idx, nullptr);
      if (a.interpolation_weights && a.num_node_kids && !a.lod_per_pixel)     // (per pixel: the compositing kernels remap alpha)
        opac = lod_opacity(opac, a.interpolation_weights[idx], a.num_node_kids[idx], nullptr);
    }
    if (a.cov3D_precomp) {
#pragma unroll
      for (int i = 0; i < 6; ++i) pr.c3[i] = a.cov3D_precomp[(size_t)idx * 6 + i];
    } else {
      load_scale_rot<LOD>(a, idx, sc_act, q_act, nullptr);
      float R[9], s[3];
      cov3d_from_scale_rot(sc_act, a.scale_modifier, q_act, pr.c3, R, s);
    }
    project_gaussian(p, cam.vm, cam.pm, a.width, a.height, a.tanfovx, a.tanfovy, gx, gy, pr);
  }
  // ---- the workgroup's sums: instances, and instances per tile band (feed the offsets scans) -----------------------------
  if (idx < a.P && pr.visible) {
    touched = (uint32_t)((pr.maxx - pr.minx) * (pr.maxy - pr.miny));
    k1_band_count(pr, touched, gx, gy, band_cnt);
  }
  {
    const uint32_t ws = wave_sum_u32(touched);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = ws;
  }
  // SH coefficients: stream the workgroup's block through LDS when most of it is visible (coalesced),
  // fall back to per-lane loads (visible lanes only) when most of the block is culled.
  bool coop = false, deferred = false;
  const bool sh_lds = a.shs && (shn & 3) == 0;
  const int nvis = __syncthreads_count(pr.visible);        // (also orders wave_tot / band_cnt)
  if (threadIdx.x < 64) {                                  // (wave 0) raw sums; with `super` also into the superblock totals
    uint32_t mine = 0;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int w = 0; w < kPreBlock / 64; ++w) mine += wave_tot[w];
      g.block_sums[blockIdx.x] = mine;
    } else if (threadIdx.x <= kBands) {
      mine = band_cnt[threadIdx.x - 1];
      g.block_band[(size_t)(threadIdx.x - 1) * (gridDim.x + 1) + blockIdx.x] = mine;
    }
    if (super && threadIdx.x <= kBands && mine)
      __hip_atomic_fetch_add(&super[threadIdx.x * kMaxSuper + (blockIdx.x / kSuper)], mine, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    // ... and an outlier files itself into the heavy list: K3 shares out its emission (binning.hip)
    if (super && threadIdx.x == 0 && heavy_thr != 0u && mine > heavy_thr) {
      const uint32_t slot = __hip_atomic_fetch_add(&super[kHeavyRow], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot < (uint32_t)kMaxHeavy) {
        super[kHeavyRow + 1 + 2 * slot] = blockIdx.x;
        super[kHeavyRow + 2 + 2 * slot] = mine;
      }
    }
  }
  // Plain layout: the block's loads are ISSUED here and land while the double-precision chain below runs -- K1 is
  // bound by latency, not by its arithmetic or by HBM.
  float4 shreg[DEFER ? 12 : 1];
  if (sh_lds) {
    coop = nvis * 2 >= kPreBlock;
    if (coop) {
      if constexpr (H48) {
        sh48_issue_rows(a.shs, blockIdx.x * kPreBlock, a.P, ((a.sh_degree + 1) * (a.sh_degree + 1) * 3 + 3) / 4, lds_sh);
      } else if (lod) {
        coop_gather_sh(a.shs, blockIdx.x * kPreBlock, a.P, shn, lds_sh);
      } else if (a.shs_rest) {
        coop_load_seg(a.shs, blockIdx.x * kPreBlock, a.P, 3, 0, sh_row_stride(shn), lds_sh);
        coop_load_seg(a.shs_rest, blockIdx.x * kPreBlock, a.P, shn - 3, 3, sh_row_stride(shn), lds_sh);
      } else if constexpr (DEFER) {
        coop_issue_sh(a.shs, blockIdx.x * kPreBlock, a.P, shn, shreg);
        deferred = true;
      } else {
        coop_load_sh(a.shs, blockIdx.x * kPreBlock, a.P, shn, lds_sh);
      }
    }
  }
  // ---- everything that does not need the coefficients: the double-precision chain, the record's geometry ----------
  int32_t rad = 0;
  uint32_t flags = 0;
  K1Rec rec;
  rec.opac = 0.f; rec.thr = 0.f; rec.ext_x = -1.0f; rec.ext_y = -1.0f; rec.invz = 0.f;
  rec.gx_hi = 0.f; rec.gy_hi = 0.f; rec.gx_lo = 0.f; rec.gy_lo = 0.f; rec.A2 = 0.f; rec.B2 = 0.f; rec.C2 = 0.f;
  if (idx < a.P && pr.visible) {
    rad = (int32_t)pr.rad_f;
    if (pr.clampx) flags |= 8u;
    if (pr.clampy) flags |= 16u;
    if constexpr (!H48) {
      opac = load_opacity<LOD>(a, idx, nullptr);
      if (a.interpolation_weights && a.num_node_kids && !a.lod_per_pixel)     // (per pixel: the compositing kernels remap alpha)
        opac = lod_opacity(opac, a.interpolation_weights[idx], a.num_node_kids[idx], nullptr);
    }
    k1_continuous(a, cam, pr, p, sc_act, q_act, opac, rec);
  }
  if constexpr (DEFER) {
    if (deferred) coop_commit_sh(shreg, blockIdx.x * kPreBlock, a.P, shn, lds_sh);
  }
  if constexpr (H48) {      // the DMA has landed before any wave passes the barrier
    if (coop) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (coop) __syncthreads();
  // ---- colour ---------------------------------------------------------------------------------------------------------
  float rgb[3] = {0.f, 0.f, 0.f};
  const bool vis = idx < a.P && pr.visible;
  float J[JAC ? 9 : 1];
  if constexpr (H48) {     // (launched with a.shs only)
    float dx, dy, dz;
    unit_dir(p, cam.cam, dx, dy, dz);
    float Jt[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (coop) {                                          // (the barrier above waited for the DMA)
      sh48_row_from_lds<JAC>(lds_sh, (int)threadIdx.x, vis, a.sh_degree, dx, dy, dz, rgb, Jt);
    } else if (vis) {
      sh48_row_from_global<JAC>(a.shs, idx, a.sh_degree, dx, dy, dz, rgb, Jt);
    }
    if constexpr (JAC) {
#pragma unroll
      for (int i = 0; i < 9; ++i) J[i] = Jt[i];
    }
    if (vis) {
      rgb[0] += 0.5f; rgb[1] += 0.5f; rgb[2] += 0.5f;
      if (rgb[0] < 0.f) { rgb[0] = 0.f; flags |= 1u; }
      if (rgb[1] < 0.f) { rgb[1] = 0.f; flags |= 2u; }
      if (rgb[2] < 0.f) { rgb[2] = 0.f; flags |= 4u; }
    }
  } else if (vis) {
    if (a.colors_precomp) {
      rgb[0] = a.colors_precomp[idx * 3 + 0];
      rgb[1] = a.colors_precomp[idx * 3 + 1];
      rgb[2] = a.colors_precomp[idx * 3 + 2];
    } else {
      float sh[48];
      if (coop) lds_row_read(lds_sh, shn, sh);
      else if (lod) load_sh_lod(a, idx, sh);
      else if (a.shs_rest) load_sh_split(a.shs, a.shs_rest, idx, a.M, sh);
      else load_sh(a.shs, idx, a.M, sh);
      float dx, dy, dz;
      unit_dir(p, cam.cam, dx, dy, dz);
      // (the same four-coefficient steps as the half-row route: one piece of code, one rounding behaviour -- the in-kernel
      // LOD interpolation must give the bits of the Python glue's rows rendered by the plain call)
      float Jt[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      sh48_accumulate4<JAC, 0>(a.sh_degree, dx, dy, dz, sh, rgb, Jt);
      sh48_accumulate4<JAC, 4>(a.sh_degree, dx, dy, dz, sh + 12, rgb, Jt);
      sh48_accumulate4<JAC, 8>(a.sh_degree, dx, dy, dz, sh + 24, rgb, Jt);
      sh48_accumulate4<JAC, 12>(a.sh_degree, dx, dy, dz, sh + 36, rgb, Jt);
      if constexpr (JAC) {
        // (row-per-lane stores: the routes off the M = 16 plain layout; that one stores whole lines, below)
        float* jd = g.shjac + (size_t)idx * kJacStride;
#pragma unroll
        for (int k = 0; k < 9; ++k) jd[k] = Jt[k];
      }
      float r0 = rgb[0], r1 = rgb[1], r2 = rgb[2];
      r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
      if (r0 < 0.f) { r0 = 0.f; flags |= 1u; }
      if (r1 < 0.f) { r1 = 0.f; flags |= 2u; }
      if (r2 < 0.f) { r2 = 0.f; flags |= 4u; }
      rgb[0] = r0; rgb[1] = r1; rgb[2] = r2;
    }
  }
  // ---- the record and the per-Gaussian arrays ----------------------------------------------------------------------------
  const uint32_t rectbits = (uint32_t)pr.minx | ((uint32_t)pr.miny << 10) | ((uint32_t)(pr.maxx - pr.minx) << 20);
  bool rows_stored = false;
  if constexpr (H48) {
    if (coop) {                            // (uniform; the wave's own part of the image is free by now)
      rows_stored = true;
      float4* wl = reinterpret_cast<float4*>(reinterpret_cast<char*>(lds_sh) +
                                             (threadIdx.x >> 6) * (kK1ImageBytes / (kPreBlock / 64)));
      const unsigned long long vmask = __ballot(vis);
      const size_t row0 = (size_t)blockIdx.x * kPreBlock + (threadIdx.x & ~63u);
      const float4 r4[4] = {make_float4(rec.gx_hi, rec.gy_hi, rec.A2, rec.B2), make_float4(rec.C2, rec.opac, rgb[0], rgb[1]),
                            make_float4(rgb[2], rec.invz, rec.ext_x, __uint_as_float(rectbits)),
                            make_float4(rec.gx_lo, rec.gy_lo, rec.thr, rec.ext_y)};
      wave_store_rows<4>(reinterpret_cast<float4*>(g.records), row0, r4, vmask, wl);
      if constexpr (JAC) {
        wave_store_jac(g.shjac, row0, J, vmask, wl);
      }
    }
  }
  if (idx < a.P) {
    if (pr.visible && !rows_stored) {
      float4* recp = reinterpret_cast<float4*>(g.records) + (size_t)idx * kRecVec;
      recp[0] = make_float4(rec.gx_hi, rec.gy_hi, rec.A2, rec.B2);
      recp[1] = make_float4(rec.C2, rec.opac, rgb[0], rgb[1]);
      recp[2] = make_float4(rgb[2], rec.invz, rec.ext_x, __uint_as_float(rectbits));
      recp[3] = make_float4(rec.gx_lo, rec.gy_lo, rec.thr, rec.ext_y);
      if constexpr (H48 && JAC) {
        float* jd = g.shjac + (size_t)idx * kJacStride;
#pragma unroll
        for (int k = 0; k < 9; ++k) jd[k] = J[k];
      }
    }
    // zero rectangle (= zero instances) for culled Gaussians: the binning kernels derive counts from it
    reinterpret_cast<uint2*>(g.rects)[idx] =
        pr.visible ? make_uint2((uint32_t)pr.minx | ((uint32_t)pr.miny << 16), (uint32_t)pr.maxx | ((uint32_t)pr.maxy << 16))
                   : make_uint2(0u, 0u);
    radii[idx] = rad;
    g.depths[idx] = pr.tz;            // every Gaussian: the depth sort runs over all P keys
    g.tiles_touched[idx] = touched;
    g.flags[idx] = flags;
  }
}

template <bool JAC, bool LOD, bool DEFER>
__global__ __launch_bounds__(kPreBlock) void preprocess_fwd_kernel(hgs_raster_args a, GeomWs g,
                                                                   int32_t* __restrict__ radii, uint32_t* __restrict__ super, uint32_t heavy_thr) {
  preprocess_fwd_body<JAC, LOD, DEFER, false>(a, g, radii, super, heavy_thr);
}
// The M = 16 DMA route: 48 KB of LDS per workgroup (three workgroups per compute unit), the registers held to what four
// waves per SIMD leave (128).
template <bool JAC>
__global__ __launch_bounds__(kPreBlock, 4) void preprocess_fwd_h48_kernel(hgs_raster_args a, GeomWs g,
                                                                          int32_t* __restrict__ radii,
                                                                          uint32_t* __restrict__ super, uint32_t heavy_thr) {
  preprocess_fwd_body<JAC, false, false, true>(a, g, radii, super, heavy_thr);
}

// Exclusive scan of the per-workgroup sums (nblk = P/256): grid row 0 scans block_sums, rows 1..kBands the columns of
// block_band (common.h: chained_scan_inplace, one workgroup per 8192 entries); every array has n + 1 entries, the total
// lands in entry n.
// `total_mirror` (may be null): a second home for the grand total -- mapped host memory, so that the host learns the
// instance count without a copy command in the stream (under the profiler a 4-byte D2H copy + its barriers showed as
// ~10 us of stream time; free-running frames measure the same either way).
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(uint32_t* __restrict__ sums0,
                                                               uint32_t* __restrict__ bands, int n,
                                                               unsigned long long* __restrict__ chain,
                                                               uint32_t* __restrict__ total_mirror, int c_off,
                                                               int chunks) {
  uint32_t* __restrict__ sums = blockIdx.y == 0 ? sums0 : bands + (size_t)(blockIdx.y - 1) * (n + 1);
  const uint32_t total = chained_scan_inplace(sums, n, chain + (size_t)blockIdx.y * chunks, c_off, chunks);
  if (threadIdx.x == 0 && blockIdx.y == 0 && (int)blockIdx.x + c_off == chunks - 1 && total_mirror) *total_mirror = total;
}

// ---------------------------------------------------------------------------
// K8: chain rule from the per-instance sums of the render backward to the op's
// inputs.  Instance sums (12 floats each, see render.hip):
//   0: sum X*dx   1: sum X*dy   2: sum X*dx^2   3: sum X*dx*dy   4: sum X*dy^2
//   5: sum G*dL/dalpha          6..8: sum w*dL/dC_k              9: sum w*dL/dD
// with X = dL/dpower, w = alpha*T.
// ---------------------------------------------------------------------------
// ---- in-kernel LOD scatter (hgs_raster_args.lod_scatter) ----------------------------------------------------------
// Row i of the op is w_i * attr[r_i] + (1 - w_i) * attr[p_i]; its gradient g_i goes to the node row (w_i g_i: the node
// rows of a cut are unique -> plain stores) and to the parent row ((1 - w_i) g_i, summed over the siblings).  Siblings
// are CONSECUTIVE rows when the parent indices are non-decreasing (expand_to_size emits them so): the first lane of a
// run sums it from LDS in row order and stores once.  Runs cut by a workgroup boundary (at most two per workgroup) and
// every run of an order that is not non-decreasing (flag word, launch_lod_monotone) use atomic adds instead; two
// partial sums added to a zero-filled row give the same bits in either order.
__device__ __forceinline__ int lod_parent_at(const hgs_raster_args& a, int i) {   // parent row of op row i, any i < P
  return i < a.lod_n ? a.lod_parent_indices[i] : a.lod_rows - (a.P - a.lod_n) + (i - a.lod_n);
}
struct LodRun {
  bool leader;    // first row of its run within this workgroup
  bool atomic;    // the run continues in a neighbouring workgroup, or the order is not monotone
  int end;        // one past the run's last row (workgroup-relative)
};
__device__ __forceinline__ LodRun lod_run(const hgs_raster_args& a, const int* lds_par, int t, int count, int block_first,
                                          bool nonmono) {
  LodRun r;
  const int p = lds_par[t];
  r.leader = (t == 0) || lds_par[t - 1] != p;
  r.end = t + 1;
  r.atomic = nonmono;
  if (r.leader) {
    while (r.end < count && lds_par[r.end] == p) ++r.end;
    if (t == 0 && block_first > 0 && lod_parent_at(a, block_first - 1) == p) r.atomic = true;
    if (r.end == count && block_first + count < a.P && lod_parent_at(a, block_first + count) == p) r.atomic = true;
  }
  return r;
}
__device__ __forceinline__ void lod_add(float* dst, float v, bool atomic) {
  if (atomic) atomicAdd(dst, v); else *dst = v;
}

// ---- long runs of instance records (K8a) ------------------------------------------------------------------------------
// K8a sums each Gaussian's run of instance records (one lane per Gaussian).  Benchmark scenes have 2.7 records per
// Gaussian; a TRAINED scene at 1080p has 28 on average and Gaussians that cover thousands of tiles (the 1080p run of
// profiles/r05_config2_config3_scripts_run1.log: K8 1.9 ms of a 3.6 ms frame, 11.7 ms on a hierarchy cut -- one lane
// walking 3 000 records while 63 wait).  When the frame's mean run is long (launch_preprocess_bwd) two kernels run in
// front of K8a.  Round 5 gave a wave 8 consecutive Gaussians whatever their runs; a hierarchy cut lists its big nodes side
// by side (4 096 rows holding a third of the frame's records), so a few hundred waves on a few compute units did all the
// work: 0.47 ms.  Now the RECORDS are dealt out:
//   k8_worklist_kernel   every run of more than kK8LongRun records is cut into segments of kK8Seg records and one
//                        (Gaussian, segment) pair per segment goes onto a worklist (a workgroup reserves its pairs' places
//                        with one atomic add: the ORDER of the list is not deterministic, what a pair stands for is);
//   k8_presum_work_kernel a fixed grid of waves takes the pairs in turn: lane i sums records i, i + 64, ... of the segment
//                        (consecutive 40-byte records, coalesced; double-precision partial sums), the 64 partials of each of
//                        the ten components are folded in a fixed order and the ten sums are stored, as doubles, OVER THE
//                        SEGMENT'S FIRST TWO RECORDS -- the segment is consumed, nobody else reads it.
// K8a then adds a long run's segment sums in segment order.  Every sum is a fixed tree: bit-reproducible.
__device__ __forceinline__ uint32_t k8_nseg(uint32_t n) {        // n > kK8LongRun; the last segment holds >= 2 records
  uint32_t c = (n + kK8Seg - 1u) / kK8Seg;
  if (c > 1u && n - (c - 1u) * kK8Seg < 2u) --c;
  return c;
}
// (at most kWorklistGrid workgroups, each over a contiguous share of the 256-Gaussian blocks: one workgroup per block had
// 1 465 workgroups queue up on the one counter -- 20 us of a 2 us job on the trained-scale frame)
constexpr int kWorklistGrid = 256;
__global__ __launch_bounds__(kPreBlock) void k8_worklist_kernel(int P, const uint32_t* __restrict__ tiles_touched,
                                                                uint2* __restrict__ work, uint32_t* __restrict__ counter) {
  __shared__ uint32_t wave_tot[kPreBlock / 64];
  __shared__ uint32_t base_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nblk = (P + kPreBlock - 1) / kPreBlock;
  const int per = (nblk + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b0 = (int)blockIdx.x * per, b1 = min(b0 + per, nblk);
  // pass 1: this workgroup's pairs, one reservation
  uint32_t mine = 0;
  for (int b = b0; b < b1; ++b) {
    const int idx = b * kPreBlock + (int)threadIdx.x;
    const uint32_t n = idx < P ? tiles_touched[idx] : 0u;
    mine += n > kK8LongRun ? k8_nseg(n) : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
  if (lane == 0) wave_tot[wave] = mine;
  __syncthreads();
  uint32_t total = 0;
#pragma unroll
  for (int w = 0; w < kPreBlock / 64; ++w) total += wave_tot[w];
  if (total == 0u) return;                                               // (workgroup-uniform)
  if (threadIdx.x == 0) base_s = atomicAdd(counter, total);
  __syncthreads();
  uint32_t run = base_s;
  // pass 2: block by block, the pairs in Gaussian order
  for (int b = b0; b < b1; ++b) {
    const int idx = b * kPreBlock + (int)threadIdx.x;
    const uint32_t n = idx < P ? tiles_touched[idx] : 0u;
    const uint32_t c = n > kK8LongRun ? k8_nseg(n) : 0u;
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    __syncthreads();                                                     // (the previous block's totals have been read)
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kPreBlock / 64; ++w) {
      const uint32_t t = wave_tot[w];
      before += w < wave ? t : 0u;
      tot += t;
    }
    uint2* dst = work + run + before + inc - c;
    for (uint32_t j = 0; j < c; ++j) dst[j] = make_uint2((uint32_t)idx, j);
    run += tot;
  }
}

constexpr int kPresumGrid = 2048;        // workgroups of four waves: every wave takes pairs w, w + 8192, ...
__global__ __launch_bounds__(kPreBlock) void k8_presum_work_kernel(const uint2* __restrict__ work,
                                                                   const uint32_t* __restrict__ counter,
                                                                   const uint32_t* __restrict__ tiles_touched,
                                                                   const uint32_t* __restrict__ offsets,
                                                                   float* __restrict__ inst) {
  __shared__ double red_all[kPreBlock / 64][640 + 40];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n_work = *counter;
  double* red = red_all[wave];
  for (uint32_t item = blockIdx.x * (kPreBlock / 64) + wave; item < n_work; item += gridDim.x * (kPreBlock / 64)) {   // (wave-uniform)
    const uint2 w = work[item];
    const uint32_t n = tiles_touched[w.x];
    const uint32_t r0 = w.y * kK8Seg;
    const uint32_t r1 = (w.y + 1u == k8_nseg(n)) ? n : r0 + kK8Seg;
    const size_t rbase = (size_t)offsets[w.x] + r0;
    const float2* ip = reinterpret_cast<const float2*>(inst) + rbase * 5;
    double t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0.0;
#pragma unroll 4
    for (uint32_t k = (uint32_t)lane; k < r1 - r0; k += 64u) {
      const float2 v0 = ip[k * 5 + 0], v1 = ip[k * 5 + 1], v2 = ip[k * 5 + 2], v3 = ip[k * 5 + 3], v4 = ip[k * 5 + 4];
      t[0] += v0.x; t[1] += v0.y; t[2] += v1.x; t[3] += v1.y;
      t[4] += v2.x; t[5] += v2.y; t[6] += v3.x; t[7] += v3.y;
      t[8] += v4.x; t[9] += v4.y;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) red[i * 64 + lane] = t[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // fold in two fixed stages: lane 4 c + q sums partials 16 q .. 16 q + 15 of component c, lane c the four quarter sums
    if (lane < 40) {
      double acc = 0.0;
      const double* src = red + (lane >> 2) * 64 + (lane & 3) * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += src[j];
      red[640 + lane] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 10) {
      const double* q = red + 640 + lane * 4;
      reinterpret_cast<double*>(inst + rbase * kInstStride)[lane] = ((q[0] + q[1]) + q[2]) + q[3];   // (40-byte records: 8-byte aligned)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                     // (the next pair overwrites the scratch)
  }
}

template <bool ACC, bool LOD>   // ACC: add into the gradient buffers (accumulation over the views of one optimizer step)
__global__ __launch_bounds__(kPreBlock) void preprocess_bwd_kernel(hgs_raster_args a, GeomWs g,
                                                                   const float* __restrict__ inst,
                                                                   float* __restrict__ drgb,
                                                                   float* __restrict__ dmean_rows,
                                                                   const uint32_t* __restrict__ lod_flag,
                                                                   hgs_raster_grads out, int presummed) {
  const int idx = blockIdx.x * kPreBlock + threadIdx.x;
  const bool in_range = idx < a.P;
  const uint32_t n = in_range ? g.tiles_touched[idx] : 0u;   // (out-of-range lanes stay for the wave-wide staging below)

  // ---- sum the instance partials (one contiguous run per Gaussian, emission order) --------------------------------
  // The runs of a wave's 64 Gaussians lie BACK TO BACK in the scratch (emission order = Gaussian order), ~170 records of
  // 40 bytes: the wave streams that range through LDS with all lanes loading (coalesced, a dozen loads per lane in
  // flight) and every lane then sums its own run out of LDS -- in the same order as before, so the sums keep their bits.
  // Round 3 had each lane walk its run in global memory: 2.7 dependent round trips per Gaussian on average, the longest
  // run of the wave for everybody, at 3 waves per SIMD (160 registers) to hide them.
  double s[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) s[i] = 0.0;
  static_assert(kInstStride == 10, "instance record = the ten sums, 40 bytes");
  // Two routes per wave:
  //   * (always in frames of short runs, launch_preprocess_bwd) the wave streams its contiguous range of records through
  //     LDS, each lane sums its own run from there;
  //   * `presummed` and a run of more than kK8LongRun records in the wave: k8_presum_work_kernel left every segment's ten
  //     sums at the segment's start; the other lanes walk their short runs in global memory (the wave's range is mostly
  //     the long run).
  {
    constexpr uint32_t kStageRec = 256;                                  // records per wave and pass: 10 KB
    __shared__ float2 stage[kPreBlock / 64][kStageRec * 5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long long_mask = presummed ? __ballot(n > kK8LongRun) : 0ull;
    if (long_mask == 0ull) {
      uint32_t incl = n;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      const uint32_t my0 = incl - n, my1 = incl;                         // this lane's run, in records of the wave's range
      const uint32_t total = __shfl(incl, 63, 64);
      if (total) {                                                       // (wave-uniform)
        const unsigned long long nz = __ballot(n != 0u);
        const int first = __ffsll((long long)nz) - 1;
        const uint32_t off_mine = n ? g.offsets[idx] : 0u;
        const size_t base = (size_t)__shfl(off_mine, first, 64);         // (no instances before the first non-empty lane)
        float2* st = stage[wave];
        for (uint32_t c0 = 0; c0 < total; c0 += kStageRec) {
          const uint32_t cn = min(kStageRec, total - c0) * 5u;           // float2 words of this pass
          const float2* src = reinterpret_cast<const float2*>(inst) + (base + c0) * 5;
          for (uint32_t t0 = 0; t0 < cn; t0 += 64u * 4u) {               // four loads in flight per lane
            float2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t t = t0 + (uint32_t)j * 64u + (uint32_t)lane;
              v[j] = src[min(t, cn - 1u)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t t = t0 + (uint32_t)j * 64u + (uint32_t)lane;
              if (t < cn) st[t] = v[j];
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const uint32_t k0 = max(my0, c0), k1 = min(my1, c0 + kStageRec);
          for (uint32_t k = k0; k < k1; ++k) {
            const float2* r = st + (k - c0) * 5u;
            const float2 v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3], v4 = r[4];
            s[0] += v0.x; s[1] += v0.y; s[2] += v1.x; s[3] += v1.y;
            s[4] += v2.x; s[5] += v2.y; s[6] += v3.x; s[7] += v3.y;
            s[8] += v4.x; s[9] += v4.y;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();                               // (the next pass overwrites the stage)
        }
      }
    } else if (n != 0u) {
      const size_t off = (size_t)g.offsets[idx];
      if (n > kK8LongRun) {
        const uint32_t nseg = k8_nseg(n);
        for (uint32_t j = 0; j < nseg; ++j) {                              // segment sums, in segment order
          const double* ps = reinterpret_cast<const double*>(inst + (off + (size_t)j * kK8Seg) * kInstStride);
#pragma unroll
          for (int i = 0; i < 10; ++i) s[i] += ps[i];
        }
      } else {
        const float2* ip = reinterpret_cast<const float2*>(inst) + off * 5;
#pragma unroll 2
        for (uint32_t k = 0; k < n; ++k) {
          const float2 v0 = ip[k * 5 + 0], v1 = ip[k * 5 + 1], v2 = ip[k * 5 + 2], v3 = ip[k * 5 + 3], v4 = ip[k * 5 + 4];
          s[0] += v0.x; s[1] += v0.y; s[2] += v1.x; s[3] += v1.y;
          s[4] += v2.x; s[5] += v2.y; s[6] += v3.x; s[7] += v3.y;
          s[8] += v4.x; s[9] += v4.y;
        }
      }
    }
  }
  if constexpr (!LOD) {
    if (!in_range) return;
  }

  float d_mean[3] = {0.f, 0.f, 0.f};
  float d_m2[3] = {0.f, 0.f, 0.f};
  float d_op = 0.f;
  float d_scale[3] = {0.f, 0.f, 0.f};
  float d_rot[4] = {0.f, 0.f, 0.f, 0.f};
  float d_c3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float d_col[3] = {0.f, 0.f, 0.f};
  float sums6 = 0.f, sums7 = 0.f, sums8 = 0.f;

  if (n == 0) {
    // culled: all-zero gradients (K8b zero-fills dL/dSH)
  } else {
    CamLds cam;
    load_camera(a, cam);
    sums6 = (float)s[6]; sums7 = (float)s[7]; sums8 = (float)s[8];
    // ---- recompute the forward projection (double chain; clamp decisions from K1's flags) ----
    const uint32_t flags = g.flags[idx];
    float p[3];
    load_mean<LOD>(a, lod_row_gather<LOD>(a, idx), p);
    float q[4] = {1.f, 0.f, 0.f, 0.f};
    float sc[3] = {1.f, 1.f, 1.f};
    double qnorm = 1.0;
    ProjD pd;
    if (a.cov3D_precomp) {
#pragma unroll
      for (int i = 0; i < 6; ++i) pd.c3[i] = (double)a.cov3D_precomp[(size_t)idx * 6 + i];
    } else {
      load_scale_rot<LOD>(a, idx, sc, q, &qnorm);
      cov3d_from_scale_rot_d(sc, a.scale_modifier, q, pd);
    }
    project_gaussian_d(p, cam.vm, cam.pm, a.width, a.height, a.tanfovx, a.tanfovy, (flags & 8u) != 0,
                       (flags & 16u) != 0, pd);

    const double A = pd.conA, B = pd.conB, C = pd.conC;
    const double gA = -0.5 * s[2], gB = -s[3], gC = -0.5 * s[4];
    const double ggx = -(A * s[0] + B * s[1]);
    const double ggy = -(C * s[1] + B * s[0]);
    const double dm2x = ggx * 0.5 * (double)a.width, dm2y = ggy * 0.5 * (double)a.height;
    d_m2[0] = (float)dm2x;
    d_m2[1] = (float)dm2y;
    double dmean[3] = {0.0, 0.0, 0.0};

    // screen position -> clip space -> world
    const double dhx = dm2x * pd.pw, dhy = dm2y * pd.pw;
    const double dhw = -(dm2x * pd.hx + dm2y * pd.hy) * pd.pw * pd.pw;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      dmean[j] += (double)cam.pm[j * 4 + 0] * dhx + (double)cam.pm[j * 4 + 1] * dhy + (double)cam.pm[j * 4 + 3] * dhw;

    // conic -> 2D covariance
    const double a2 = pd.a, b2 = pd.b, c2 = pd.c;
    const double di2 = pd.di * pd.di;
    const double ga = (-c2 * c2 * gA + b2 * c2 * gB - b2 * b2 * gC) * di2;
    const double gb = (2.0 * b2 * c2 * gA - (a2 * c2 + b2 * b2) * gB + 2.0 * a2 * b2 * gC) * di2;
    const double gc = (-b2 * b2 * gA + a2 * b2 * gB - a2 * a2 * gC) * di2;
    const double hb = 0.5 * gb;
    // dL/dSigma (full symmetric matrix) = T^T G2 T
    double Gs[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        Gs[i][j] = pd.T0[i] * (ga * pd.T0[j] + hb * pd.T1[j]) + pd.T1[i] * (hb * pd.T0[j] + gc * pd.T1[j]);
    d_c3[0] = (float)Gs[0][0];
    d_c3[1] = (float)(2.0 * Gs[0][1]);
    d_c3[2] = (float)(2.0 * Gs[0][2]);
    d_c3[3] = (float)Gs[1][1];
    d_c3[4] = (float)(2.0 * Gs[1][2]);
    d_c3[5] = (float)Gs[2][2];
    // dL/dT = 2 G2 T Sigma = 2 G2 U
    double dT0[3], dT1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dT0[j] = 2.0 * ga * pd.U0[j] + gb * pd.U1[j];
      dT1[j] = gb * pd.U0[j] + 2.0 * gc * pd.U1[j];
    }
    double gJ00 = 0.0, gJ02 = 0.0, gJ11 = 0.0, gJ12 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      gJ00 += dT0[j] * (double)cam.vm[j * 4 + 0];
      gJ02 += dT0[j] * (double)cam.vm[j * 4 + 2];
      gJ11 += dT1[j] * (double)cam.vm[j * 4 + 1];
      gJ12 += dT1[j] * (double)cam.vm[j * 4 + 2];
    }
    const double itz = pd.itz, itz2 = itz * itz, itz3 = itz2 * itz;
    const double g_txc = -pd.fx * itz2 * gJ02;
    const double g_tyc = -pd.fy * itz2 * gJ12;
    double g_tz = -pd.fx * itz2 * gJ00 + 2.0 * pd.fx * pd.txc * itz3 * gJ02 - pd.fy * itz2 * gJ11 +
                  2.0 * pd.fy * pd.tyc * itz3 * gJ12;
    double g_tx = 0.0, g_ty = 0.0;
    if (flags & 8u) g_tz += g_txc * (pd.txc * itz); else g_tx = g_txc;
    if (flags & 16u) g_tz += g_tyc * (pd.tyc * itz); else g_ty = g_tyc;
    // inverse depth
    g_tz += -s[9] * itz2;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      dmean[j] += (double)cam.vm[j * 4 + 0] * g_tx + (double)cam.vm[j * 4 + 1] * g_ty + (double)cam.vm[j * 4 + 2] * g_tz;
    d_mean[0] = (float)dmean[0]; d_mean[1] = (float)dmean[1]; d_mean[2] = (float)dmean[2];

    // Sigma -> scale / rotation
    if (!a.cov3D_precomp) {
      const double* R = pd.R;
      const double* sv = pd.s;
      double dM[3][3];   // dL/dM, M_ik = R_ik s_k ; dL/dM = 2 Gs M
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < 3; ++j) acc += Gs[i][j] * (R[j * 3 + k] * sv[k]);
          dM[i][k] = 2.0 * acc;
        }
      double gR[3][3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          acc += R[i * 3 + k] * dM[i][k];
          gR[i][k] = dM[i][k] * sv[k];
        }
        double dsk = (double)a.scale_modifier * acc;
        if (a.activations & HGS_ACT_SCALE_EXP) dsk *= (double)sc[k];          // d exp(raw) / d raw = exp(raw)
        d_scale[k] = (float)dsk;
      }
      const double r = q[0], x = q[1], y = q[2], z = q[3];
      double dq[4];
      dq[0] = 2.0 * (-z * gR[0][1] + y * gR[0][2] + z * gR[1][0] - x * gR[1][2] - y * gR[2][0] + x * gR[2][1]);
      dq[1] = 2.0 * (y * gR[0][1] + z * gR[0][2] + y * gR[1][0] - 2.0 * x * gR[1][1] - r * gR[1][2] +
                     z * gR[2][0] + r * gR[2][1] - 2.0 * x * gR[2][2]);
      dq[2] = 2.0 * (-2.0 * y * gR[0][0] + x * gR[0][1] + r * gR[0][2] + x * gR[1][0] + z * gR[1][2] -
                     r * gR[2][0] + z * gR[2][1] - 2.0 * y * gR[2][2]);
      dq[3] = 2.0 * (-2.0 * z * gR[0][0] - r * gR[0][1] + x * gR[0][2] + r * gR[1][0] - 2.0 * z * gR[1][1] +
                     y * gR[1][2] + x * gR[2][0] + y * gR[2][1]);
      if (a.activations & HGS_ACT_ROT_NORMALIZE) {     // through q = raw / |raw|: (I - q q^T) / |raw|
        const double dot = r * dq[0] + x * dq[1] + y * dq[2] + z * dq[3];
        const double inv = 1.0 / qnorm;
        dq[0] = (dq[0] - r * dot) * inv; dq[1] = (dq[1] - x * dot) * inv;
        dq[2] = (dq[2] - y * dot) * inv; dq[3] = (dq[3] - z * dot) * inv;
      }
      d_rot[0] = (float)dq[0]; d_rot[1] = (float)dq[1]; d_rot[2] = (float)dq[2]; d_rot[3] = (float)dq[3];
    }

    // opacity (through the LOD remap)
    {
      float dod = 1.0f;
      double dact = 1.0;
      const float o_act = load_opacity<LOD>(a, idx, &dact);
      float o_rec = o_act;      // the opacity K1 put into the record (same functions, same float)
      if (a.interpolation_weights && a.num_node_kids && !a.lod_per_pixel)
        o_rec = lod_opacity(o_act, a.interpolation_weights[idx], a.num_node_kids[idx], &dod);
      // the instance records carry sum X = sum (o G) dL/dalpha; dL/do = sum G dL/dalpha = (sum X) / o.  o <= 0: never
      // blended, sum X = 0
      // (exact division, once per Gaussian: rcp_d seeds from v_rcp_f32, which overflows for a denormal opacity -- the
      // Newton step would then turn 0 x inf into a NaN that poisons the optimizer state)
      const float sums5 = o_rec > 0.0f ? (float)(s[5] / (double)o_rec) : 0.0f;
      d_op = a.activations ? (float)((double)sums5 * (double)dod * dact) : sums5 * dod;
    }

    // colour: precomputed colours get their gradient here; SH colours hand the clamp-masked dL/drgb to K8b
    float gr[3] = {sums6, sums7, sums8};
    if (a.colors_precomp) {
      d_col[0] = gr[0]; d_col[1] = gr[1]; d_col[2] = gr[2];
    } else {
      if (flags & 1u) gr[0] = 0.f;
      if (flags & 2u) gr[1] = 0.f;
      if (flags & 4u) gr[2] = 0.f;
      drgb[idx * 3 + 0] = gr[0];
      drgb[idx * 3 + 1] = gr[1];
      drgb[idx * 3 + 2] = gr[2];
    }
  }

  if constexpr (LOD) {
    if (a.lod_scatter) {
      // scales, rotations, opacity: scattered here; the mean's gradient goes on to K8b (which adds the view-direction
      // term and scatters it together with the SH gradients)
      __shared__ float lds_v[8 * kPreBlock];
      __shared__ int lds_par[kPreBlock];
      __shared__ unsigned char lds_vis[kPreBlock];
      const int t = threadIdx.x;
      const int block_first = blockIdx.x * kPreBlock;
      const int count = min(kPreBlock, a.P - block_first);
      LodRow l;
      l.r = l.p = 0; l.w = 1.0f; l.u = 0.0f;
      if (in_range) l = lod_row<true>(a, idx);
      const bool self = l.p == l.r;
      const bool vis = in_range && n != 0;
      const float wn = self ? 1.0f : l.w, u = (self || !vis) ? 0.0f : l.u;
      float sgn = 1.0f;
      if (vis && !self && u != 0.0f) {          // (weight 1: nothing goes to the parent, its quaternion is not needed)
        const float4 qa = reinterpret_cast<const float4*>(a.rotations)[l.r];
        const float4 qb = reinterpret_cast<const float4*>(a.rotations)[l.p];
        sgn = (qa.x * qb.x + qa.y * qb.y + qa.z * qb.z + qa.w * qb.w) < 0.0f ? -1.0f : 1.0f;
      }
      if (in_range) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          dmean_rows[idx * 3 + j] = d_mean[j];
          out.dL_dmeans2D[idx * 3 + j] = d_m2[j];
        }
      }
      if (vis) {      // node row (culled rows have zero gradients: the caller's zero fill stands)
        out.dL_dopacity[l.r] = wn * d_op;
#pragma unroll
        for (int j = 0; j < 3; ++j) out.dL_dscales[l.r * 3 + j] = wn * d_scale[j];
        reinterpret_cast<float4*>(out.dL_drotations)[l.r] = make_float4(wn * d_rot[0], wn * d_rot[1], wn * d_rot[2], wn * d_rot[3]);
      }
      lds_par[t] = in_range ? (int)l.p : -1;
      lds_vis[t] = (vis && !self) ? 1 : 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) lds_v[j * kPreBlock + t] = u * d_scale[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_v[(3 + j) * kPreBlock + t] = (u * sgn) * d_rot[j];
      lds_v[7 * kPreBlock + t] = u * d_op;
      __syncthreads();
      if (t < count) {
        const LodRun run = lod_run(a, lds_par, t, count, block_first, *lod_flag != 0u);
        if (run.leader) {
          float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          bool any = false;
          for (int j = t; j < run.end; ++j) {
            if (!lds_vis[j]) continue;
            any = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc8[k] += lds_v[k * kPreBlock + j];
          }
          if (any) {
            const size_t pp = (size_t)lds_par[t];
#pragma unroll
            for (int k = 0; k < 3; ++k) lod_add(out.dL_dscales + pp * 3 + k, acc8[k], run.atomic);
#pragma unroll
            for (int k = 0; k < 4; ++k) lod_add(out.dL_drotations + pp * 4 + k, acc8[3 + k], run.atomic);
            lod_add(out.dL_dopacity + pp, acc8[7], run.atomic);
          }
        }
      }
      return;
    }
    if (!in_range) return;
  }
  // accumulate_grads: add to what the buffers hold (gradient accumulation over several views of one optimizer
  // step); dL/dmeans2D (a per-view statistic) and dL/dcolors_precomp (the gradient of a view's colours) are always
  // overwritten
  constexpr bool acc = ACC;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    out.dL_dmeans3D[idx * 3 + j] = d_mean[j] + (acc ? out.dL_dmeans3D[idx * 3 + j] : 0.f);
    out.dL_dmeans2D[idx * 3 + j] = d_m2[j];
  }
  out.dL_dopacity[idx] = d_op + (acc ? out.dL_dopacity[idx] : 0.f);
  if (out.dL_dcolors) {
#pragma unroll
    for (int j = 0; j < 3; ++j) out.dL_dcolors[idx * 3 + j] = d_col[j];   // per view, like dL/dmeans2D: never accumulated
  }
  if (out.dL_dscales) {
#pragma unroll
    for (int j = 0; j < 3; ++j) out.dL_dscales[idx * 3 + j] = d_scale[j] + (acc ? out.dL_dscales[idx * 3 + j] : 0.f);
  }
  if (out.dL_drotations) {
    float4* dst = reinterpret_cast<float4*>(out.dL_drotations) + idx;
    float4 v = make_float4(d_rot[0], d_rot[1], d_rot[2], d_rot[3]);
    if (acc) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *dst = v;
  }
  if (out.dL_dcov3D) {
#pragma unroll
    for (int j = 0; j < 6; ++j) out.dL_dcov3D[(size_t)idx * 6 + j] = d_c3[j] + (acc ? out.dL_dcov3D[(size_t)idx * 6 + j] : 0.f);
  }
}


// K8b: SH part of the backward.  Pure streaming kernel (192 B of coefficients in, 192 B of gradients out
// per Gaussian at M = 16), split from the double-precision geometry chain of K8a so that it runs at high
// occupancy.  Adds the view-direction term to dL/dmeans3D written by K8a.
template <bool ACC, bool JAC, bool LOD>   // JAC: d(rgb)/d(direction) was stored by K1 (prepare_backward): the coefficients are not read
__global__ __launch_bounds__(kPreBlock) void sh_bwd_kernel(hgs_raster_args a, GeomWs g,
                                                           const float* __restrict__ drgb,
                                                           const float* __restrict__ dmean_rows,
                                                           const uint32_t* __restrict__ lod_flag, hgs_raster_grads out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int n = a.M * 3;
  const bool split = a.shs_rest != nullptr;       // features_dc / features_rest as two tensors
  const bool coop = (n & 3) == 0;                 // 16-byte granules (M = 16, 4, ...); else per-lane access
  const int block_first = blockIdx.x * kPreBlock;
  const int idx = block_first + threadIdx.x;
  const bool valid = idx < a.P;
  const bool active = valid && g.tiles_touched[idx] != 0;
  if (!JAC && coop) {
    if (split) {
      coop_load_seg(a.shs, block_first, a.P, 3, 0, sh_row_stride(n), lds);
      coop_load_seg(a.shs_rest, block_first, a.P, n - 3, 3, sh_row_stride(n), lds);
    } else {
      coop_load_sh(a.shs, block_first, a.P, n, lds);
    }
    __syncthreads();
  }
  float dsh[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) dsh[i] = 0.f;
  float sh[JAC ? 1 : 48];
  if constexpr (!JAC) {
    if (active) {
      if (coop) lds_row_read(lds, n, sh);
      else if (split) load_sh_split(a.shs, a.shs_rest, idx, a.M, sh);
      else load_sh(a.shs, idx, a.M, sh);
    }
  }
  if (!JAC && coop) __syncthreads();              // every row has been read: the buffer becomes the output stage
  float vdir[3] = {0.f, 0.f, 0.f};                // view-direction term of dL/dmean
  if (active) {
    const float gr[3] = {drgb[idx * 3 + 0], drgb[idx * 3 + 1], drgb[idx * 3 + 2]};
    float pm[3];
    load_mean<LOD>(a, lod_row_gather<LOD>(a, idx), pm);
    float ux, uy, uz;
    const float inv = unit_dir(pm, a.campos, ux, uy, uz);
    float b[16];
    sh_basis(a.sh_degree, ux, uy, uz, b);
    const int nb = (a.sh_degree + 1) * (a.sh_degree + 1);
    float gdx = 0.f, gdy = 0.f, gdz = 0.f;
    if constexpr (JAC) {
      const float* jp = g.shjac + (size_t)idx * kJacStride;
      float J[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) J[k] = jp[k];
      gdx = gr[0] * J[0] + gr[1] * J[1] + gr[2] * J[2];
      gdy = gr[0] * J[3] + gr[1] * J[4] + gr[2] * J[5];
      gdz = gr[0] * J[6] + gr[1] * J[7] + gr[2] * J[8];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k < nb) { dsh[k * 3 + 0] = b[k] * gr[0]; dsh[k * 3 + 1] = b[k] * gr[1]; dsh[k * 3 + 2] = b[k] * gr[2]; }
      }
    } else {
      float dbx[16], dby[16], dbz[16];
      sh_basis_grad(a.sh_degree, ux, uy, uz, dbx, dby, dbz);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k < nb) {
          dsh[k * 3 + 0] = b[k] * gr[0]; dsh[k * 3 + 1] = b[k] * gr[1]; dsh[k * 3 + 2] = b[k] * gr[2];
          const float dotc = gr[0] * sh[k * 3 + 0] + gr[1] * sh[k * 3 + 1] + gr[2] * sh[k * 3 + 2];
          gdx += dbx[k] * dotc; gdy += dby[k] * dotc; gdz += dbz[k] * dotc;
        }
      }
    }
    // through the normalisation dir = d/|d|
    const float dot = ux * gdx + uy * gdy + uz * gdz;
    vdir[0] = (gdx - ux * dot) * inv; vdir[1] = (gdy - uy * dot) * inv; vdir[2] = (gdz - uz * dot) * inv;
    if (!(LOD && a.lod_scatter)) {
      out.dL_dmeans3D[idx * 3 + 0] += vdir[0];
      out.dL_dmeans3D[idx * 3 + 1] += vdir[1];
      out.dL_dmeans3D[idx * 3 + 2] += vdir[2];
    }
  }
  if constexpr (LOD) {
    if (a.lod_scatter) {
      // scatter of the SH and mean gradients (see lod_run): rows staged in LDS (the four pad floats of a row carry its
      // node row, parent row, weight and "on screen"), every 16 bytes of a row handled by one lane
      __shared__ float lds_m[3 * kPreBlock];
      __shared__ int lds_par[kPreBlock];
      const int t = threadIdx.x;
      const int count = min(kPreBlock, a.P - block_first);
      const int stride = sh_row_stride(n), cpr = n >> 2;
      LodRow l;
      l.r = l.p = 0; l.w = 1.0f; l.u = 0.0f;
      if (valid) l = lod_row<true>(a, idx);
      if (valid) {
        lds_row_write(lds, n, dsh);
        float* pad = lds + t * stride + n;
        pad[0] = __uint_as_float((uint32_t)l.r); pad[1] = __uint_as_float((uint32_t)l.p); pad[2] = l.w;
        pad[3] = __uint_as_float(active ? 1u : 0u);
#pragma unroll
        for (int j = 0; j < 3; ++j) lds_m[j * kPreBlock + t] = active ? dmean_rows[idx * 3 + j] + vdir[j] : 0.0f;
      }
      lds_par[t] = valid ? (int)l.p : -1;
      __syncthreads();
      const bool nonmono = *lod_flag != 0u;
      float4* dsh_full = reinterpret_cast<float4*>(out.dL_dshs);
      for (int v = t; v < count * cpr; v += kPreBlock) {
        const int row = v / cpr, c = v - row * cpr;
        const float* rp = lds + row * stride;
        const uint32_t r = __float_as_uint(rp[n]), pr = __float_as_uint(rp[n + 1]);
        const bool self = r == pr;
        if (__float_as_uint(rp[n + 3])) {                         // node row: plain store
          const float wn = self ? 1.0f : rp[n + 2];
          const float4 g4 = *reinterpret_cast<const float4*>(rp + c * 4);
          dsh_full[(size_t)r * cpr + c] = make_float4(wn * g4.x, wn * g4.y, wn * g4.z, wn * g4.w);
        }
        const LodRun run = lod_run(a, lds_par, row, count, block_first, nonmono);
        if (run.leader) {
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          bool any = false;
          for (int j = row; j < run.end; ++j) {
            const float* rj = lds + j * stride;
            if (!__float_as_uint(rj[n + 3]) || __float_as_uint(rj[n]) == __float_as_uint(rj[n + 1])) continue;
            any = true;
            const float uj = 1.0f - rj[n + 2];
            const float4 g4 = *reinterpret_cast<const float4*>(rj + c * 4);
            acc.x += uj * g4.x; acc.y += uj * g4.y; acc.z += uj * g4.z; acc.w += uj * g4.w;
          }
          if (any) {
            float* dst = reinterpret_cast<float*>(dsh_full + (size_t)pr * cpr + c);
            if (run.atomic) {
              atomicAdd(dst + 0, acc.x); atomicAdd(dst + 1, acc.y); atomicAdd(dst + 2, acc.z); atomicAdd(dst + 3, acc.w);
            } else {
              *reinterpret_cast<float4*>(dst) = acc;
            }
          }
        }
      }
      if (t < count) {                                            // the mean: one lane per row
        const float* rp = lds + t * stride;
        const uint32_t r = __float_as_uint(rp[n]), pr = __float_as_uint(rp[n + 1]);
        const bool self = r == pr;
        if (__float_as_uint(rp[n + 3])) {
          const float wn = self ? 1.0f : rp[n + 2];
#pragma unroll
          for (int j = 0; j < 3; ++j) out.dL_dmeans3D[(size_t)r * 3 + j] = wn * lds_m[j * kPreBlock + t];
        }
        const LodRun run = lod_run(a, lds_par, t, count, block_first, nonmono);
        if (run.leader) {
          float acc3[3] = {0.f, 0.f, 0.f};
          bool any = false;
          for (int j = t; j < run.end; ++j) {
            const float* rj = lds + j * stride;
            if (!__float_as_uint(rj[n + 3]) || __float_as_uint(rj[n]) == __float_as_uint(rj[n + 1])) continue;
            any = true;
            const float uj = 1.0f - rj[n + 2];
#pragma unroll
            for (int k = 0; k < 3; ++k) acc3[k] += uj * lds_m[k * kPreBlock + j];
          }
          if (any) {
#pragma unroll
            for (int k = 0; k < 3; ++k) lod_add(out.dL_dmeans3D + (size_t)pr * 3 + k, acc3[k], run.atomic);
          }
        }
      }
      return;
    }
  }
  if (coop) {
    if (valid) lds_row_write(lds, n, dsh);
    __syncthreads();
    if (split) {
      coop_store_seg<ACC>(out.dL_dshs, block_first, a.P, 3, 0, sh_row_stride(n), lds);
      coop_store_seg<ACC>(out.dL_dshs_rest, block_first, a.P, n - 3, 3, sh_row_stride(n), lds);
    } else {
      coop_store_sh<ACC>(out.dL_dshs, block_first, a.P, n, lds);
    }
  } else if (valid && split) {
    float* d0 = out.dL_dshs + (size_t)idx * 3;
    float* d1 = out.dL_dshs_rest + (size_t)idx * (n - 3);
#pragma unroll
    for (int i = 0; i < 48; ++i) {       // static indices keep dsh[] in registers
      if (i < 3) d0[i] = dsh[i] + (ACC ? d0[i] : 0.f);
      else if (i < n) d1[i - 3] = dsh[i] + (ACC ? d1[i - 3] : 0.f);
    }
  } else if (valid) {
    if (ACC) {
      const float* old = out.dL_dshs + (size_t)idx * n;
#pragma unroll
      for (int i = 0; i < 48; ++i)       // static indices keep dsh[] in registers
        if (i < n) dsh[i] += old[i];
    }
    store_sh(out.dL_dshs, idx, a.M, dsh);
  }
}

// K8b for several views of the same Gaussians at once (hgs_raster_sh_bwd_batched): the SH block is read once and
// the gradient block written once for all views -- per view this kernel's traffic is otherwise the largest of the
// streaming kernels (384 B per Gaussian).
// COOP: 3M % 4 == 0, the SH block goes through LDS, else per-lane access.  COLOR: the views come from the batched
// colour route (gradient w.r.t. the clamped colour + clamp mask) instead of deferred raster backwards (masked
// gradient + visibility).
template <bool ACC, bool COOP, bool COLOR>
__device__ __forceinline__ void sh_bwd_batched_body(const ShBwdViews& v, int P, int M, int sh_degree,
                                                    const float* __restrict__ means3D, const float* __restrict__ shs,
                                                    float* __restrict__ dL_dshs, float* __restrict__ dL_dmeans3D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int n = M * 3;
  constexpr bool coop = COOP;
  const int block_first = blockIdx.x * kPreBlock;
  const int idx = block_first + threadIdx.x;
  const bool valid = idx < P;
  if (coop) {
    coop_load_sh(shs, block_first, P, n, lds);
    __syncthreads();
  }
  // the coefficients stay in this lane's LDS row and are re-read per view (keeping them in registers next to the
  // 48 accumulators and the basis arrays costs a wave of occupancy)
  float sh_reg[COOP ? 1 : 48];
  if constexpr (!COOP) { if (valid) load_sh(shs, idx, M, sh_reg); }
  float dsh[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) dsh[i] = 0.f;
  float gm0 = 0.f, gm1 = 0.f, gm2 = 0.f;
  if (valid) {
    const float px = means3D[idx * 3 + 0], py = means3D[idx * 3 + 1], pz = means3D[idx * 3 + 2];
    const int nb = (sh_degree + 1) * (sh_degree + 1);
    for (int w = 0; w < v.n; ++w) {
      if (!COLOR && static_cast<const uint32_t*>(v.mask[w])[idx] == 0) continue;
      float gr0 = v.drgb[w][idx * 3 + 0], gr1 = v.drgb[w][idx * 3 + 1], gr2 = v.drgb[w][idx * 3 + 2];
      if (COLOR) {            // the incoming gradient is w.r.t. the clamped colour
        const uint32_t m = static_cast<const uint8_t*>(v.mask[w])[idx];
        if (m & 1u) gr0 = 0.f;
        if (m & 2u) gr1 = 0.f;
        if (m & 4u) gr2 = 0.f;
        if (gr0 == 0.f && gr1 == 0.f && gr2 == 0.f) continue;      // culled / not contributing in this view
      }
      // dotc[k] = gr . c_k straight from the LDS row (or the registers of the per-lane path): no 48-float copy
      float dotc[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) dotc[k] = 0.f;
      if constexpr (COOP) {
        const float4* row4 = reinterpret_cast<const float4*>(lds + threadIdx.x * sh_row_stride(n));
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          if (i * 4 < n) {
            const float4 t = row4[i];
            const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int ch = (i * 4 + c) % 3;
              dotc[(i * 4 + c) / 3] += (ch == 0 ? gr0 : ch == 1 ? gr1 : gr2) * tv[c];
            }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 48; ++i) dotc[i / 3] += (i % 3 == 0 ? gr0 : i % 3 == 1 ? gr1 : gr2) * sh_reg[i];
      }
      const float pxyz[3] = {px, py, pz};
      float ux, uy, uz;
      const float inv = unit_dir(pxyz, v.campos[w], ux, uy, uz);
      {
        float b[16];
        sh_basis(sh_degree, ux, uy, uz, b);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (k < nb) { dsh[k * 3 + 0] += b[k] * gr0; dsh[k * 3 + 1] += b[k] * gr1; dsh[k * 3 + 2] += b[k] * gr2; }
        }
      }
      float gdx = 0.f, gdy = 0.f, gdz = 0.f;
      {
        float dbx[16], dby[16], dbz[16];
        sh_basis_grad(sh_degree, ux, uy, uz, dbx, dby, dbz);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (k < nb) { gdx += dbx[k] * dotc[k]; gdy += dby[k] * dotc[k]; gdz += dbz[k] * dotc[k]; }
        }
      }
      const float dot = ux * gdx + uy * gdy + uz * gdz;
      gm0 += (gdx - ux * dot) * inv;
      gm1 += (gdy - uy * dot) * inv;
      gm2 += (gdz - uz * dot) * inv;
    }
    dL_dmeans3D[idx * 3 + 0] += gm0;
    dL_dmeans3D[idx * 3 + 1] += gm1;
    dL_dmeans3D[idx * 3 + 2] += gm2;
  }
  if (coop) {
    __syncthreads();            // every lane is done reading its coefficients: the buffer becomes the output stage
    if (valid) lds_row_write(lds, n, dsh);
    __syncthreads();
    coop_store_sh<ACC>(dL_dshs, block_first, P, n, lds);
  } else if (valid) {
    if (ACC) {
      const float* old = dL_dshs + (size_t)idx * n;
#pragma unroll
      for (int i = 0; i < 48; ++i)
        if (i < n) dsh[i] += old[i];
    }
    store_sh(dL_dshs, idx, M, dsh);
  }
}

}  // namespace

template <bool ACC, bool COOP>
__global__ __launch_bounds__(kPreBlock) void sh_bwd_batched_kernel(ShBwdViews v, int P, int M, int sh_degree,
                                                                   const float* __restrict__ means3D,
                                                                   const float* __restrict__ shs,
                                                                   float* __restrict__ dL_dshs,
                                                                   float* __restrict__ dL_dmeans3D) {
  sh_bwd_batched_body<ACC, COOP, false>(v, P, M, sh_degree, means3D, shs, dL_dshs, dL_dmeans3D);
}
template <bool ACC, bool COOP>     // 3 workgroups per CU = what the LDS stage allows: keeps the registers under 170
__global__ __launch_bounds__(kPreBlock, 3) void sh_bwd_batched_color_kernel(ShBwdViews v, int P, int M, int sh_degree,
                                                                            const float* __restrict__ means3D,
                                                                            const float* __restrict__ shs,
                                                                            float* __restrict__ dL_dshs,
                                                                            float* __restrict__ dL_dmeans3D) {
  sh_bwd_batched_body<ACC, COOP, true>(v, P, M, sh_degree, means3D, shs, dL_dshs, dL_dmeans3D);
}

// Colours of one block of Gaussians for every view: SH block through LDS once, basis per view.
__global__ __launch_bounds__(kPreBlock) void sh_colors_batched_kernel(ShFwdViews v, int P, int M, int sh_degree,
                                                                      const float* __restrict__ means3D,
                                                                      const float* __restrict__ shs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int n = M * 3;
  const bool coop = (n & 3) == 0;
  const int block_first = blockIdx.x * kPreBlock;
  const int idx = block_first + threadIdx.x;
  if (coop) {
    coop_load_sh(shs, block_first, P, n, lds);
    __syncthreads();
  }
  if (idx >= P) return;
  float sh[48];
  if (coop) lds_row_read(lds, n, sh); else load_sh(shs, idx, M, sh);
  const float px = means3D[idx * 3 + 0], py = means3D[idx * 3 + 1], pz = means3D[idx * 3 + 2];
  const int nb = (sh_degree + 1) * (sh_degree + 1);
  for (int w = 0; w < v.n; ++w) {
    const float pxyz[3] = {px, py, pz};
    float dx, dy, dz;
    unit_dir(pxyz, v.campos[w], dx, dy, dz);
    float b[16];
    sh_basis(sh_degree, dx, dy, dz, b);
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < nb) {
        r0 += b[k] * sh[k * 3 + 0];
        r1 += b[k] * sh[k * 3 + 1];
        r2 += b[k] * sh[k * 3 + 2];
      }
    }
    r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
    uint32_t m = 0;
    if (r0 < 0.f) { r0 = 0.f; m |= 1u; }
    if (r1 < 0.f) { r1 = 0.f; m |= 2u; }
    if (r2 < 0.f) { r2 = 0.f; m |= 4u; }
    v.rgb_out[w][idx * 3 + 0] = r0;
    v.rgb_out[w][idx * 3 + 1] = r1;
    v.rgb_out[w][idx * 3 + 2] = r2;
    v.clamp_out[w][idx] = (uint8_t)m;
  }
}

int launch_sh_colors_batched(const ShFwdViews& v, int32_t P, int32_t M, int32_t sh_degree, const float* means3D,
                             const float* shs, hipStream_t s) {
  const int nblk = (P + kPreBlock - 1) / kPreBlock;
  if (nblk <= 0) return HGS_OK;
  const size_t lds_bytes = (size_t)kPreBlock * (M * 3 + 4) * sizeof(float);
  hipLaunchKernelGGL(sh_colors_batched_kernel, dim3(nblk), dim3(kPreBlock), lds_bytes, s, v, P, M, sh_degree, means3D, shs);
  HGS_LAUNCH_CHECK("sh_colors_batched", s, false);
  return HGS_OK;
}

int launch_sh_bwd_batched(const ShBwdViews& v, int32_t P, int32_t M, int32_t sh_degree, const float* means3D,
                          const float* shs, float* dL_dshs, float* dL_dmeans3D, bool accumulate, hipStream_t s) {
  const int nblk = (P + kPreBlock - 1) / kPreBlock;
  if (nblk <= 0) return HGS_OK;
  const size_t lds_bytes = (size_t)kPreBlock * (M * 3 + 4) * sizeof(float);
  const bool coop = ((M * 3) & 3) == 0;
  const bool color = v.color != 0;
  auto kern = color ? (coop ? (accumulate ? sh_bwd_batched_color_kernel<true, true> : sh_bwd_batched_color_kernel<false, true>)
                            : (accumulate ? sh_bwd_batched_color_kernel<true, false> : sh_bwd_batched_color_kernel<false, false>))
                    : (coop ? (accumulate ? sh_bwd_batched_kernel<true, true> : sh_bwd_batched_kernel<false, true>)
                            : (accumulate ? sh_bwd_batched_kernel<true, false> : sh_bwd_batched_kernel<false, false>));
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(kPreBlock), lds_bytes, s, v, P, M, sh_degree, means3D, shs, dL_dshs,
                     dL_dmeans3D);
  HGS_LAUNCH_CHECK("sh_bwd_batched", s, false);
  return HGS_OK;
}

// Zeroed superblock totals of K1 (above), one block per (device, stream): launches on one stream run one after the other
// and the counting kernel behind K3 leaves the block at zero, so it is zeroed by the host only when it is created -- or
// when an error return between K1 and that kernel left it `dirty`.  At most kMaxSyncBlocks distinct streams are served;
// further ones (and HGS_SCAN_LAUNCH=1) take the separate scan launch.
namespace {
constexpr int kMaxSyncBlocks = 64;
struct SyncBlock { int device; hipStream_t stream; uint32_t* words; bool dirty; };
std::mutex g_sync_mu;
SyncBlock g_sync[kMaxSyncBlocks];
int g_sync_n = 0;
}  // namespace
size_t super_block_bytes() { return (size_t)(2 + kBands) * kMaxSuper * sizeof(uint32_t); }   // 9 rows of totals + the heavy list
uint32_t* super_block_acquire(hipStream_t s) {
  static const bool off = getenv("HGS_SCAN_LAUNCH") != nullptr;
  if (off) return nullptr;
  // hipStreamPerThread is ONE handle for a different stream in every host thread: two threads' frames would share a block
  // and interleave their sums.  A capturing stream must not see the allocation / memset of a first use.  Both take the
  // scan launch.
  if (s == hipStreamPerThread) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (cap != hipStreamCaptureStatusNone) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lk(g_sync_mu);
  for (int i = 0; i < g_sync_n; ++i) {
    SyncBlock& b = g_sync[i];
    if (b.device != dev || b.stream != s) continue;
    if (b.dirty) {
      if (hipMemsetAsync(b.words, 0, super_block_bytes(), s) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      b.dirty = false;
    }
    return b.words;
  }
  if (g_sync_n == kMaxSyncBlocks) return nullptr;
  uint32_t* w = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&w), super_block_bytes()) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemsetAsync(w, 0, super_block_bytes(), s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(w); return nullptr; }
  g_sync[g_sync_n++] = {dev, s, w, false};
  return w;
}
// Frees the blocks of `device` (all devices: < 0).  The caller guarantees that no call of this library is in flight there.
int super_block_release(int device) {
  std::lock_guard<std::mutex> lk(g_sync_mu);
  int kept = 0, freed = 0;
  for (int i = 0; i < g_sync_n; ++i) {
    if (device < 0 || g_sync[i].device == device) {
      int cur = 0;
      const bool sw = hipGetDevice(&cur) == hipSuccess && cur != g_sync[i].device && hipSetDevice(g_sync[i].device) == hipSuccess;
      (void)hipFree(g_sync[i].words);
      if (sw) (void)hipSetDevice(cur);
      ++freed;
    } else {
      g_sync[kept++] = g_sync[i];
    }
  }
  g_sync_n = kept;
  return freed;
}
void super_block_mark_dirty(const uint32_t* words) {
  std::lock_guard<std::mutex> lk(g_sync_mu);
  for (int i = 0; i < g_sync_n; ++i)
    if (g_sync[i].words == words) g_sync[i].dirty = true;
}

int launch_preprocess_fwd(const hgs_raster_args& a, const GeomWs& g, int32_t* radii, hipStream_t s, uint32_t* super,
                          uint32_t heavy_thr) {
  const int nblk = (a.P + kPreBlock - 1) / kPreBlock;
  if (nblk > 0) {
    const bool jac = a.prepare_backward && a.shs;
    const bool plain = a.shs && !a.shs_rest && !a.lod_render_indices && ((a.M * 3) & 3) == 0;
    const bool h48 = plain && a.M == 16;
    const bool defer = plain && !h48;
    const size_t lds_bytes = !a.shs ? 0 : h48 ? (size_t)kK1ImageBytes : (size_t)kPreBlock * (a.M * 3 + 4) * sizeof(float);
    auto k1 = a.lod_render_indices ? (jac ? preprocess_fwd_kernel<true, true, false> : preprocess_fwd_kernel<false, true, false>)
              : h48   ? (jac ? preprocess_fwd_h48_kernel<true> : preprocess_fwd_h48_kernel<false>)
              : defer ? (jac ? preprocess_fwd_kernel<true, false, true> : preprocess_fwd_kernel<false, false, true>)
                      : (jac ? preprocess_fwd_kernel<true, false, false> : preprocess_fwd_kernel<false, false, false>);
    hipLaunchKernelGGL(k1, dim3(nblk), dim3(kPreBlock), lds_bytes, s, a, g, radii, super, heavy_thr);
    HGS_LAUNCH_CHECK("preprocess_fwd", s, a.debug);
  }
  return HGS_OK;
}

int launch_scan_block_sums(const GeomWs& g, int32_t P, hipStream_t s, bool debug, uint32_t* total_mirror) {
  const int nblk = (P + kPreBlock - 1) / kPreBlock;
  const int chunks = scan_chunks(nblk), resident = scan_resident_workgroups();
  if (chunks * (1 + kBands) <= resident && !scan_split_forced()) {
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(chunks, 1 + kBands), dim3(1024), 0, s, g.block_sums, g.block_band,
                       nblk, g.scan_chain, total_mirror, 0, chunks);
    HGS_LAUNCH_CHECK("scan_block_sums", s, debug);
    return HGS_OK;
  }
  // one launch per array (the kernel's row 0 scans its first pointer) and per `resident` chunks of it: a chunk only looks
  // back at chunks of its own launch -- resident together with it -- or of launches that have completed.  (A CU mask in
  // the environment makes `resident` 1: one launch per chunk, slow and correct.)
  for (int y = 0; y < 1 + kBands; ++y) {
    uint32_t* sums = y == 0 ? g.block_sums : g.block_band + (size_t)(y - 1) * (nblk + 1);
    for (int c0 = 0; c0 < chunks; c0 += resident) {
      hipLaunchKernelGGL(scan_block_sums_kernel, dim3(min(resident, chunks - c0), 1), dim3(1024), 0, s, sums, g.block_band,
                         nblk, g.scan_chain + (size_t)y * chunks, y == 0 ? total_mirror : nullptr, c0, chunks);
      HGS_LAUNCH_CHECK("scan_block_sums", s, debug);
    }
  }
  return HGS_OK;
}

int launch_preprocess_bwd(const hgs_raster_args& a, const GeomWs& g, const float* inst_grads, float* drgb,
                          float* dmean_rows, const uint32_t* lod_flag, const hgs_raster_grads& out, uint32_t L,
                          uint2* work, uint32_t* work_counter, hipStream_t s) {
  const int nblk = (a.P + kPreBlock - 1) / kPreBlock;
  if (nblk > 0) {
    // long runs first, when the frame has them: mean run above 6 records (L > 6 P; HGS_K8_PRESUM=0 / 1 forces)
    static const char* force = getenv("HGS_K8_PRESUM");
    const bool presum = force ? force[0] == '1' : (uint64_t)L > 6ull * (uint64_t)a.P;
    if (presum) {
      HGS_HIP(hipMemsetAsync(work_counter, 0, sizeof(uint32_t), s));
      hipLaunchKernelGGL(k8_worklist_kernel, dim3(nblk < kWorklistGrid ? nblk : kWorklistGrid), dim3(kPreBlock), 0, s, a.P,
                         g.tiles_touched, work, work_counter);
      HGS_LAUNCH_CHECK("preprocess_bwd_worklist", s, a.debug);
      const int want = (int)(((size_t)L / 64 + 3) / 4);                   // no more workgroups than a pair each could use
      hipLaunchKernelGGL(k8_presum_work_kernel, dim3(want < 1 ? 1 : (want < kPresumGrid ? want : kPresumGrid)), dim3(kPreBlock), 0,
                         s, work, work_counter, g.tiles_touched, g.offsets, const_cast<float*>(inst_grads));
      HGS_LAUNCH_CHECK("preprocess_bwd_long_runs", s, a.debug);
    }
    auto k8a = a.lod_render_indices ? preprocess_bwd_kernel<false, true>      // (accumulation is refused with lod, abi.cpp)
                                    : (a.accumulate_grads ? preprocess_bwd_kernel<true, false> : preprocess_bwd_kernel<false, false>);
    hipLaunchKernelGGL(k8a, dim3(nblk), dim3(kPreBlock), 0, s, a, g, inst_grads, drgb, dmean_rows, lod_flag, out,
                       presum ? 1 : 0);
    HGS_LAUNCH_CHECK("preprocess_bwd", s, a.debug);
    if (a.shs && out.dL_dshs && !a.defer_sh_bwd) {
      const size_t lds_bytes = (size_t)kPreBlock * (a.M * 3 + 4) * sizeof(float);
      auto k8b = a.lod_render_indices ? sh_bwd_kernel<false, true, true>
                 : (a.prepare_backward && a.shs) ? (a.accumulate_grads ? sh_bwd_kernel<true, true, false> : sh_bwd_kernel<false, true, false>)
                                                 : (a.accumulate_grads ? sh_bwd_kernel<true, false, false> : sh_bwd_kernel<false, false, false>);
      hipLaunchKernelGGL(k8b, dim3(nblk), dim3(kPreBlock), lds_bytes, s, a, g, drgb, dmean_rows, lod_flag, out);
      HGS_LAUNCH_CHECK("sh_bwd", s, a.debug);
    }
  }
  return HGS_OK;
}

}  // namespace hgs
