// K12: in-op LOD attribute interpolation.
//
// Restates, as two kernels, the Python block gaussian_renderer/__init__.py:199-218 that render_post
// runs before every hierarchy-mode rasterization (about 25 torch launches + autograd scatter-adds):
//     attr_i = w_i * attr[render_indices[i]] + (1 - w_i) * attr[parent_indices[i]]
// for means / scales / SH / opacity, and for rotations with the parent quaternion flipped into the
// node's hemisphere first (:212-216).  Used when GaussianRasterizationSettings.render_indices /
// parent_indices are passed NON-empty (the reference's settings fields exist for exactly this; its own
// glue always passes them empty, so this path is reached only by callers that opt in).
//
// Backward: node rows are unique in an LOD cut (plain stores); a parent row collects up to k sibling
// contributions.  Siblings are adjacent in expand_to_size's output (ascending node index, children
// contiguous), so one lane per run sums the run in registers; if parent_indices is non-decreasing every
// parent has exactly one run and the leader stores without atomics (deterministic), otherwise leaders fall
// back to float atomics (still one atomic per run and value, not per node).
#include "common.h"

namespace hgs {
namespace {

struct LodPtrs {
  const float* means;   // [G,3]
  const float* scales;  // [G,3]
  const float* rots;    // [G,4]
  const float* shs;     // [G,M,3]
  const float* opac;    // [G]
};
struct LodOut {
  float* means;
  float* scales;
  float* rots;
  float* shs;
  float* opac;
};

__global__ __launch_bounds__(256) void lod_gather_kernel(const int32_t* __restrict__ render_indices,
                                                         const int32_t* __restrict__ parent_indices,
                                                         const float* __restrict__ weights, int n, int M,
                                                         LodPtrs in, LodOut out) {
  // contraction off: w * a + u * b rounds exactly like the reference glue's torch expression t * x[r] + (1 - t) * x[p]
  // (two rounded products, one rounded sum), so the in-op path feeds the rasterizer bit-identical rows
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t r = (size_t)render_indices[i], p = (size_t)parent_indices[i];
  const float w = weights[i], u = 1.0f - w;
  if (in.means) {
#pragma unroll
    for (int k = 0; k < 3; ++k) out.means[(size_t)i * 3 + k] = w * in.means[r * 3 + k] + u * in.means[p * 3 + k];
  }
  if (in.scales) {
#pragma unroll
    for (int k = 0; k < 3; ++k) out.scales[(size_t)i * 3 + k] = w * in.scales[r * 3 + k] + u * in.scales[p * 3 + k];
  }
  if (in.opac) out.opac[i] = w * in.opac[r] + u * in.opac[p];
  if (in.rots) {
    const float4 a = reinterpret_cast<const float4*>(in.rots)[r];
    float4 b = reinterpret_cast<const float4*>(in.rots)[p];
    const float dot = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    const float sgn = dot < 0.0f ? -1.0f : 1.0f;
    b.x *= sgn; b.y *= sgn; b.z *= sgn; b.w *= sgn;      // exact
    reinterpret_cast<float4*>(out.rots)[i] = make_float4(w * a.x + u * b.x, w * a.y + u * b.y,
                                                         w * a.z + u * b.z, w * a.w + u * b.w);
  }
}

// w * a + u * b with every operation rounded on its own (see lod_gather_kernel)
__device__ __forceinline__ float lerp_rounded(float a, float b, float w, float u) {
#pragma clang fp contract(off)
  const float x = w * a, y = u * b;
  return x + y;
}
__device__ __forceinline__ float4 lerp_rounded(float4 a, float4 b, float w, float u) {
  return make_float4(lerp_rounded(a.x, b.x, w, u), lerp_rounded(a.y, b.y, w, u), lerp_rounded(a.z, b.z, w, u),
                     lerp_rounded(a.w, b.w, w, u));
}

// SH rows (3M floats, 192 B at M = 16) are moved by their own kernels with one lane per 16-byte (or 4-byte) chunk:
// consecutive lanes walk one row, so every row is read / written as contiguous segments instead of 64 lanes striding
// through 64 different rows.
template <typename V>   // float4 when 3M % 4 == 0, else float
__global__ __launch_bounds__(256) void lod_gather_sh_kernel(const int32_t* __restrict__ render_indices,
                                                            const int32_t* __restrict__ parent_indices,
                                                            const float* __restrict__ weights, int n, int cpr,
                                                            const V* __restrict__ in, V* __restrict__ out) {
#pragma clang fp contract(off)
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)n * cpr) return;
  const int i = (int)(e / cpr), c = (int)(e - (size_t)i * cpr);
  const float w = weights[i], u = 1.0f - w;
  const V a = in[(size_t)render_indices[i] * cpr + c];
  const V b = in[(size_t)parent_indices[i] * cpr + c];
  out[e] = lerp_rounded(a, b, w, u);
}

__global__ __launch_bounds__(256) void lod_monotone_kernel(const int32_t* __restrict__ parent_indices, int n,
                                                           uint32_t* __restrict__ flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) return;
  if (i < n && parent_indices[i] < parent_indices[i - 1]) *flag = 1u;   // benign race: every writer stores 1
}

template <bool ATOMIC>
__device__ __forceinline__ void add_to(float* dst, float v) {
  if (ATOMIC) atomicAdd(dst, v); else *dst += v;
}

// dst rows must be zero-initialised by the caller.
__global__ __launch_bounds__(256) void lod_scatter_kernel(const int32_t* __restrict__ render_indices,
                                                          const int32_t* __restrict__ parent_indices,
                                                          const float* __restrict__ weights, int n, int M,
                                                          LodPtrs g,          // gradients of the interpolated rows [n,...]
                                                          const float* __restrict__ rots_full,
                                                          LodOut d,           // gradients of the full arrays [G,...]
                                                          const uint32_t* __restrict__ nonmono) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool atomic_mode = *nonmono != 0;
  const size_t r = (size_t)render_indices[i];
  const int p = parent_indices[i];
  const float w = weights[i];
  const bool self_parent = (size_t)p == r;      // root: the "parent" is the node itself, the lerp is the identity
  // ---- node row: unique per cut entry -> plain stores -----------------------------------------
  const float wn = self_parent ? 1.0f : w;
  if (g.means) for (int k = 0; k < 3; ++k) d.means[r * 3 + k] += wn * g.means[(size_t)i * 3 + k];
  if (g.scales) for (int k = 0; k < 3; ++k) d.scales[r * 3 + k] += wn * g.scales[(size_t)i * 3 + k];
  if (g.opac) d.opac[r] += wn * g.opac[i];
  if (g.rots) for (int k = 0; k < 4; ++k) d.rots[r * 4 + k] += wn * g.rots[(size_t)i * 4 + k];
  // ---- parent row: the first lane of each run of equal parents sums the run -------------------
  if (i > 0 && parent_indices[i - 1] == p) return;
  int j = i;
  float am[3] = {0.f, 0.f, 0.f}, as[3] = {0.f, 0.f, 0.f}, aq[4] = {0.f, 0.f, 0.f, 0.f}, ao = 0.f;
  for (; j < n && parent_indices[j] == p; ++j) {
    if ((size_t)p == (size_t)render_indices[j]) continue;
    const float u = 1.0f - weights[j];
    if (g.means) for (int k = 0; k < 3; ++k) am[k] += u * g.means[(size_t)j * 3 + k];
    if (g.scales) for (int k = 0; k < 3; ++k) as[k] += u * g.scales[(size_t)j * 3 + k];
    if (g.opac) ao += u * g.opac[j];
    if (g.rots) {
      const float4 a = reinterpret_cast<const float4*>(rots_full)[(size_t)render_indices[j]];
      const float4 b = reinterpret_cast<const float4*>(rots_full)[(size_t)p];
      const float sgn = (a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w) < 0.0f ? -1.0f : 1.0f;
      for (int k = 0; k < 4; ++k) aq[k] += u * sgn * g.rots[(size_t)j * 4 + k];
    }
  }
  const size_t pp = (size_t)p;
  if (atomic_mode) {
    if (g.means) for (int k = 0; k < 3; ++k) add_to<true>(d.means + pp * 3 + k, am[k]);
    if (g.scales) for (int k = 0; k < 3; ++k) add_to<true>(d.scales + pp * 3 + k, as[k]);
    if (g.opac) add_to<true>(d.opac + pp, ao);
    if (g.rots) for (int k = 0; k < 4; ++k) add_to<true>(d.rots + pp * 4 + k, aq[k]);
  } else {
    if (g.means) for (int k = 0; k < 3; ++k) add_to<false>(d.means + pp * 3 + k, am[k]);
    if (g.scales) for (int k = 0; k < 3; ++k) add_to<false>(d.scales + pp * 3 + k, as[k]);
    if (g.opac) add_to<false>(d.opac + pp, ao);
    if (g.rots) for (int k = 0; k < 4; ++k) add_to<false>(d.rots + pp * 4 + k, aq[k]);
  }
}

template <typename V, bool ATOMIC>
__global__ __launch_bounds__(256) void lod_scatter_sh_kernel(const int32_t* __restrict__ render_indices,
                                                             const int32_t* __restrict__ parent_indices,
                                                             const float* __restrict__ weights, int n, int cpr,
                                                             const V* __restrict__ g, V* __restrict__ d,
                                                             const uint32_t* __restrict__ nonmono) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)n * cpr) return;
  if (ATOMIC != (*nonmono != 0)) return;          // both instantiations are launched; the flag picks one
  const int i = (int)(e / cpr), c = (int)(e - (size_t)i * cpr);
  const int r = render_indices[i], p = parent_indices[i];
  const bool self_parent = p == r;
  const V gi = g[e];
  d[(size_t)r * cpr + c] = (self_parent ? 1.0f : weights[i]) * gi;      // node rows are unique: plain store
  if (i > 0 && parent_indices[i - 1] == p) return;                       // not the leader of its run of siblings
  V acc = gi * 0.0f;
  for (int j = i; j < n && parent_indices[j] == p; ++j) {
    if (render_indices[j] == p) continue;
    acc += (1.0f - weights[j]) * g[(size_t)j * cpr + c];
  }
  V* dst = d + (size_t)p * cpr + c;
  if (ATOMIC) {
    float* df = reinterpret_cast<float*>(dst);
    const float* af = reinterpret_cast<const float*>(&acc);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / sizeof(float)); ++k) atomicAdd(df + k, af[k]);
  } else {
    *dst += acc;
  }
}

}  // namespace

int launch_lod_monotone(const int32_t* parent_indices, int32_t n, uint32_t* flag, hipStream_t s) {
  if (n > 1) {
    hipLaunchKernelGGL(lod_monotone_kernel, dim3((n + 255) / 256), dim3(256), 0, s, parent_indices, n, flag);
    HGS_LAUNCH_CHECK("lod_monotone", s, false);
  }
  return HGS_OK;
}

}  // namespace hgs

using namespace hgs;

extern "C" {

int hgs_lod_gather(const int32_t* render_indices, const int32_t* parent_indices, const float* weights, int32_t n,
                   int32_t M, const float* means3D, const float* scales, const float* rotations, const float* shs,
                   const float* opacities, float* o_means3D, float* o_scales, float* o_rotations, float* o_shs,
                   float* o_opacities, hgs_stream_t stream, int device) {
  if (n <= 0) return HGS_OK;
  if (!render_indices || !parent_indices || !weights) { set_error("null index / weight array"); return HGS_ERR_INVALID; }
  if ((means3D && !o_means3D) || (scales && !o_scales) || (rotations && !o_rotations) || (shs && !o_shs) ||
      (opacities && !o_opacities)) { set_error("missing output for a given input"); return HGS_ERR_INVALID; }
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const LodPtrs in = {means3D, scales, rotations, shs, opacities};
  const LodOut out = {o_means3D, o_scales, o_rotations, o_shs, o_opacities};
  hipLaunchKernelGGL(lod_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, render_indices, parent_indices, weights,
                     n, M, in, out);
  HGS_LAUNCH_CHECK("lod_gather", s, false);
  if (shs) {
    const int nf = M * 3;
    if ((nf & 3) == 0) {
      const int cpr = nf / 4;
      hipLaunchKernelGGL(lod_gather_sh_kernel<float4>, dim3((uint32_t)(((size_t)n * cpr + 255) / 256)), dim3(256), 0, s,
                         render_indices, parent_indices, weights, n, cpr, reinterpret_cast<const float4*>(shs),
                         reinterpret_cast<float4*>(o_shs));
    } else {
      hipLaunchKernelGGL(lod_gather_sh_kernel<float>, dim3((uint32_t)(((size_t)n * nf + 255) / 256)), dim3(256), 0, s,
                         render_indices, parent_indices, weights, n, nf, shs, o_shs);
    }
    HGS_LAUNCH_CHECK("lod_gather_sh", s, false);
  }
  return HGS_OK;
}

int hgs_lod_gather_bwd(const int32_t* render_indices, const int32_t* parent_indices, const float* weights, int32_t n,
                       int32_t M, const float* rotations, const float* g_means3D, const float* g_scales,
                       const float* g_rotations, const float* g_shs, const float* g_opacities, float* d_means3D,
                       float* d_scales, float* d_rotations, float* d_shs, float* d_opacities, uint32_t* flag_tmp,
                       hgs_stream_t stream, int device) {
  if (n <= 0) return HGS_OK;
  if (!render_indices || !parent_indices || !weights || !flag_tmp) { set_error("null argument"); return HGS_ERR_INVALID; }
  if ((g_rotations && !rotations) || (g_means3D && !d_means3D) || (g_scales && !d_scales) ||
      (g_rotations && !d_rotations) || (g_shs && !d_shs) || (g_opacities && !d_opacities)) {
    set_error("missing input / output");
    return HGS_ERR_INVALID;
  }
  HGS_HIP(hipSetDevice(device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  HGS_HIP(hipMemsetAsync(flag_tmp, 0, sizeof(uint32_t), s));
  hipLaunchKernelGGL(lod_monotone_kernel, dim3((n + 255) / 256), dim3(256), 0, s, parent_indices, n, flag_tmp);
  HGS_LAUNCH_CHECK("lod_monotone", s, false);
  const LodPtrs g = {g_means3D, g_scales, g_rotations, g_shs, g_opacities};
  const LodOut d = {d_means3D, d_scales, d_rotations, d_shs, d_opacities};
  hipLaunchKernelGGL(lod_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, render_indices, parent_indices, weights,
                     n, M, g, rotations, d, flag_tmp);
  HGS_LAUNCH_CHECK("lod_scatter", s, false);
  if (g_shs) {
    const int nf = M * 3;
    if ((nf & 3) == 0) {
      const int cpr = nf / 4;
      const dim3 grid((uint32_t)(((size_t)n * cpr + 255) / 256));
      const float4* gv = reinterpret_cast<const float4*>(g_shs);
      float4* dv = reinterpret_cast<float4*>(d_shs);
      hipLaunchKernelGGL((lod_scatter_sh_kernel<float4, false>), grid, dim3(256), 0, s, render_indices, parent_indices,
                         weights, n, cpr, gv, dv, flag_tmp);
      hipLaunchKernelGGL((lod_scatter_sh_kernel<float4, true>), grid, dim3(256), 0, s, render_indices, parent_indices,
                         weights, n, cpr, gv, dv, flag_tmp);
    } else {
      const dim3 grid((uint32_t)(((size_t)n * nf + 255) / 256));
      hipLaunchKernelGGL((lod_scatter_sh_kernel<float, false>), grid, dim3(256), 0, s, render_indices, parent_indices,
                         weights, n, nf, g_shs, d_shs, flag_tmp);
      hipLaunchKernelGGL((lod_scatter_sh_kernel<float, true>), grid, dim3(256), 0, s, render_indices, parent_indices,
                         weights, n, nf, g_shs, d_shs, flag_tmp);
    }
    HGS_LAUNCH_CHECK("lod_scatter_sh", s, false);
  }
  return HGS_OK;
}

}  // extern "C"
