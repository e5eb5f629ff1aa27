// Per-Gaussian projection math shared by the forward and backward preprocess kernels.
//
// Everything that feeds a DISCRETE decision (cull, radius, tile rectangle, depth key)
// follows the float32 operation order of oracle/raster_oracle.py::geometry_spec with
// floating-point contraction switched off, so radii / tile rectangles / (tile|depth)
// keys are bit-identical to the oracle's.  Restates the public 3DGS projection
// (SURVEY.md App. A 1-5); conventions pinned by the reference's
// utils/general_utils.py:82-114 (quaternion -> R, Sigma = L L^T) and
// scene/cameras.py:95-97 (row-vector matrices).
#pragma once
#include "common.h"

namespace hgs {

struct Proj {
  float tx, ty, tz;        // view-space position
  float hx, hy, hw, pw;    // clip-space, 1/(w+eps)
  float txc, tyc;          // clamped tx, ty (EWA)
  bool clampx, clampy;
  float fx, fy;
  float J00, J02, J11, J12;
  float T0[3], T1[3];      // T = J * W
  float U0[3], U1[3];      // U = T * Sigma
  float c3[6];             // Sigma (xx,xy,xz,yy,yz,zz)
  float a, b, c, det;      // 2D covariance (+0.3 dilation) and determinant
  float conA, conB, conC;  // conic
  float px, py;            // pixel-centre position
  float rad_f;             // screen radius (float, integral)
  int minx, miny, maxx, maxy;
  bool visible;
};

__device__ __forceinline__ void quat_to_rot(const float q[4], float R[9]) {
#pragma clang fp contract(off)
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.0f - 2.0f * (y * y + z * z);
  R[1] = 2.0f * (x * y - r * z);
  R[2] = 2.0f * (x * z + r * y);
  R[3] = 2.0f * (x * y + r * z);
  R[4] = 1.0f - 2.0f * (x * x + z * z);
  R[5] = 2.0f * (y * z - r * x);
  R[6] = 2.0f * (x * z - r * y);
  R[7] = 2.0f * (y * z + r * x);
  R[8] = 1.0f - 2.0f * (x * x + y * y);
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float sc[3], float mod, const float q[4],
                                                     float c3[6], float R[9], float s[3]) {
#pragma clang fp contract(off)
  s[0] = mod * sc[0];
  s[1] = mod * sc[1];
  s[2] = mod * sc[2];
  quat_to_rot(q, R);
  float L[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) L[i * 3 + k] = R[i * 3 + k] * s[k];
  int n = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) {
      c3[n++] = (L[i * 3 + 0] * L[j * 3 + 0] + L[i * 3 + 1] * L[j * 3 + 1]) + L[i * 3 + 2] * L[j * 3 + 2];
    }
}

__device__ __forceinline__ float xform_row(const float* m, float x, float y, float z, int row) {
#pragma clang fp contract(off)
  return ((m[row] * x + m[4 + row] * y) + m[8 + row] * z) + m[12 + row];
}

__device__ __forceinline__ int tile_clamp(float v, int hi) {
#pragma clang fp contract(off)
  float t = truncf(v * 0.0625f);
  if (t != t) t = 0.0f;
  t = fminf((float)hi, fmaxf(0.0f, t));
  return (int)t;
}

// vm / pm: stored (column-major standard) matrices in registers or LDS.
__device__ __forceinline__ void project_gaussian(const float p[3], const float* vm, const float* pm,
                                                 int W, int H, float tanfovx, float tanfovy,
                                                 int gx, int gy, Proj& o) {
#pragma clang fp contract(off)
  const float x = p[0], y = p[1], z = p[2];
  o.tx = xform_row(vm, x, y, z, 0);
  o.ty = xform_row(vm, x, y, z, 1);
  o.tz = xform_row(vm, x, y, z, 2);
  o.visible = o.tz > 0.2f;
  o.hx = xform_row(pm, x, y, z, 0);
  o.hy = xform_row(pm, x, y, z, 1);
  o.hw = xform_row(pm, x, y, z, 3);
  o.pw = 1.0f / (o.hw + 0.0000001f);
  const float ndcx = o.hx * o.pw;
  const float ndcy = o.hy * o.pw;

  o.fx = (float)W / (2.0f * tanfovx);
  o.fy = (float)H / (2.0f * tanfovy);
  const float limx = 1.3f * tanfovx;
  const float limy = 1.3f * tanfovy;
  const float txtz = o.tx / o.tz;
  const float tytz = o.ty / o.tz;
  o.clampx = (txtz < -limx) || (txtz > limx);
  o.clampy = (tytz < -limy) || (tytz > limy);
  o.txc = fminf(limx, fmaxf(-limx, txtz)) * o.tz;
  o.tyc = fminf(limy, fmaxf(-limy, tytz)) * o.tz;
  const float tz2 = o.tz * o.tz;
  o.J00 = o.fx / o.tz;
  o.J02 = -(o.fx * o.txc) / tz2;
  o.J11 = o.fy / o.tz;
  o.J12 = -(o.fy * o.tyc) / tz2;
  // standard-orientation view rotation: Wm[i][j] = vm[j*4+i]
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    o.T0[j] = o.J00 * vm[j * 4 + 0] + o.J02 * vm[j * 4 + 2];
    o.T1[j] = o.J11 * vm[j * 4 + 1] + o.J12 * vm[j * 4 + 2];
  }
  const float* c3 = o.c3;
  const float S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    o.U0[j] = (o.T0[0] * S[0][j] + o.T0[1] * S[1][j]) + o.T0[2] * S[2][j];
    o.U1[j] = (o.T1[0] * S[0][j] + o.T1[1] * S[1][j]) + o.T1[2] * S[2][j];
  }
  o.a = ((o.U0[0] * o.T0[0] + o.U0[1] * o.T0[1]) + o.U0[2] * o.T0[2]) + 0.3f;
  o.b = (o.U0[0] * o.T1[0] + o.U0[1] * o.T1[1]) + o.U0[2] * o.T1[2];
  o.c = ((o.U1[0] * o.T1[0] + o.U1[1] * o.T1[1]) + o.U1[2] * o.T1[2]) + 0.3f;
  o.det = o.a * o.c - o.b * o.b;
  if (o.det == 0.0f) o.visible = false;
  const float det_inv = 1.0f / o.det;
  o.conA = o.c * det_inv;
  o.conB = (-o.b) * det_inv;
  o.conC = o.a * det_inv;

  const float mid = 0.5f * (o.a + o.c);
  const float disc = fmaxf(0.1f, mid * mid - o.det);
  const float sq = sqrtf(disc);
  const float lam = fmaxf(mid + sq, mid - sq);
  o.rad_f = ceilf(3.0f * sqrtf(lam));
  o.px = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
  o.py = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
  o.minx = tile_clamp(o.px - o.rad_f, gx);
  o.maxx = tile_clamp(o.px + o.rad_f + 15.0f, gx);
  o.miny = tile_clamp(o.py - o.rad_f, gy);
  o.maxy = tile_clamp(o.py + o.rad_f + 15.0f, gy);
  const bool finite = (fabsf(o.rad_f) <= 3.0e38f) && (fabsf(o.px) <= 3.0e38f) && (fabsf(o.py) <= 3.0e38f);
  if (!finite || (o.maxx - o.minx) * (o.maxy - o.miny) <= 0) o.visible = false;
}


// ---------------------------------------------------------------------------------------------
// Double-precision twin of the projection for everything CONTINUOUS (pixel centre, conic, and the
// whole backward chain).  The float32 chain above stays the authority for discrete decisions
// (cull, radius, tile rectangle, depth key, clamp flags); this one removes float32's ~1e-4 px
// error of the pixel centre at 1080p+ (ulp of 1900.x is 1.2e-4), which otherwise dominates the
// gradient error of sub-pixel Gaussians.  K1/K8 are HBM-bound; the extra flops are free.
// ---------------------------------------------------------------------------------------------
// 1 / d for the double-precision chain: the float reciprocal (v_rcp_f32, 1 ulp) refined by ONE Newton step in double --
// relative error ~1e-14, five instructions where an IEEE double division expands to a dozen (v_div_scale, v_rcp_f64,
// three refinement FMAs, v_div_fmas, v_div_fixup) at half rate.  The chain's divisors (w + 1e-7, z > 0.2, the dilated
// determinant >= 0.09) are far inside float range; a result of 1e-14 relative accuracy moves the pixel centre by 3e-11 px.
__device__ __forceinline__ double rcp_d(double d) {
  const double r = (double)__builtin_amdgcn_rcpf((float)d);
  return fma(fma(-d, r, 1.0), r, r);
}

struct ProjD {
  double tx, ty, tz, hx, hy, hw, pw, itz, di;      // itz = 1 / tz, di = 1 / det
  double txc, tyc, fx, fy;
  double T0[3], T1[3], U0[3], U1[3];
  double c3[6];
  double R[9], s[3];
  double a, b, c, det, conA, conB, conC;
  double px, py;
};

__device__ __forceinline__ void cov3d_from_scale_rot_d(const float sc[3], float mod, const float q[4], ProjD& o) {
  const double r = q[0], x = q[1], y = q[2], z = q[3];
  o.s[0] = (double)mod * sc[0];
  o.s[1] = (double)mod * sc[1];
  o.s[2] = (double)mod * sc[2];
  double* R = o.R;
  R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - r * z);       R[2] = 2.0 * (x * z + r * y);
  R[3] = 2.0 * (x * y + r * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - r * x);
  R[6] = 2.0 * (x * z - r * y);       R[7] = 2.0 * (y * z + r * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
  double L[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) L[i * 3 + k] = R[i * 3 + k] * o.s[k];
  int n = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j)
      o.c3[n++] = L[i * 3 + 0] * L[j * 3 + 0] + L[i * 3 + 1] * L[j * 3 + 1] + L[i * 3 + 2] * L[j * 3 + 2];
}

// clampx / clampy: the float32 chain's EWA clamp decisions (flags bits 3,4) so both chains agree.
__device__ __forceinline__ void project_gaussian_d(const float p[3], const float* vm, const float* pm, int W, int H,
                                                   float tanfovx, float tanfovy, bool clampx, bool clampy,
                                                   ProjD& o) {
  const double x = p[0], y = p[1], z = p[2];
  o.tx = (double)vm[0] * x + (double)vm[4] * y + (double)vm[8] * z + (double)vm[12];
  o.ty = (double)vm[1] * x + (double)vm[5] * y + (double)vm[9] * z + (double)vm[13];
  o.tz = (double)vm[2] * x + (double)vm[6] * y + (double)vm[10] * z + (double)vm[14];
  o.hx = (double)pm[0] * x + (double)pm[4] * y + (double)pm[8] * z + (double)pm[12];
  o.hy = (double)pm[1] * x + (double)pm[5] * y + (double)pm[9] * z + (double)pm[13];
  o.hw = (double)pm[3] * x + (double)pm[7] * y + (double)pm[11] * z + (double)pm[15];
  o.pw = rcp_d(o.hw + 1e-7);
  o.px = ((o.hx * o.pw + 1.0) * (double)W - 1.0) * 0.5;
  o.py = ((o.hy * o.pw + 1.0) * (double)H - 1.0) * 0.5;
  o.fx = (double)W / (2.0 * (double)tanfovx);
  o.fy = (double)H / (2.0 * (double)tanfovy);
  const double limx = 1.3 * (double)tanfovx, limy = 1.3 * (double)tanfovy;
  o.txc = clampx ? (o.tx < 0.0 ? -limx : limx) * o.tz : o.tx;
  o.tyc = clampy ? (o.ty < 0.0 ? -limy : limy) * o.tz : o.ty;
  const double itz = o.itz = rcp_d(o.tz);
  const double J00 = o.fx * itz, J02 = -(o.fx * o.txc) * itz * itz;
  const double J11 = o.fy * itz, J12 = -(o.fy * o.tyc) * itz * itz;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    o.T0[j] = J00 * (double)vm[j * 4 + 0] + J02 * (double)vm[j * 4 + 2];
    o.T1[j] = J11 * (double)vm[j * 4 + 1] + J12 * (double)vm[j * 4 + 2];
  }
  const double* c3 = o.c3;
  const double S[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    o.U0[j] = o.T0[0] * S[0][j] + o.T0[1] * S[1][j] + o.T0[2] * S[2][j];
    o.U1[j] = o.T1[0] * S[0][j] + o.T1[1] * S[1][j] + o.T1[2] * S[2][j];
  }
  o.a = o.U0[0] * o.T0[0] + o.U0[1] * o.T0[1] + o.U0[2] * o.T0[2] + 0.3;
  o.b = o.U0[0] * o.T1[0] + o.U0[1] * o.T1[1] + o.U0[2] * o.T1[2];
  o.c = o.U1[0] * o.T1[0] + o.U1[1] * o.T1[1] + o.U1[2] * o.T1[2] + 0.3;
  o.det = o.a * o.c - o.b * o.b;
  const double di = o.di = rcp_d(o.det);
  o.conA = o.c * di;
  o.conB = -o.b * di;
  o.conC = o.a * di;
}

// ---------------------------------------------------------------------------------------------
// Raw-parameter path (hgs_raster_args.activations): the activations of
// scene/gaussian_model.py:108-128 evaluated in double and rounded ONCE to float32 -- that float32
// value is what every later stage (discrete float32 chain and continuous double chain) sees, exactly
// as if the caller had passed it.  nrm receives max(|q_raw|, 1e-12) (1 without normalisation).
// ---------------------------------------------------------------------------------------------
// ---- in-kernel LOD interpolation (hgs_raster_args.lod_*) -------------------------------------------------------------
struct LodRow {
  size_t r, p;    // node row, parent row of the attribute arrays (r == p: the row is taken as it is)
  float w, u;     // weight of the node row, 1 - w
};
// LOD is a template parameter of the kernels that use these helpers: the plain call must not pay registers for it
template <bool LOD>
__device__ __forceinline__ LodRow lod_row(const hgs_raster_args& a, int idx) {
  LodRow l;
  if constexpr (!LOD) { l.r = l.p = (size_t)idx; l.w = 1.0f; l.u = 0.0f; return l; }
  if (idx < a.lod_n) {
    l.r = (size_t)a.lod_render_indices[idx];
    l.p = (size_t)a.lod_parent_indices[idx];
    l.w = a.interpolation_weights[idx];
    l.u = 1.0f - l.w;
  } else {                                     // skybox tail of the arrays
    l.r = l.p = (size_t)(a.lod_rows - (a.P - a.lod_n) + (idx - a.lod_n));
    l.w = 1.0f; l.u = 0.0f;
  }
  return l;
}
// The row as the GATHERS see it: with weight exactly 1 the parent row is not read -- w x + 0 y = x for every finite y --
// the node row stands in for it (same address twice: no second HBM row; the load itself stays unconditional -- a branch
// around it serialises the row's loads and cost K1 +30 %).  82 % of the rows of the 50 M-node render loop
// and 25-90 % of a train_post-shaped cut have weight 1 (the parent is more than twice too coarse), and a row costs
// 236 bytes.  Differences to the Python expression t * x[r] + (1 - t) * x[p]: a -0.0 attribute stays -0.0 (the
// expression gives +0.0 next to a positive parent value) and a non-finite parent attribute does not reach rows it has
// no weight in.  The gradient scatter keeps using lod_row: the parent of a weight-1 row still belongs to its siblings'
// run.
template <bool LOD>
__device__ __forceinline__ LodRow lod_row_gather(const hgs_raster_args& a, int idx) {
  LodRow l = lod_row<LOD>(a, idx);
  if constexpr (LOD) {
    if (l.w == 1.0f) l.p = l.r;
  }
  return l;
}
// w * x + u * y with both products and the sum rounded separately (what torch's t * x[r] + (1 - t) * x[p] does)
__device__ __forceinline__ float lod_lerp(float x, float y, float w, float u) {
#pragma clang fp contract(off)
  const float a = w * x, b = u * y;
  return a + b;
}
template <bool LOD>
__device__ __forceinline__ void load_mean(const hgs_raster_args& a, const LodRow& l, float p[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float x = a.means3D[l.r * 3 + k];
    if constexpr (LOD) p[k] = lod_lerp(x, a.means3D[l.p * 3 + k], l.w, l.u);
    else p[k] = x;
  }
}

template <bool LOD>
__device__ __forceinline__ void load_scale_rot(const hgs_raster_args& a, int idx, float sc[3], float q[4],
                                               double* nrm) {
  if constexpr (LOD) {                         // interpolated row (no activations in this mode)
    const LodRow l = lod_row_gather<true>(a, idx);
#pragma unroll
    for (int k = 0; k < 3; ++k) sc[k] = lod_lerp(a.scales[l.r * 3 + k], a.scales[l.p * 3 + k], l.w, l.u);
    const float4 qa = reinterpret_cast<const float4*>(a.rotations)[l.r];
    float4 qb = reinterpret_cast<const float4*>(a.rotations)[l.p];
    const float sgn = (qa.x * qb.x + qa.y * qb.y + qa.z * qb.z + qa.w * qb.w) < 0.0f ? -1.0f : 1.0f;
    q[0] = lod_lerp(qa.x, sgn * qb.x, l.w, l.u); q[1] = lod_lerp(qa.y, sgn * qb.y, l.w, l.u);
    q[2] = lod_lerp(qa.z, sgn * qb.z, l.w, l.u); q[3] = lod_lerp(qa.w, sgn * qb.w, l.w, l.u);
    if (nrm) *nrm = 1.0;
    return;
  }
  sc[0] = a.scales[idx * 3 + 0]; sc[1] = a.scales[idx * 3 + 1]; sc[2] = a.scales[idx * 3 + 2];
  if (a.activations & HGS_ACT_SCALE_EXP) {
#pragma unroll
    for (int i = 0; i < 3; ++i) sc[i] = (float)exp((double)sc[i]);
  }
  const float4 qv = reinterpret_cast<const float4*>(a.rotations)[idx];
  q[0] = qv.x; q[1] = qv.y; q[2] = qv.z; q[3] = qv.w;
  double n = 1.0;
  if (a.activations & HGS_ACT_ROT_NORMALIZE) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    n = fmax(sqrt(((w * w + x * x) + y * y) + z * z), 1e-12);
    q[0] = (float)(w / n); q[1] = (float)(x / n); q[2] = (float)(y / n); q[3] = (float)(z / n);
  }
  if (nrm) *nrm = n;
}

// activated opacity; dact (optional) = d(activated)/d(raw)
template <bool LOD>
__device__ __forceinline__ float load_opacity(const hgs_raster_args& a, int idx, double* dact) {
  if constexpr (LOD) {
    const LodRow l = lod_row_gather<true>(a, idx);
    if (dact) *dact = 1.0;
    return lod_lerp(a.opacities[l.r], a.opacities[l.p], l.w, l.u);
  }
  const float raw = a.opacities[idx];
  if (a.activations & HGS_ACT_OPACITY_SIGMOID) {
    const double o = 1.0 / (1.0 + exp(-(double)raw));
    if (dact) *dact = o * (1.0 - o);
    return (float)o;
  }
  if (a.activations & HGS_ACT_OPACITY_ABS) {
    if (dact) *dact = raw > 0.f ? 1.0 : (raw < 0.f ? -1.0 : 0.0);
    return fabsf(raw);
  }
  if (dact) *dact = 1.0;
  return raw;
}

// Hierarchy-mode opacity remap (DESIGN.md 'LOD opacity'; oracle: raster_oracle.lod_opacity): a per-GAUSSIAN remap of
// the opacity.  k stacked copies of 1 - (1 - o)^(1/k) match one copy of o only where the falloff is 1 (the centre);
// off-centre the children are more opaque than the parent (measured: profiles/r04_lod_remap_kat.txt).  Whether upstream
// remaps o or the per-pixel alpha is what the pin kit's upstream_raster_post.npz decides; this stays one function.
__device__ __forceinline__ float lod_opacity(float o, float w, int kids, float* dout_do) {
  if (kids < 2) {
    if (dout_do) *dout_do = 1.0f;
    return o;
  }
  const float invk = 1.0f / (float)kids;
  const float oc = fminf(o, 0.99f);
  const float base = 1.0f - oc;
  const float pw = powf(base, invk);
  if (dout_do) {
    const float dstack = (o < 0.99f) ? invk * pw / base : 0.0f;
    *dout_do = w + (1.0f - w) * dstack;
  }
  return w * o + (1.0f - w) * (1.0f - pw);
}

// unit view direction (p - campos) / |p - campos| and 1 / |p - campos|: one piece of code (contraction off) for every
// kernel that evaluates the SH basis, so that forward, backward and all instantiations see the same direction bits
__device__ __forceinline__ float unit_dir(const float p[3], const float* cam, float& dx, float& dy, float& dz) {
#pragma clang fp contract(off)
  dx = p[0] - cam[0]; dy = p[1] - cam[1]; dz = p[2] - cam[2];
  const float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
  dx *= inv; dy *= inv; dz *= inv;
  return inv;
}

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f;
constexpr float SH_C2_1 = -1.0925484305920792f;
constexpr float SH_C2_2 = 0.31539156525252005f;
constexpr float SH_C2_3 = -1.0925484305920792f;
constexpr float SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f;
constexpr float SH_C3_1 = 2.890611442640554f;
constexpr float SH_C3_2 = -0.4570457994644658f;
constexpr float SH_C3_3 = 0.3731763325901154f;
constexpr float SH_C3_4 = -0.4570457994644658f;
constexpr float SH_C3_5 = 1.445305721320277f;
constexpr float SH_C3_6 = -0.5900435899266435f;

// SH basis values for a unit direction (utils/sh_utils.py:57-112 polynomial).
// (contraction off, like the geometry chain: the values must not depend on what the surrounding kernel lets the
// compiler fuse -- K1's instantiations have to give the same colour bits, tests/test_lod_gpu.py)
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float b[16]) {
#pragma clang fp contract(off)
  b[0] = SH_C0;
  if (deg > 0) {
    b[1] = -SH_C1 * y;
    b[2] = SH_C1 * z;
    b[3] = -SH_C1 * x;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = SH_C2_0 * xy;
      b[5] = SH_C2_1 * yz;
      b[6] = SH_C2_2 * (2.0f * zz - xx - yy);
      b[7] = SH_C2_3 * xz;
      b[8] = SH_C2_4 * (xx - yy);
      if (deg > 2) {
        b[9] = SH_C3_0 * y * (3.0f * xx - yy);
        b[10] = SH_C3_1 * xy * z;
        b[11] = SH_C3_2 * y * (4.0f * zz - xx - yy);
        b[12] = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
        b[13] = SH_C3_4 * x * (4.0f * zz - xx - yy);
        b[14] = SH_C3_5 * z * (xx - yy);
        b[15] = SH_C3_6 * x * (xx - 3.0f * yy);
      }
    }
  }
}

// d(basis_k)/d(x,y,z)
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float dbx[16],
                                              float dby[16], float dbz[16]) {
#pragma clang fp contract(off)
  dbx[0] = dby[0] = dbz[0] = 0.0f;
  if (deg > 0) {
    dbx[1] = 0.0f;      dby[1] = -SH_C1;    dbz[1] = 0.0f;
    dbx[2] = 0.0f;      dby[2] = 0.0f;      dbz[2] = SH_C1;
    dbx[3] = -SH_C1;    dby[3] = 0.0f;      dbz[3] = 0.0f;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      dbx[4] = SH_C2_0 * y;          dby[4] = SH_C2_0 * x;          dbz[4] = 0.0f;
      dbx[5] = 0.0f;                 dby[5] = SH_C2_1 * z;          dbz[5] = SH_C2_1 * y;
      dbx[6] = SH_C2_2 * -2.0f * x;  dby[6] = SH_C2_2 * -2.0f * y;  dbz[6] = SH_C2_2 * 4.0f * z;
      dbx[7] = SH_C2_3 * z;          dby[7] = 0.0f;                 dbz[7] = SH_C2_3 * x;
      dbx[8] = SH_C2_4 * 2.0f * x;   dby[8] = SH_C2_4 * -2.0f * y;  dbz[8] = 0.0f;
      if (deg > 2) {
        dbx[9] = SH_C3_0 * 6.0f * xy;               dby[9] = SH_C3_0 * (3.0f * xx - 3.0f * yy);      dbz[9] = 0.0f;
        dbx[10] = SH_C3_1 * yz;                     dby[10] = SH_C3_1 * xz;                          dbz[10] = SH_C3_1 * xy;
        dbx[11] = SH_C3_2 * -2.0f * xy;             dby[11] = SH_C3_2 * (4.0f * zz - xx - 3.0f * yy); dbz[11] = SH_C3_2 * 8.0f * yz;
        dbx[12] = SH_C3_3 * -6.0f * xz;             dby[12] = SH_C3_3 * -6.0f * yz;                  dbz[12] = SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
        dbx[13] = SH_C3_4 * (4.0f * zz - 3.0f * xx - yy); dby[13] = SH_C3_4 * -2.0f * xy;            dbz[13] = SH_C3_4 * 8.0f * xz;
        dbx[14] = SH_C3_5 * 2.0f * xz;              dby[14] = SH_C3_5 * -2.0f * yz;                  dbz[14] = SH_C3_5 * (xx - yy);
        dbx[15] = SH_C3_6 * (3.0f * xx - 3.0f * yy); dby[15] = SH_C3_6 * -6.0f * xy;                 dbz[15] = 0.0f;
      }
    }
  }
}

}  // namespace hgs
