// TEST INFRASTRUCTURE: the per-Gaussian projection of K1 (csrc/preprocess.hip, preprocess_fwd_kernel) driven on the
// host, calling the product's own gaussian_math.h (see common.h beside this file).  Built with
//   g++ -O2 -ffp-contract=off -shared -fPIC
// -- the header's float32 chain is written with contraction off and explicit parentheses, x86-64 SSE arithmetic rounds
// every operation to float32, division and square root are correctly rounded on both sides: the results are the
// device's, bit for bit.
#include "gaussian_math.h"

extern "C" void host_geometry(const float* means, const float* scales, const float* rots, const float* cov3d,
                              const float* vm, const float* pm, int W, int H, float tanfovx, float tanfovy, float mod,
                              int P, uint8_t* visible, int32_t* radii, float* depth, int32_t* rect, uint32_t* touched,
                              float* conic, float* cov3d_out, float* pxpy, uint8_t* clamp, double* cont) {
  const int gx = (W + 15) / 16, gy = (H + 15) / 16;
  for (int i = 0; i < P; ++i) {
    hgs::Proj pr;
    pr.visible = false;
    float R[9], s[3];
    const float q_id[4] = {1.f, 0.f, 0.f, 0.f}, s_id[3] = {1.f, 1.f, 1.f};
    const float* q = rots ? rots + (size_t)i * 4 : q_id;
    const float* sc = scales ? scales + (size_t)i * 3 : s_id;
    if (cov3d) {
      for (int k = 0; k < 6; ++k) pr.c3[k] = cov3d[(size_t)i * 6 + k];
    } else {
      hgs::cov3d_from_scale_rot(sc, mod, q, pr.c3, R, s);
    }
    hgs::project_gaussian(means + (size_t)i * 3, vm, pm, W, H, tanfovx, tanfovy, gx, gy, pr);
    // what K1 stores (preprocess_fwd_kernel: radii, rects, depths, tiles_touched, flags)
    visible[i] = pr.visible ? 1 : 0;
    radii[i] = pr.visible ? (int32_t)pr.rad_f : 0;
    depth[i] = pr.tz;
    rect[i * 4 + 0] = pr.minx; rect[i * 4 + 1] = pr.miny; rect[i * 4 + 2] = pr.maxx; rect[i * 4 + 3] = pr.maxy;
    touched[i] = pr.visible ? (uint32_t)((pr.maxx - pr.minx) * (pr.maxy - pr.miny)) : 0u;
    conic[i * 3 + 0] = pr.conA; conic[i * 3 + 1] = pr.conB; conic[i * 3 + 2] = pr.conC;
    for (int k = 0; k < 6; ++k) cov3d_out[(size_t)i * 6 + k] = pr.c3[k];
    pxpy[i * 2 + 0] = pr.px; pxpy[i * 2 + 1] = pr.py;
    clamp[i] = (pr.clampx ? 1 : 0) | (pr.clampy ? 2 : 0);
    // the continuous twin (double): pixel centre, conic, 2D covariance, 1 / z
    hgs::ProjD pd;
    if (cov3d) {
      for (int k = 0; k < 6; ++k) pd.c3[k] = (double)pr.c3[k];
    } else {
      hgs::cov3d_from_scale_rot_d(sc, mod, q, pd);
    }
    hgs::project_gaussian_d(means + (size_t)i * 3, vm, pm, W, H, tanfovx, tanfovy, pr.clampx, pr.clampy, pd);
    double* c = cont + (size_t)i * 9;
    c[0] = pd.px; c[1] = pd.py; c[2] = pd.conA; c[3] = pd.conB; c[4] = pd.conC; c[5] = pd.a; c[6] = pd.b; c[7] = pd.c;
    c[8] = pd.itz;
  }
}

extern "C" void host_sh_basis(int deg, const float* dirs, int n, float* b, float* dbx, float* dby, float* dbz) {
  for (int i = 0; i < n; ++i) {
    hgs::sh_basis(deg, dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2], b + i * 16);
    hgs::sh_basis_grad(deg, dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2], dbx + i * 16, dby + i * 16, dbz + i * 16);
  }
}

extern "C" float host_lod_opacity(float o, float w, int kids, float* dout_do) { return hgs::lod_opacity(o, w, kids, dout_do); }
