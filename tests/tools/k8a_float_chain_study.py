#!/usr/bin/env python
"""Study (CPU, numpy): how much accuracy does K8a's chain rule lose when the part BELOW the conic runs in float32?

K8a (csrc/preprocess.hip, preprocess_bwd_kernel) recomputes the forward projection of a Gaussian in double
(gaussian_math.h: cov3d_from_scale_rot_d, project_gaussian_d) and then chains the per-Gaussian sums of the render
backward to the op's inputs, in double as well.  45 % of the kernel's instructions are double precision and its 160
registers hold it at 3 waves per SIMD (profiles/r04_kernel_resources.md).  The variant -DHGS_K8A_F32=1 keeps the
recomputed forward in double -- the determinant of the 2D covariance needs it -- and runs everything after the conic in
float32 (122 registers, 4 waves per SIMD, double share 28 %).  This script restates both versions in numpy, operation
by operation (numpy's float32 rounds after every operation, the GPU contracts to FMAs: numpy is the pessimistic one), on
the scenes of the scale-parity suite with random per-Gaussian sums, and reports the float32 version's error against the
double version with the two bounds of tests/parity.py: norm-wise max|d| / max|ref| per tensor (tolerance 1e-5) and the
element-wise `|d| <= 1e-5 |ref| + 1e-6 max|ref|` (reported as a multiple of that bound).

    python tests/tools/k8a_float_chain_study.py [P]

Test infrastructure; imports nothing from oracle/ (it compares two restatements of the KERNEL's own arithmetic)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hierarchical-3d-gaussians_amd")):
    sys.path.insert(0, p)
from hgs import synth                    # noqa: E402

F64 = np.float64


def forward_double(p, sc, q, vm, pm, W, H, tfx, tfy, mod=1.0):
    """gaussian_math.h: cov3d_from_scale_rot_d + project_gaussian_d, vectorised; inputs are float32 VALUES held in
    float64 arrays.  vm / pm: the 16 floats of the row-vector matrices as the kernel indexes them (m[j * 4 + i])."""
    r, x, y, z = (q[:, i] for i in range(4))
    s = mod * sc
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                  2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], axis=1)        # [N, 9]
    L = R.reshape(-1, 3, 3) * s[:, None, :]
    Sig = L @ L.transpose(0, 2, 1)                                                                   # [N, 3, 3]
    X, Y, Z = p[:, 0], p[:, 1], p[:, 2]
    tx = vm[0] * X + vm[4] * Y + vm[8] * Z + vm[12]
    ty = vm[1] * X + vm[5] * Y + vm[9] * Z + vm[13]
    tz = vm[2] * X + vm[6] * Y + vm[10] * Z + vm[14]
    hx = pm[0] * X + pm[4] * Y + pm[8] * Z + pm[12]
    hy = pm[1] * X + pm[5] * Y + pm[9] * Z + pm[13]
    hw = pm[3] * X + pm[7] * Y + pm[11] * Z + pm[15]
    pw = 1.0 / (hw + 1e-7)
    fx, fy = W / (2.0 * tfx), H / (2.0 * tfy)
    limx, limy = 1.3 * tfx, 1.3 * tfy
    clampx, clampy = np.abs(tx / tz) > limx, np.abs(ty / tz) > limy
    txc = np.where(clampx, np.where(tx < 0, -limx, limx) * tz, tx)
    tyc = np.where(clampy, np.where(ty < 0, -limy, limy) * tz, ty)
    itz = 1.0 / tz
    J00, J02 = fx * itz, -(fx * txc) * itz * itz
    J11, J12 = fy * itz, -(fy * tyc) * itz * itz
    T0 = np.stack([J00 * vm[j * 4 + 0] + J02 * vm[j * 4 + 2] for j in range(3)], axis=1)
    T1 = np.stack([J11 * vm[j * 4 + 1] + J12 * vm[j * 4 + 2] for j in range(3)], axis=1)
    U0 = np.einsum("ni,nij->nj", T0, Sig)
    U1 = np.einsum("ni,nij->nj", T1, Sig)
    a = (U0 * T0).sum(1) + 0.3
    b = (U0 * T1).sum(1)
    c = (U1 * T1).sum(1) + 0.3
    det = a * c - b * b
    di = 1.0 / det
    return dict(R=R, s=s, pw=pw, hx=hx, hy=hy, itz=itz, fx=np.full_like(tz, fx), fy=np.full_like(tz, fy), txc=txc, tyc=tyc,
                T0=T0, T1=T1, U0=U0, U1=U1, a=a, b=b, c=c, di=di, conA=c * di, conB=-b * di, conC=a * di,
                clampx=clampx, clampy=clampy, tz=tz)


def chain(pd, s, q, vm, pm, W, H, dt):
    """preprocess_bwd_kernel below `project_gaussian_d`, every operand cast to `dt` first, results as float32."""
    c_ = lambda v: np.asarray(v, dtype=F64).astype(dt)
    two, half = dt(2), dt(0.5)
    A, B, C = c_(pd["conA"]), c_(pd["conB"]), c_(pd["conC"])
    S = [c_(s[:, i]) for i in range(10)]
    gA, gB, gC = dt(-0.5) * S[2], -S[3], dt(-0.5) * S[4]
    ggx = -(A * S[0] + B * S[1])
    ggy = -(C * S[1] + B * S[0])
    dm2x, dm2y = ggx * half * dt(W), ggy * half * dt(H)
    pw = c_(pd["pw"])
    dhx, dhy = dm2x * pw, dm2y * pw
    dhw = -(dm2x * c_(pd["hx"]) + dm2y * c_(pd["hy"])) * pw * pw
    vmr, pmr = [dt(v) for v in vm], [dt(v) for v in pm]
    dmean = [pmr[j * 4 + 0] * dhx + pmr[j * 4 + 1] * dhy + pmr[j * 4 + 3] * dhw for j in range(3)]
    a2, b2, c2 = c_(pd["a"]), c_(pd["b"]), c_(pd["c"])
    di = c_(pd["di"])
    di2 = di * di
    ga = (-c2 * c2 * gA + b2 * c2 * gB - b2 * b2 * gC) * di2
    gb = (two * b2 * c2 * gA - (a2 * c2 + b2 * b2) * gB + two * a2 * b2 * gC) * di2
    gc = (-b2 * b2 * gA + a2 * b2 * gB - a2 * a2 * gC) * di2
    hb = half * gb
    T0, T1, U0, U1 = ([c_(pd[k][:, j]) for j in range(3)] for k in ("T0", "T1", "U0", "U1"))
    Gs = [[T0[i] * (ga * T0[j] + hb * T1[j]) + T1[i] * (hb * T0[j] + gc * T1[j]) for j in range(3)] for i in range(3)]
    d_c3 = np.stack([Gs[0][0], two * Gs[0][1], two * Gs[0][2], Gs[1][1], two * Gs[1][2], Gs[2][2]], axis=1)
    dT0 = [two * ga * U0[j] + gb * U1[j] for j in range(3)]
    dT1 = [gb * U0[j] + two * gc * U1[j] for j in range(3)]
    gJ00 = sum(dT0[j] * vmr[j * 4 + 0] for j in range(3))
    gJ02 = sum(dT0[j] * vmr[j * 4 + 2] for j in range(3))
    gJ11 = sum(dT1[j] * vmr[j * 4 + 1] for j in range(3))
    gJ12 = sum(dT1[j] * vmr[j * 4 + 2] for j in range(3))
    itz = c_(pd["itz"])
    itz2 = itz * itz
    itz3 = itz2 * itz
    fx, fy, txc, tyc = c_(pd["fx"]), c_(pd["fy"]), c_(pd["txc"]), c_(pd["tyc"])
    g_txc = -fx * itz2 * gJ02
    g_tyc = -fy * itz2 * gJ12
    g_tz = -fx * itz2 * gJ00 + two * fx * txc * itz3 * gJ02 - fy * itz2 * gJ11 + two * fy * tyc * itz3 * gJ12
    g_tz = g_tz + np.where(pd["clampx"], g_txc * (txc * itz), dt(0))
    g_tz = g_tz + np.where(pd["clampy"], g_tyc * (tyc * itz), dt(0))
    g_tx = np.where(pd["clampx"], dt(0), g_txc)
    g_ty = np.where(pd["clampy"], dt(0), g_tyc)
    g_tz = g_tz + -S[9] * itz2
    dmean = [dmean[j] + (vmr[j * 4 + 0] * g_tx + vmr[j * 4 + 1] * g_ty + vmr[j * 4 + 2] * g_tz) for j in range(3)]
    Rm = [c_(pd["R"][:, i]) for i in range(9)]
    sv = [c_(pd["s"][:, i]) for i in range(3)]
    dM = [[two * sum(Gs[i][j] * (Rm[j * 3 + k] * sv[k]) for j in range(3)) for k in range(3)] for i in range(3)]
    gR = [[dM[i][k] * sv[k] for k in range(3)] for i in range(3)]
    d_scale = np.stack([sum(Rm[i * 3 + k] * dM[i][k] for i in range(3)) for k in range(3)], axis=1)
    r, x, y, z = (c_(q[:, i]) for i in range(4))
    dq0 = two * (-z * gR[0][1] + y * gR[0][2] + z * gR[1][0] - x * gR[1][2] - y * gR[2][0] + x * gR[2][1])
    dq1 = two * (y * gR[0][1] + z * gR[0][2] + y * gR[1][0] - two * x * gR[1][1] - r * gR[1][2] + z * gR[2][0] +
                 r * gR[2][1] - two * x * gR[2][2])
    dq2 = two * (dt(-2) * y * gR[0][0] + x * gR[0][1] + r * gR[0][2] + x * gR[1][0] + z * gR[1][2] - r * gR[2][0] +
                 z * gR[2][1] - two * y * gR[2][2])
    dq3 = two * (dt(-2) * z * gR[0][0] - r * gR[0][1] + x * gR[0][2] + r * gR[1][0] - two * z * gR[1][1] + y * gR[1][2] +
                 x * gR[2][0] + y * gR[2][1])
    f32 = lambda v: np.asarray(v).astype(np.float32)
    return dict(d_means2D=f32(np.stack([dm2x, dm2y], axis=1)), d_means3D=f32(np.stack(dmean, axis=1)), d_cov3D=f32(d_c3),
                d_scales=f32(d_scale), d_rotations=f32(np.stack([dq0, dq1, dq2, dq3], axis=1)))


def errors(ref, got):
    out = {}
    for k in ref:
        r, g = ref[k].astype(F64), got[k].astype(F64)
        m = np.abs(r).max()
        d = np.abs(g - r)
        out[k] = dict(norm=float(d.max() / m), mixed=float((d / (1e-5 * np.abs(r) + 1e-6 * m)).max()),
                      p999_rel=float(np.quantile((d / np.maximum(np.abs(r), 1e-300))[np.abs(r) >= 1e-3 * m], 0.999)))
    return out


def sums(rng, pd, mode, N):
    """Ten per-Gaussian sums of the render backward (render.hip): 0,1 ~ sum X dx, X dy; 2..4 ~ sum X dx^2, X dx dy,
    X dy^2; 5 opacity; 6..8 colour; 9 depth.  'unit': independent N(0, 1); 'footprint': the moments scaled by the
    Gaussian's own extent (dx ~ sigma), which is how they are correlated with the geometry in a real frame."""
    s = rng.standard_normal((N, 10))
    if mode == "footprint":
        sx, sy = np.sqrt(pd["a"]), np.sqrt(pd["c"])
        s[:, 0] *= sx; s[:, 1] *= sy
        s[:, 2] *= sx * sx; s[:, 3] *= sx * sy; s[:, 4] *= sy * sy
    return s.astype(np.float32).astype(F64)           # (the kernel sums float32 records)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    W, H = 1920, 1080
    cam = synth.make_camera(W, H)
    vm = cam.world_view_transform.reshape(-1).double().numpy()
    pm = cam.full_proj_transform.reshape(-1).double().numpy()
    scenes = {
        "metric (s_px 0.5..4)": synth.make_scene(P, cam, seed=0),
        "heavy (s_px 1..8)": synth.make_scene(P, cam, seed=1, s_px=(1.0, 8.0)),
        "trained-like (needles, aniso 0.05)": synth.make_scene_trained_like(P, cam, seed=2),
        "trained-like, aniso 0.005": synth.make_scene_trained_like(P, cam, seed=3, aniso=0.005),
    }
    rng = np.random.default_rng(7)
    report = {}
    for name, sc in scenes.items():
        p = sc.means3D.double().numpy()
        s3 = sc.scales.double().numpy()
        q = sc.rotations.double().numpy()
        pd = forward_double(p, s3, q, vm, pm, W, H, cam.tanfovx, cam.tanfovy)
        keep = pd["tz"] > 0.2
        pd = {k: v[keep] for k, v in pd.items()}
        q = q[keep]
        cond = pd["a"] * pd["c"] * pd["di"]
        report[name] = {"gaussians": int(keep.sum()), "max a*c/det": float(cond.max()), "p99 a*c/det": float(np.quantile(cond, 0.99))}
        for mode in ("unit", "footprint"):
            s = sums(rng, pd, mode, int(keep.sum()))
            ref = chain(pd, s, q, vm, pm, W, H, np.float64)
            got = chain(pd, s, q, vm, pm, W, H, np.float32)
            report[name][mode] = errors(ref, got)
    print(json.dumps(report, indent=1))
    worst_norm = max(e["norm"] for sc in report.values() for m in ("unit", "footprint") for e in sc[m].values())
    worst_mixed = max(e["mixed"] for sc in report.values() for m in ("unit", "footprint") for e in sc[m].values())
    print(f"worst norm-wise error {worst_norm:.2e} (tolerance 1e-5); worst element-wise = {worst_mixed:.2f} x the mixed bound",
          file=sys.stderr)


if __name__ == "__main__":
    main()
