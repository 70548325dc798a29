"""Known-answer fixture of ONE small MSCKF update, evaluated independently of oracle/ and of the GPU library.

    python tools/make_known_answer.py            # writes tests/golden/known_answer_msckf_<name>.json.gz

The reference (rpng/open_vins v2.7) ships no golden vectors for this path and cannot be built in this image, so this script
manufactures the pin: it restates the reference's FORMULAS (file:line below) in mpmath at 50 significant digits, with the
reference's float32 operations emulated operation by operation in numpy.float32 (they decide branches of the Levenberg loop and
quantise the predicted pixel), and writes the inputs together with every intermediate the parity tests compare:

    clone-camera pose table          ov_msckf/src/update/UpdaterMSCKF.cpp:97-115
    linear triangulation             ov_core/src/feat/FeatureInitializer.cpp:30-117
    Levenberg / Gauss-Newton         ov_core/src/feat/FeatureInitializer.cpp:202-375, compute_error :377-423
    camera models                    ov_core/src/cam/CamBase.h:130-135, CamRadtan.h:127-200, CamEqui.h
    feature Jacobian                 ov_msckf/src/update/UpdaterHelper.cpp:192-424 (global representations, FEJ)
    left-nullspace projection        ov_msckf/src/update/UpdaterHelper.cpp:426-454
    chi2 gate                        ov_msckf/src/update/UpdaterMSCKF.cpp:209-234
    stacking, compression            ov_msckf/src/update/UpdaterMSCKF.cpp:237-277, UpdaterHelper.cpp:456-487
    EKF update                       ov_msckf/src/state/StateHelper.cpp:116-197
    box-plus                         ov_type/JPLQuat.h:114-125, PoseJPL.h:74-91, Vec.h:55-58

It imports nothing from oracle/ and does not load libovgpu.so; open_vins_amd.synth only supplies the INPUT snapshot (poses,
pixels, prior), which is data, not algorithm.  Quantities that depend on the choice of an orthonormal basis (the Givens sweeps of
the reference against any other elimination order) are recorded through their invariants: chi2, H'^T H', H'^T r', dx, P'.

Every data-dependent decision of the reference (anchor camera, condition / depth / baseline checks, each accept / reject of the
Levenberg loop, the chi2 gate) is taken here in extended precision and its MARGIN is recorded; the script refuses to write a
fixture in which a decision is closer to its threshold than a double-precision evaluation could resolve (1e-9 relative), so the
recorded path is the path the reference's double arithmetic takes.
"""
from __future__ import annotations

import argparse
import gzip
import json
import os
import sys

import numpy as np
from mpmath import mp, mpf

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

mp.dps = 50
F32 = np.float32
MARGIN = 1e-9


class Margins:
    def __init__(self):
        self.worst = {}

    def note(self, what, lhs, rhs):
        """Records how far a comparison `lhs ? rhs` is from flipping, relative to the magnitudes involved."""
        lhs, rhs = mpf(lhs), mpf(rhs)
        scale = max(abs(lhs), abs(rhs), mpf(1e-300))
        m = float(abs(lhs - rhs) / scale)
        self.worst[what] = min(self.worst.get(what, 1.0), m)
        if m < MARGIN:
            raise SystemExit(f"decision '{what}' is razor-edge (margin {m:.3e}): pick another seed")


MG = Margins()


# ------------------------------------------------------------------------------------------------ small linear algebra in mp
def M(rows):
    return mp.matrix(rows)


def eye(n):
    return mp.eye(n)


def skew(w):  # quat_ops.h:135-139
    return M([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def quat_2_rot(q):  # quat_ops.h:152-157 (JPL)
    v = M([[q[0]], [q[1]], [q[2]]])
    return (2 * q[3] * q[3] - 1) * eye(3) - 2 * q[3] * skew(q) + 2 * (v * v.T)


def quat_multiply(q, p):  # quat_ops.h:181-196: [q4 I - skew(q), q; -q^T, q4] p, q4 >= 0, normalised
    qv, pv = M([[q[0]], [q[1]], [q[2]]]), M([[p[0]], [p[1]], [p[2]]])
    top = q[3] * pv - skew(q) * pv + p[3] * qv
    w = q[3] * p[3] - (qv.T * pv)[0]
    out = [top[0], top[1], top[2], w]
    if out[3] < 0:
        out = [-x for x in out]
    n = mp.sqrt(sum(x * x for x in out))
    return [x / n for x in out]


def col(v):
    return M([[x] for x in v])


def vec(m):
    return [m[i] for i in range(m.rows * m.cols)]


def solve_spd(A, B):
    return mp.cholesky_solve(A, B) if hasattr(mp, "cholesky_solve") else mp.lu_solve(A, B)


def f32(x):
    """mp / float -> the float32 nearest to the DOUBLE nearest to x (the reference holds doubles and casts them)."""
    return F32(float(x))


# ------------------------------------------------------------------------------------------------ camera models
def distort_d(cam, fisheye, xn, yn):
    """CamBase::distort_d (CamBase.h:130-135): the normalised point is cast to float, distort_f evaluates in double with float
    products where the reference multiplies two floats, the pixel is cast back to float.  Returns the two float32 pixels."""
    x, y = f32(xn), f32(yn)  # ept1 = uv_norm.cast<float>()
    if not fisheye:  # CamRadtan.h:127-147
        # r = std::sqrt(float * float + float * float): float arithmetic, float sqrt
        r = np.sqrt(F32(F32(x * x) + F32(y * y)))
        r = mpf(float(r))
        r2 = r * r
        r4 = r2 * r2
        xm, ym = mpf(float(x)), mpf(float(y))
        two_xx = mpf(float(F32(F32(F32(2) * x) * x)))  # 2 * uv_norm(0) * uv_norm(0): int * float * float stays float
        two_yy = mpf(float(F32(F32(F32(2) * y) * y)))
        x1 = xm * (1 + cam[4] * r2 + cam[5] * r4) + 2 * cam[6] * xm * ym + cam[7] * (r2 + two_xx)
        y1 = ym * (1 + cam[4] * r2 + cam[5] * r4) + cam[6] * (r2 + two_yy) + 2 * cam[7] * xm * ym
    else:  # CamEqui.h distort_f
        r = np.sqrt(F32(F32(x * x) + F32(y * y)))
        r = mpf(float(r))
        theta = mp.atan(r)
        t2 = theta * theta
        theta_d = theta + cam[4] * theta * t2 + cam[5] * theta * t2 * t2 + cam[6] * theta * t2 * t2 * t2 + cam[7] * theta * t2 * t2 * t2 * t2
        inv_r = 1 / r if r > mpf("1e-8") else mpf(1)
        cdist = theta_d * inv_r if r > mpf("1e-8") else mpf(1)
        x1, y1 = mpf(float(x)) * cdist, mpf(float(y)) * cdist
    u = cam[0] * x1 + cam[2]
    v = cam[1] * y1 + cam[3]
    return f32(u), f32(v)


def distort_jacobian(cam, fisheye, xn, yn):
    """compute_distort_jacobian on the DOUBLE normalised point (CamRadtan.h:155-199 / CamEqui.h): dz_dzn 2x2, dz_dzeta 2x8."""
    x, y = xn, yn
    if not fisheye:
        r = mp.sqrt(x * x + y * y)
        r2 = r * r
        r4 = r2 * r2
        x2, y2, xy = x * x, y * y, x * y
        dzn = M([[cam[0] * ((1 + cam[4] * r2 + cam[5] * r4) + (2 * cam[4] * x2 + 4 * cam[5] * x2 * r2) + 2 * cam[6] * y + (2 * cam[7] * x + 4 * cam[7] * x)),
                  cam[0] * (2 * cam[4] * xy + 4 * cam[5] * xy * r2 + 2 * cam[6] * x + 2 * cam[7] * y)],
                 [cam[1] * (2 * cam[4] * xy + 4 * cam[5] * xy * r2 + 2 * cam[6] * x + 2 * cam[7] * y),
                  cam[1] * ((1 + cam[4] * r2 + cam[5] * r4) + (2 * cam[4] * y2 + 4 * cam[5] * y2 * r2) + 2 * cam[7] * x + (2 * cam[6] * y + 4 * cam[6] * y))]])
        x1 = x * (1 + cam[4] * r2 + cam[5] * r4) + 2 * cam[6] * x * y + cam[7] * (r2 + 2 * x * x)
        y1 = y * (1 + cam[4] * r2 + cam[5] * r4) + cam[6] * (r2 + 2 * y * y) + 2 * cam[7] * x * y
        dze = mp.zeros(2, 8)
        dze[0, 0], dze[0, 2] = x1, 1
        dze[0, 4], dze[0, 5], dze[0, 6], dze[0, 7] = cam[0] * x * r2, cam[0] * x * r4, 2 * cam[0] * x * y, cam[0] * (r2 + 2 * x * x)
        dze[1, 1], dze[1, 3] = y1, 1
        dze[1, 4], dze[1, 5], dze[1, 6], dze[1, 7] = cam[1] * y * r2, cam[1] * y * r4, cam[1] * (r2 + 2 * y * y), 2 * cam[1] * x * y
        return dzn, dze
    # CamEqui.h compute_distort_jacobian
    r = mp.sqrt(x * x + y * y)
    theta = mp.atan(r)
    t2 = theta * theta
    theta_d = theta + cam[4] * theta * t2 + cam[5] * theta * t2 * t2 + cam[6] * theta * t2 * t2 * t2 + cam[7] * theta * t2 * t2 * t2 * t2
    small = not (r > mpf("1e-8"))
    inv_r = mpf(1) if small else 1 / r
    cdist = mpf(1) if small else theta_d * inv_r
    duv_dxy = M([[cam[0], 0], [0, cam[1]]])
    dxy_dxyn = M([[cdist, 0], [0, cdist]])
    dxy_dr = M([[-x * theta_d * inv_r * inv_r], [-y * theta_d * inv_r * inv_r]])
    dr_dxyn = M([[x * inv_r, y * inv_r]])
    dxy_dthd = M([[x * inv_r], [y * inv_r]])
    dthd_dth = 1 + 3 * cam[4] * t2 + 5 * cam[5] * t2 * t2 + 7 * cam[6] * t2 * t2 * t2 + 9 * cam[7] * t2 * t2 * t2 * t2
    dth_dr = 1 / (r * r + 1)
    dzn = duv_dxy * (dxy_dxyn + (dxy_dr + dxy_dthd * dthd_dth * dth_dr) * dr_dxyn)
    x1, y1 = x * cdist, y * cdist
    dze = mp.zeros(2, 8)
    t3 = theta * t2
    dze[0, 0], dze[0, 2] = x1, 1
    dze[0, 4], dze[0, 5], dze[0, 6], dze[0, 7] = [cam[0] * x * inv_r * t3 * t2 ** k for k in range(4)]
    dze[1, 1], dze[1, 3] = y1, 1
    dze[1, 4], dze[1, 5], dze[1, 6], dze[1, 7] = [cam[1] * y * inv_r * t3 * t2 ** k for k in range(4)]
    return dzn, dze


# ------------------------------------------------------------------------------------------------ the update
def groups_of(prob, f):
    """Camera groups of feature f in the order the flattened view lists them (the iteration order of Feature::timestamps), each
    a list of measurement indices in time order."""
    a, b = int(prob.meas_offsets[f]), int(prob.meas_offsets[f + 1])
    order, groups = [], {}
    for i in range(a, b):
        k = int(prob.cam_idx[i])
        if k not in groups:
            groups[k] = []
            order.append(k)
        groups[k].append(i)
    return [(k, groups[k]) for k in order]


def known_answer(prob, opts):
    C_, K_, N = prob.C, prob.K, prob.N
    P = M(prob.P.tolist())
    clone = [[mpf(float(x)) for x in row] for row in prob.clone_q_p]
    clone_fej = [[mpf(float(x)) for x in row] for row in prob.clone_q_p_fej]
    calib = [[mpf(float(x)) for x in row] for row in prob.calib_q_p]
    intr = [[mpf(float(x)) for x in row] for row in prob.intrinsics]
    fisheye = [bool(x) for x in prob.cam_is_fisheye]
    uv = prob.uv.reshape(-1, 2)
    uvn = prob.uvn.reshape(-1, 2)
    sigma2 = mpf(opts["sigma_pix"]) ** 2

    R_GtoI = [quat_2_rot(c) for c in clone]
    p_IinG = [col(c[4:7]) for c in clone]
    R_GtoI_fej = [quat_2_rot(c) for c in clone_fej]
    p_IinG_fej = [col(c[4:7]) for c in clone_fej]
    R_ItoC = [quat_2_rot(c) for c in calib]
    p_IinC = [col(c[4:7]) for c in calib]
    # UpdaterMSCKF.cpp:97-115
    R_GtoC = [[R_ItoC[k] * R_GtoI[j] for j in range(C_)] for k in range(K_)]
    p_CinG = [[p_IinG[j] - R_GtoC[k][j].T * p_IinC[k] for j in range(C_)] for k in range(K_)]

    out = dict(features=[])
    H_big, r_big = [], []
    cols = column_map(prob, opts)
    D = len(cols)

    for f in range(prob.F):
        rec = dict()
        groups = groups_of(prob, f)
        meas = [i for _, g in groups for i in g]
        m = len(meas)
        # ---- anchor (FeatureInitializer.cpp:36-46): strictly more measurements, groups in iteration order
        most, anchor_cam = 0, groups[0][0]
        for k, g in groups:
            if len(g) > most:
                anchor_cam, most = k, len(g)
        anchor_i = dict(groups)[anchor_cam][-1]
        aj = int(prob.clone_idx[anchor_i])
        R_GtoA, p_AinG = R_GtoC[anchor_cam][aj], p_CinG[anchor_cam][aj]
        rec["anchor_meas"] = anchor_i  # index into the flat measurement arrays (include/ovgpu.h)

        def rel(i):
            k, j = int(prob.cam_idx[i]), int(prob.clone_idx[i])
            R_AtoCi = R_GtoC[k][j] * R_GtoA.T
            p_CiinA = R_GtoA * (p_CinG[k][j] - p_AinG)
            return R_AtoCi, p_CiinA

        # ---- linear triangulation (FeatureInitializer.cpp:48-116)
        A, b = mp.zeros(3, 3), mp.zeros(3, 1)
        for i in meas:
            R_AtoCi, p_CiinA = rel(i)
            bi = R_AtoCi.T * col([mpf(float(uvn[i, 0])), mpf(float(uvn[i, 1])), 1])
            bi = bi / mp.norm(bi)
            Bp = skew(vec(bi))
            Ai = Bp.T * Bp
            A += Ai
            b += Ai * p_CiinA
        p_f = mp.lu_solve(A, b)
        sv = mp.svd_r(A, compute_uv=False)
        condA = max(vec(sv)) / min(vec(sv))
        MG.note("triangulation: condition number", condA, opts["max_cond_number"])
        MG.note("triangulation: min depth", p_f[2], opts["min_dist"])
        MG.note("triangulation: max depth", p_f[2], opts["max_dist"])
        ok = not (condA > opts["max_cond_number"] or p_f[2] < opts["min_dist"] or p_f[2] > opts["max_dist"])
        rec["tri_ok"] = bool(ok)
        rec["p_FinA_linear"] = vec(p_f)
        rec["p_FinG_linear"] = vec(R_GtoA.T * p_f + p_AinG)
        if not ok:
            rec["status"] = "TRI_FAILED"
            out["features"].append(rec)
            continue

        # ---- Levenberg loop (FeatureInitializer.cpp:202-325) with the float32 residual path
        def meas_model(i, alpha, beta, rho):
            R, p_CiinA = rel(i)
            p_AinCi = -R * p_CiinA
            h1 = R[0, 0] * alpha + R[0, 1] * beta + R[0, 2] + rho * p_AinCi[0]
            h2 = R[1, 0] * alpha + R[1, 1] * beta + R[1, 2] + rho * p_AinCi[1]
            h3 = R[2, 0] * alpha + R[2, 1] * beta + R[2, 2] + rho * p_AinCi[2]
            return R, p_AinCi, h1, h2, h3

        def float_residual(i, h1, h2, h3):
            z0, z1 = f32(h1 / h3), f32(h2 / h3)          # Eigen::Matrix<float,2,1> z << hi1 / hi3, hi2 / hi3
            r0, r1 = F32(uvn[i, 0] - z0), F32(uvn[i, 1] - z1)  # float subtraction
            nrm = np.sqrt(F32(F32(r0 * r0) + F32(r1 * r1)))     # res.norm() in float
            return r0, r1, mpf(float(nrm)) ** 2                 # std::pow(float, 2) is evaluated in double: exact

        def compute_error(alpha, beta, rho):  # :377-423
            err = mpf(0)
            for i in meas:
                _, _, h1, h2, h3 = meas_model(i, alpha, beta, rho)
                err += float_residual(i, h1, h2, h3)[2]
            return err

        rho = 1 / p_f[2]
        alpha, beta = p_f[0] / p_f[2], p_f[1] / p_f[2]
        lam, eps, runs, recompute = mpf(opts["init_lamda"]), mpf(10000), 0, True
        cost_old = compute_error(alpha, beta, rho)
        Hess, grad = mp.zeros(3, 3), mp.zeros(3, 1)
        trace = []
        while runs < opts["max_runs"] and lam < opts["max_lamda"] and eps > opts["min_dx"]:
            if recompute:
                Hess, grad = mp.zeros(3, 3), mp.zeros(3, 1)
                for i in meas:
                    R, pA, h1, h2, h3 = meas_model(i, alpha, beta, rho)
                    d = h3 * h3
                    H = M([[(R[0, 0] * h3 - h1 * R[2, 0]) / d, (R[0, 1] * h3 - h1 * R[2, 1]) / d, (pA[0] * h3 - h1 * pA[2]) / d],
                           [(R[1, 0] * h3 - h2 * R[2, 0]) / d, (R[1, 1] * h3 - h2 * R[2, 1]) / d, (pA[1] * h3 - h2 * pA[2]) / d]])
                    r0, r1, _ = float_residual(i, h1, h2, h3)
                    grad += H.T * col([mpf(float(r0)), mpf(float(r1))])
                    Hess += H.T * H
            Hl = Hess.copy()
            for d_ in range(3):
                Hl[d_, d_] *= (1 + lam)
            dx = mp.lu_solve(Hl, grad)
            cost = compute_error(alpha + dx[0], beta + dx[1], rho + dx[2])
            MG.note("gauss-newton: cost <= cost_old", cost, cost_old) if cost != cost_old else None
            if cost <= cost_old:
                MG.note("gauss-newton: relative decrease vs min_dcost", (cost_old - cost) / cost_old, opts["min_dcost"])
            if cost <= cost_old and (cost_old - cost) / cost_old < opts["min_dcost"]:
                alpha, beta, rho = alpha + dx[0], beta + dx[1], rho + dx[2]
                eps = 0
                trace.append("converged")
                break
            if cost <= cost_old:
                recompute, cost_old = True, cost
                alpha, beta, rho = alpha + dx[0], beta + dx[1], rho + dx[2]
                runs += 1
                lam = lam / opts["lam_mult"]
                eps = mp.norm(dx)
                MG.note("gauss-newton: |dx| vs min_dx", eps, opts["min_dx"])
                trace.append("accept")
            else:
                recompute = False
                lam = lam * opts["lam_mult"]
                trace.append("inflate")
        rec["gn_trace"] = trace
        p_FinA = col([alpha / rho, beta / rho, 1 / rho])
        # tangent plane of p_FinA (:331-333): the last two columns of the Householder Q span the plane orthogonal to p_FinA
        n_ = p_FinA / mp.norm(p_FinA)
        base_max = mpf(0)
        for i in meas:
            _, p_CiinA = rel(i)
            perp = p_CiinA - n_ * (n_.T * p_CiinA)[0]
            base_max = max(base_max, mp.norm(perp))
        ratio = mp.norm(p_FinA) / base_max
        MG.note("gauss-newton: min depth", p_FinA[2], opts["min_dist"])
        MG.note("gauss-newton: max depth", p_FinA[2], opts["max_dist"])
        MG.note("gauss-newton: baseline ratio", ratio, opts["max_baseline"])
        gn_ok = not (p_FinA[2] < opts["min_dist"] or p_FinA[2] > opts["max_dist"] or ratio > opts["max_baseline"])
        p_FinG = R_GtoA.T * p_FinA + p_AinG
        rec["p_FinA"], rec["p_FinG"], rec["gn_ok"] = vec(p_FinA), vec(p_FinG), bool(gn_ok)
        if not gn_ok:
            rec["status"] = "GN_FAILED"
            out["features"].append(rec)
            continue

        # ---- Jacobian (UpdaterHelper.cpp:192-424), GLOBAL_3D / GLOBAL_FULL_INVERSE_DEPTH: p_FinG_fej = p_FinG (UpdaterMSCKF.cpp:186-194)
        if opts["feat_rep_msckf"] == 0:
            dpfg_dlambda = eye(3)
        else:  # UpdaterHelper.cpp:46-74 (GLOBAL_FULL_INVERSE_DEPTH)
            g = p_FinG
            rho_ = 1 / mp.norm(g)
            phi = mp.acos(rho_ * g[2])
            theta = mp.atan2(g[1], g[0])
            sph, cph, sth, cth = mp.sin(phi), mp.cos(phi), mp.sin(theta), mp.cos(theta)
            dpfg_dlambda = (1 / rho_) * M([[-sth * sph, cth * cph, -(1 / rho_) * cth * sph],
                                           [cth * sph, sth * cph, -(1 / rho_) * sth * sph],
                                           [0, -sph, -(1 / rho_) * cph]])
        H_f = mp.zeros(2 * m, 3)
        H_x = mp.zeros(2 * m, D)
        res = mp.zeros(2 * m, 1)
        pix = []
        for c, i in enumerate(meas):
            k, j = int(prob.cam_idx[i]), int(prob.clone_idx[i])
            p_FinIi = R_GtoI[j] * (p_FinG - p_IinG[j])
            p_FinCi = R_ItoC[k] * p_FinIi + p_IinC[k]
            xn, yn = p_FinCi[0] / p_FinCi[2], p_FinCi[1] / p_FinCi[2]
            ud, vd = distort_d(intr[k], fisheye[k], xn, yn)
            pix.append([float(ud), float(vd)])
            res[2 * c] = mpf(float(uv[i, 0])) - mpf(float(ud))
            res[2 * c + 1] = mpf(float(uv[i, 1])) - mpf(float(vd))
            Rg, pg = R_GtoI[j], p_IinG[j]
            if opts["do_fej"]:  # :336-345: clone first estimates; uv_norm is NOT recomputed
                Rg, pg = R_GtoI_fej[j], p_IinG_fej[j]
                p_FinIi = Rg * (p_FinG - pg)
                p_FinCi = R_ItoC[k] * p_FinIi + p_IinC[k]
            dzn, dze = distort_jacobian(intr[k], fisheye[k], xn, yn)
            z = p_FinCi[2]
            dzn_dpfc = M([[1 / z, 0, -p_FinCi[0] / (z * z)], [0, 1 / z, -p_FinCi[1] / (z * z)]])
            dpfc_dpfg = R_ItoC[k] * Rg
            dpfc_dclone = mp.zeros(3, 6)
            dpfc_dclone[0:3, 0:3] = R_ItoC[k] * skew(vec(p_FinIi))
            dpfc_dclone[0:3, 3:6] = -dpfc_dpfg
            dz_dpfc = dzn * dzn_dpfc
            dz_dpfg = dz_dpfc * dpfc_dpfg
            H_f[2 * c:2 * c + 2, :] = dz_dpfg * dpfg_dlambda
            blk = dz_dpfc * dpfc_dclone
            c0 = cols.index(int(prob.clone_cov_id[j]))
            H_x[2 * c:2 * c + 2, c0:c0 + 6] = blk
            if opts["do_calib_camera_pose"]:
                dcal = mp.zeros(3, 6)
                dcal[0:3, 0:3] = skew(vec(p_FinCi - p_IinC[k]))
                dcal[0:3, 3:6] = eye(3)
                c0 = cols.index(int(prob.calib_cov_id[k]))
                H_x[2 * c:2 * c + 2, c0:c0 + 6] = dz_dpfc * dcal
            if opts["do_calib_camera_intrinsics"]:
                c0 = cols.index(int(prob.intr_cov_id[k]))
                H_x[2 * c:2 * c + 2, c0:c0 + 8] = dze
        rec["pixels_predicted"] = pix
        rec["H_f"], rec["res"] = H_f, res
        # H_x is recorded by its non-zero blocks: per measurement the 2 x 6 clone block, 2 x 6 extrinsics block, 2 x 8 intrinsics block
        blocks = []
        for c, i in enumerate(meas):
            k, j = int(prob.cam_idx[i]), int(prob.clone_idx[i])
            b_ = dict(clone=H_x[2 * c:2 * c + 2, cols.index(int(prob.clone_cov_id[j])):cols.index(int(prob.clone_cov_id[j])) + 6])
            if opts["do_calib_camera_pose"]:
                b_["extrinsics"] = H_x[2 * c:2 * c + 2, cols.index(int(prob.calib_cov_id[k])):cols.index(int(prob.calib_cov_id[k])) + 6]
            if opts["do_calib_camera_intrinsics"]:
                b_["intrinsics"] = H_x[2 * c:2 * c + 2, cols.index(int(prob.intr_cov_id[k])):cols.index(int(prob.intr_cov_id[k])) + 8]
            blocks.append(b_)
        rec["H_x_blocks"] = blocks

        # ---- left nullspace of H_f (UpdaterHelper.cpp:426-454): any orthonormal basis; here from a full QR
        Q, _ = mp.qr(H_f, mode="full")
        Nn = Q[:, 3:]
        Hp, rp = Nn.T * H_x, Nn.T * res
        # ---- chi2 gate (UpdaterMSCKF.cpp:209-234)
        Pm = mp.zeros(D, D)
        for a_, ca in enumerate(cols):
            for b_, cb in enumerate(cols):
                Pm[a_, b_] = P[ca, cb]
        S = Hp * Pm * Hp.T + sigma2 * eye(2 * m - 3)
        chi2 = (rp.T * mp.lu_solve(S, rp))[0]
        thr = opts["chi2_multipler"] * mp.mpf(chi2_quantile_95(2 * m - 3))
        MG.note("chi2 gate", chi2, thr)
        rec["chi2"], rec["chi2_thresh"] = chi2, thr
        rec["rr_projected"] = (rp.T * rp)[0]
        if chi2 > thr:
            rec["status"] = "CHI2_REJECTED"
        else:
            rec["status"] = "USED"
            H_big.append(Hp)
            r_big.append(rp)
        out["features"].append(rec)
        if os.environ.get("KA_VERBOSE"):
            print(f, rec["status"], "cond", mp.nstr(condA, 6), "p_FinA", [mp.nstr(x, 8) for x in rec.get("p_FinA", [])], rec.get("gn_trace"),
                  "chi2", mp.nstr(rec.get("chi2", mpf(0)), 8), mp.nstr(rec.get("chi2_thresh", mpf(0)), 8))

    # ---- stack, (compression = an orthogonal transform: invariants only), EKF update (StateHelper.cpp:116-197)
    rows = sum(h.rows for h in H_big)
    Hs, rs = mp.zeros(rows, D), mp.zeros(rows, 1)
    o = 0
    for h, r in zip(H_big, r_big):
        Hs[o:o + h.rows, :] = h
        rs[o:o + h.rows, :] = r
        o += h.rows
    out["rows_stacked"], out["D"], out["col_cov_id"] = rows, D, cols
    out["rows_compressed"] = min(rows, D)
    Gs = Hs.T * Hs
    out["Gram_stack_upper"] = [[Gs[i, j] for j in range(i, D)] for i in range(D)]
    out["gvec_stack"] = Hs.T * rs
    out["rr_stack"] = (rs.T * rs)[0]
    Hfull = mp.zeros(rows, N)
    for a_, ca in enumerate(cols):
        Hfull[:, ca] = Hs[:, a_]
    if rows == 0:
        raise SystemExit("no feature passed the gate: nothing to pin, pick another snapshot")
    Ma = P * Hfull.T
    S = Hfull * Ma + sigma2 * eye(rows)
    Kg = Ma * mp.inverse(S)  # K = M_a S^-1
    Pn = P - Kg * Ma.T
    Pn = (Pn + Pn.T) / 2
    dx = Kg * rs
    out["dx"] = dx
    out["P_post_upper"] = [[Pn[i, j] for j in range(i, N)] for i in range(N)]
    # ---- box-plus (JPLQuat.h:114-125: dq = [0.5 dth; 1] normalised, q <- dq (x) q; PoseJPL.h:74-91; Vec.h:55-58)
    def pose_plus(qp, d):
        dq = [d[0] / 2, d[1] / 2, d[2] / 2, mpf(1)]
        n = mp.sqrt(sum(x * x for x in dq))
        dq = [x / n for x in dq]
        q = quat_multiply(dq, qp[:4])
        return q + [qp[4] + d[3], qp[5] + d[4], qp[6] + d[5]]
    out["clone_q_p_post"] = [pose_plus(clone[j], [dx[int(prob.clone_cov_id[j]) + t] for t in range(6)]) for j in range(C_)]
    out["calib_q_p_post"] = [pose_plus(calib[k], [dx[int(prob.calib_cov_id[k]) + t] for t in range(6)]) if opts["do_calib_camera_pose"] else calib[k]
                             for k in range(K_)]
    out["intrinsics_post"] = [[intr[k][t] + (dx[int(prob.intr_cov_id[k]) + t] if opts["do_calib_camera_intrinsics"] else 0) for t in range(8)] for k in range(K_)]
    return out


def column_map(prob, opts):
    """Canonical column order of the parity tests (include/ovgpu.h): calibrated camera variables and clones by covariance id."""
    ids = [(int(c), 6) for c in prob.clone_cov_id]
    if opts["do_calib_camera_pose"]:
        ids += [(int(c), 6) for c in prob.calib_cov_id]
    if opts["do_calib_camera_intrinsics"]:
        ids += [(int(c), 8) for c in prob.intr_cov_id]
    cols = []
    for c0, n in sorted(ids):
        cols += list(range(c0, c0 + n))
    return cols


def chi2_quantile_95(dof):
    """boost::math::quantile(chi_squared(dof), 0.95) (UpdaterMSCKF.cpp:52-55): root of the regularised lower incomplete gamma."""
    k = mpf(dof) / 2
    f = lambda x: mp.gammainc(k, 0, x / 2, regularized=True) - mpf("0.95")
    return mp.findroot(f, mpf(dof) + 2 * mp.sqrt(2 * mpf(dof)))


# ------------------------------------------------------------------------------------------------ serialisation
def ser(x):
    if isinstance(x, mp.matrix):
        if x.cols == 1:
            return [mp.nstr(x[i], 25) for i in range(x.rows)]
        return [[mp.nstr(x[i, j], 25) for j in range(x.cols)] for i in range(x.rows)]
    if isinstance(x, mpf):
        return mp.nstr(x, 25)
    if isinstance(x, (list, tuple)):
        return [ser(v) for v in x]
    if isinstance(x, dict):
        return {k: ser(v) for k, v in x.items()}
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


CASES = {
    # name: (synth.make_problem keywords, option overrides)
    "radtan_fej": (dict(cfg=2, C=12, K=2, F=8, track="ragged", min_obs=6, seed=2, outlier_frac=0.15), dict(chi2_multipler=1.0)),
    "equi_nofej_invdepth": (dict(cfg=2, C=12, K=2, F=6, track="ragged", min_obs=6, seed=3, fisheye=True),
                            dict(chi2_multipler=1.0, do_fej=0, feat_rep_msckf=1)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default=None)
    args = ap.parse_args()
    from open_vins_amd import synth
    for name, (kw, over) in CASES.items():
        if args.case and args.case != name:
            continue
        global MG
        MG = Margins()
        prob = synth.make_problem(**kw)
        opts = dict(chi2_multipler=5.0, sigma_pix=1.0, triangulate_1d=0, refine_features=1, max_runs=5, init_lamda=1e-3, max_lamda=1e10,
                    min_dx=1e-6, min_dcost=1e-6, lam_mult=10.0, min_dist=0.10, max_dist=60.0, max_baseline=40.0, max_cond_number=10000.0,
                    do_fej=1, do_calib_camera_pose=1, do_calib_camera_intrinsics=1, feat_rep_msckf=0)
        opts.update(over)
        ans = known_answer(prob, opts)
        doc = dict(
            about="Known-answer fixture: tools/make_known_answer.py (mpmath, 50 digits, float32 operations of the reference emulated); "
                  "values are decimal strings with 25 significant digits.",
            options=opts,
            inputs=dict(N=prob.N, C=prob.C, K=prob.K, P=prob.P.tolist(), clone_q_p=prob.clone_q_p.tolist(), clone_q_p_fej=prob.clone_q_p_fej.tolist(),
                        clone_cov_id=prob.clone_cov_id.tolist(), calib_q_p=prob.calib_q_p.tolist(), intrinsics=prob.intrinsics.tolist(),
                        cam_is_fisheye=prob.cam_is_fisheye.tolist(), calib_cov_id=prob.calib_cov_id.tolist(), intr_cov_id=prob.intr_cov_id.tolist(),
                        meas_offsets=prob.meas_offsets.tolist(), uv=[float(x) for x in prob.uv], uvn=[float(x) for x in prob.uvn],
                        clone_idx=prob.clone_idx.tolist(), cam_idx=prob.cam_idx.tolist()),
            decision_margins={k: v for k, v in MG.worst.items()},
            answer=ser(ans),
        )
        path = os.path.join(ROOT, "tests", "golden", f"known_answer_msckf_{name}.json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as fh:
            fh.write(json.dumps(doc).encode())
        st = [f["status"] for f in ans["features"]]
        print(f"{name}: {prob.F} features {st}, rows {ans['rows_stacked']} -> {ans['rows_compressed']}, D {ans['D']}, "
              f"worst decision margin {min(MG.worst.values()):.2e} -> {path} ({os.path.getsize(path) // 1024} KB)")


if __name__ == "__main__":
    main()
