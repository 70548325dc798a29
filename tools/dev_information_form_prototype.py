"""Numerical prototype (round 5, NEGATIVE result): the Gram matrix of the prior-whitened projected stack against its
information form  L^T (sum Hx^T Hx - W^T W) L  that would need no stack in HBM.  CPU only (numpy + the oracle: a developer tool, not a product path).
The information form loses precision linearly in eps x cond of the prior block: posterior error 2e-9 at cond 1e12, 4e-7 at 3e16,
against 9e-13 for the whitened Gram matrix."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from open_vins_amd import capi, synth
from oracle import pyoracle
ld = np.longdouble
def run(prior, F=300, seed=0):
    prob = synth.make_problem(2, F=F)
    if prior == "realistic":
        prob.P = synth.realistic_prior(prob)
    elif prior == "realistic_tight":
        prob.P = synth.realistic_prior(prob, sigma_g_p=10.0, sigma_g_th=0.2, q_th=1e-5, q_p=1e-6)
    opts = capi.default_options(chi2_multipler=1.0)
    views = capi.Views(prob)
    cols = pyoracle.column_map(opts, views)
    D = len(cols)
    tri = pyoracle.triangulate(opts, views)
    P = prob.P
    # covariance rows of the columns: col_cov ids
    idx = np.array(cols)
    PDD = P[np.ix_(idx, idx)]
    L = np.linalg.cholesky(PDD)
    Lq = L.astype(ld)
    G_old = np.zeros((D, D)); G_ref = np.zeros((D, D), dtype=ld); A = np.zeros((D, D)); A_ref = np.zeros((D,D),dtype=ld)
    g_old = np.zeros(D); b = np.zeros(D); g_ref = np.zeros(D, dtype=ld)
    nuse = 0
    for f in range(F):
        if tri["status"][f] != 0: continue
        Hf, Hx, r = pyoracle.feature_jacobian(opts, views, f, tri["p_FinG"][f])
        nuse += 1
        # old: whiten then project (float64)
        Y = Hx @ L
        Q, _ = np.linalg.qr(Hf, mode="complete")
        Q1, Q2 = Q[:, :3], Q[:, 3:]
        Yp = Q2.T @ Y; rp = Q2.T @ r
        G_old += Yp.T @ Yp; g_old += Yp.T @ rp
        # new: unwhitened information
        W = Q1.T @ Hx; wr = Q1.T @ r
        A += Hx.T @ Hx - W.T @ W; b += Hx.T @ r - W.T @ wr
        # reference in extended precision
        Hxq, Hfq, rq = Hx.astype(ld), Hf.astype(ld), r.astype(ld)
        # Gram-Schmidt QR in longdouble for Q1
        Q1q = np.zeros((Hf.shape[0], 3), dtype=ld)
        for k in range(3):
            v = Hfq[:, k].copy()
            for _ in range(2):
                for j in range(k): v -= (Q1q[:, j] @ v) * Q1q[:, j]
            Q1q[:, k] = v / np.sqrt(v @ v)
        Yq = Hxq @ Lq
        Zq = Q1q.T @ Yq
        G_ref += Yq.T @ Yq - Zq.T @ Zq
        zr = Q1q.T @ rq
        g_ref += Yq.T @ rq - Zq.T @ zr
    G_new = L.T @ A @ L; g_new = L.T @ b
    nrm = np.linalg.norm(G_ref.astype(float))
    e_old = np.linalg.norm((G_old - G_ref).astype(float)) / nrm
    e_new = np.linalg.norm((G_new - G_ref).astype(float)) / nrm
    # downstream: dx, P' of the whitened update
    def upd(G, g):
        Aw = np.eye(D) + G            # sigma = 1
        C = np.linalg.cholesky(Aw)
        # dx_D = L (Aw^-1 g) ; P'_DD = L Aw^-1 L^T
        y = np.linalg.solve(Aw, g)
        return L @ y, L @ np.linalg.solve(Aw, L.T)
    dx_o, P_o = upd(G_old, g_old); dx_n, P_n = upd(G_new, g_new)
    dx_r, P_r = upd(G_ref.astype(float), g_ref.astype(float))
    print(f"{prior:16s} F_used={nuse} cond(P_DD)={np.linalg.cond(PDD):.1e}  |dG|/|G| old {e_old:.1e} new {e_new:.1e}   dx err old {np.linalg.norm(dx_o-dx_r)/np.linalg.norm(dx_r):.1e} new {np.linalg.norm(dx_n-dx_r)/np.linalg.norm(dx_r):.1e}   P err old {np.linalg.norm(P_o-P_r)/np.linalg.norm(P_r):.1e} new {np.linalg.norm(P_n-P_r)/np.linalg.norm(P_r):.1e}")
for pr in ["default", "realistic", "realistic_tight"]:
    run(pr)
