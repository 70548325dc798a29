"""Developer experiment (CPU only): which square root of the whitened Gram matrix keeps mode A's closed loop at the Householder level?
The oracle's compressed triangle R (exact null space) stands in for the stack; A = R L (P_DD = L L^T), G_w = A^T A, then variants of
R_w with R_w^T R_w = G_w, un-whitened H_c = R_w L^-1, the stock EKFUpdate (oracle restatement), posterior fed back, 52 frames."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, scipy.linalg as sla
from open_vins_amd import capi, closed_loop
from oracle import pyoracle

opts = capi.default_options(chi2_multipler=1.0)

def chol_zero_rows(M):  # unpivoted, non-positive pivots leave a zero row (k_gram_chol)
    M = M.copy(); n = M.shape[0]; R = np.zeros_like(M)
    for k in range(n):
        d = M[k, k]
        if d <= 0: continue
        R[k, k:] = M[k, k:] / np.sqrt(d)
        M[k + 1:, k + 1:] -= np.outer(R[k, k + 1:], R[k, k + 1:])
    return R

def chol_pivoted(M, nfree, tol):  # diagonal pivoting among the first nfree columns (the last column = g is only carried)
    M = M.copy(); n = M.shape[0]; R = np.zeros_like(M)
    done = np.zeros(n, bool); done[nfree:] = True
    d0 = np.max(np.diag(M)[:nfree])
    for k in range(nfree):
        dg = np.where(done, -np.inf, np.diag(M))
        p = int(np.argmax(dg))
        d = M[p, p]
        if d <= tol * d0: break
        row = M[p, :] / np.sqrt(d)
        row[done[:n] & (np.arange(n) < nfree)] = 0.0
        R[k, :] = row
        M -= np.outer(row, row)
        done[p] = True
    return R

def variant(name):
    def upd(prob):
        v = capi.Views(prob)
        ref = pyoracle.msckf_update(opts, v, want_compressed=True)
        R, rc, cols = ref["H_comp"], ref["r_comp"], pyoracle.column_map(opts, v)
        D = R.shape[1]
        if name == "householder":
            H, rr = R, rc
        else:
            Pdd = prob.P[np.ix_(cols, cols)]
            L = np.linalg.cholesky(Pdd)
            if TALL:  # a tall stack with the same triangle: H = Q R, Q (m x rows) with orthonormal columns; partial Gram matrices per 256 chunks
                Q, _ = np.linalg.qr(rng.standard_normal((TALL * R.shape[0], R.shape[0])))
                A = np.hstack([(Q @ R) @ L, (Q @ rc)[:, None]])
                Gw = sum(c.T @ c for c in np.array_split(A, 256))
            else:
                A = np.hstack([R @ L, rc[:, None]])
                Gw = A.T @ A
            if name == "chol0":
                Rw = chol_zero_rows(Gw)[:D]
            elif name.startswith("pchol"):
                Rw = chol_pivoted(Gw, D, float(name[5:] or 0))[:D]
            elif name == "eigh":
                w, V = np.linalg.eigh(Gw[:D, :D])
                w = np.maximum(w, 0)
                Rw1 = (V * np.sqrt(w)).T
                # r_c: Rw1^T r_c = g  ->  r_c = diag(1/sqrt w) V^T g on the kept directions
                keep = w > 1e-14 * w.max()
                rcw = np.where(keep, (V.T @ Gw[:D, D]) / np.sqrt(np.where(keep, w, 1)), 0.0)
                Rw = np.hstack([Rw1, rcw[:, None]])
            Hc = sla.solve_triangular(L, Rw[:, :D].T, lower=True, trans='T').T  # Rw L^-1
            H, rr = Hc, Rw[:, D]
        st, P1, dx = pyoracle.ekf_update(prob.P, H, rr, cols, opts.sigma_pix ** 2)
        out = pyoracle.apply_dx(opts, v, dx)
        out.update(P=P1, feat_status=ref["feat_status"])
        errs.setdefault(name, []).append((np.linalg.norm(dx - ref["dx"]) / np.linalg.norm(ref["dx"]), np.linalg.norm(P1 - ref["P"]) / np.linalg.norm(ref["P"])))
        return out
    return upd

if __name__ == "__main__":
    errs = {}
    TALL = int(os.environ.get("TALL", "0"))
    rng = np.random.default_rng(5)
    K = int(os.environ.get("K", "1"))
    C, F = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 50)
    stream = closed_loop.Stream(C=C, feats_per_frame=F, seed=7, K=K, T=int(os.environ.get("T", "0")) or None)
    base = closed_loop.run(stream, lambda prob: pyoracle.msckf_update(opts, capi.Views(prob)))
    for name in ("householder", "chol0", "pchol0", "pchol1e-15", "pchol1e-13", "eigh"):
        res = closed_loop.run(stream, variant(name))
        dev = np.abs(res["est"] - base["est"]).max()
        e = np.array(errs[name])
        print(f"{name:12s} closed-loop deviation {dev:.1e}   one-step dx max {e[:,0].max():.1e} P max {e[:,1].max():.1e}", flush=True)
