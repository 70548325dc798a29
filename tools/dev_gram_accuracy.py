"""Accuracy of a Gram-matrix (Cholesky-QR) measurement compression against the Householder/Givens one, on the CPU.

Builds the stacked MSCKF system [H | r] of a synthetic snapshot with the oracle's per-feature stages, then runs the EKF
update from (a) the oracle's own compression and (b) R = chol([H r]^T [H r]) with non-positive pivots dropped, and prints the
relative differences of dx and P.  Development evidence for DESIGN.md; the product path never runs this."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_vins_amd import capi, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402


def semidefinite_cholesky(G):
    """Upper-triangular R with R^T R = G for a positive SEMI-definite G: rows whose pivot is not positive are zero."""
    n = G.shape[0]
    R = np.zeros_like(G)
    S = G.copy()
    d0 = np.diag(G).copy()
    dropped = 0
    for k in range(n):
        d = S[k, k]
        if not d > 1e-15 * d0[k]:
            dropped += 1
            continue
        R[k, k:] = S[k, k:] / np.sqrt(d)
        S[k + 1:, k + 1:] -= np.outer(R[k, k + 1:], R[k, k + 1:])
    return R, dropped


def main(cfg=2, F=800, pscale=1.0, f32=False):
    prob = synth.make_problem(cfg, F=F)
    prob.P = prob.P * pscale
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    tri = pyoracle.triangulate(opts, v)
    ref = pyoracle.msckf_update(opts, v, given=tri)
    cols = pyoracle.column_map(opts, v)
    Hs, rs = [], []
    for f in np.nonzero(ref["feat_status"] == capi.FEAT_USED)[0]:
        H_f, H_x, res = pyoracle.feature_jacobian(opts, v, f, tri["p_FinG"][f], tri["p_FinA"][f], int(tri["anchor_meas"][f]))
        _, Hp, rp = pyoracle.nullspace_project(H_f, H_x, res)
        Hs.append(Hp), rs.append(rp)
    H, r = np.vstack(Hs), np.concatenate(rs)
    A = np.hstack([H, r[:, None]])
    if f32:
        A = A.astype(np.float32).astype(np.float64)
    D = H.shape[1]
    # (a) the oracle's compression + update on this very stack
    Hc, rc = pyoracle.measurement_compress(H, r)
    _, Pa, dxa = pyoracle.ekf_update(prob.P, Hc, rc, cols, opts.sigma_pix ** 2)
    # (b) Gram route, summed in blocks of 304 rows like 256 workgroups would
    G = np.zeros((D + 1, D + 1))
    for i in range(0, A.shape[0], 304):
        G += A[i:i + 304].T @ A[i:i + 304]
    R, dropped = semidefinite_cholesky(G)
    _, Pb, dxb = pyoracle.ekf_update(prob.P, R[:D, :D], R[:D, D], cols, opts.sigma_pix ** 2)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    sv = np.linalg.svd(H, compute_uv=False)
    print(f"cfg {cfg} F {F} P x{pscale:g}: stack {H.shape}, sigma_max/sigma_min {sv[0] / sv[-1]:.2e}, dropped pivots {dropped}")
    print(f"  oracle update from its own compression vs full oracle: dx {rel(dxa, ref['dx']):.1e} P {rel(Pa, ref['P']):.1e}")
    print(f"  Gram/Cholesky vs Householder compression:              dx {rel(dxb, dxa):.1e} P {rel(Pb, Pa):.1e} "
          f"max|dP|/max|P| {np.abs(Pb - Pa).max() / np.abs(Pa).max():.1e}")
    w, V = np.linalg.eigh(Pa)
    print(f"  largest posterior eigenvalue {w[-1]:.3e}: relative change of that eigenvalue {abs(V[:, -1] @ (Pb - Pa) @ V[:, -1]) / w[-1]:.1e}")


if __name__ == "__main__":
    main(2, int(sys.argv[1]) if len(sys.argv) > 1 else 800, float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
