"""Developer experiment (CPU only): mode A's pivoted-Gram factor over the prior conditioning sweep of tests/test_gpu_fullsize.py.
Emulation of the device route in numpy: A = H L (P_DD = L L^T, tall stack with the oracle's triangle), Gram matrix, diagonally pivoted
Cholesky, un-whitening by back substitution, then the STOCK EKFUpdate (oracle) — against an extended-precision evaluation, next to
the Householder triangle's own error."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np, scipy.linalg as sla
from open_vins_amd import capi, synth
from oracle import pyoracle as oracle
from test_gpu_fullsize import SWEEP, GATE_OPEN, extended_precision_update, LD, _rel
from dev_mode_a_numerics import chol_pivoted

rng = np.random.default_rng(3)
prob = synth.make_problem(2, F=300)
for kw in SWEEP:
    prob.P = synth.realistic_prior(prob, **kw)
    opts = capi.default_options(chi2_multipler=GATE_OPEN)
    v = capi.Views(prob)
    tri = oracle.triangulate(opts, v)
    ref = oracle.msckf_update(opts, v, want_compressed=True, given=tri)
    cols = oracle.column_map(opts, v)
    R, rc = ref["H_comp"], ref["r_comp"]
    D = R.shape[1]
    PDD = prob.P[np.ix_(cols, cols)]
    ev = np.linalg.eigvalsh(PDD)
    P_true, dx_true = extended_precision_update(prob.P, cols, R, rc, opts.sigma_pix ** 2)
    e_ref = (_rel(ref["P"].astype(LD), P_true), _rel(ref["dx"].astype(LD), dx_true))
    L = np.linalg.cholesky(PDD)
    Q, _ = np.linalg.qr(rng.standard_normal((10 * R.shape[0], R.shape[0])))
    A = np.hstack([(Q @ R) @ L, (Q @ rc)[:, None]])
    Gw = sum(c.T @ c for c in np.array_split(A, 256))
    Rw = chol_pivoted(Gw, D, 1e-15)[:D]
    Hc = sla.solve_triangular(L, Rw[:, :D].T, lower=True, trans='T').T
    st, P1, dx1 = oracle.ekf_update(prob.P, Hc, Rw[:, D], cols, opts.sigma_pix ** 2)
    e_a = (_rel(P1.astype(LD), P_true), _rel(dx1.astype(LD), dx_true))
    print(f"cond(P_DD) {ev[-1] / max(ev[0], 1e-300):.1e}: mode A pivoted |dP|/|P| {e_a[0]:.1e} |ddx|/|dx| {e_a[1]:.1e}   Householder triangle {e_ref[0]:.1e} {e_ref[1]:.1e}   info gain |G_w|/s^2 {np.linalg.norm(Gw[:D,:D], 2):.1e}", flush=True)
