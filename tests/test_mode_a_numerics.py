"""CPU: why mode A's compressed system comes from a PIVOTED Cholesky factorisation.

The oracle's closed loop (BASELINE configs[0] shape, 52 frames, posterior fed back) with mode A emulated in numpy: the oracle's
compressed triangle stands in for the stack, A = R L (P_DD = L L^T), G_w = A^T A, a square root R_w of G_w, H_c = R_w L^-1, then the
STOCK EKFUpdate (the oracle's restatement of StateHelper.cpp:116-197).  G_w is positive semi-definite (gauge directions):
  * unpivoted Cholesky (non-positive pivots leave zero rows — what gram::k_gram_chol does) drifts by ~1e-6;
  * diagonally pivoted Cholesky (what gram::k_gram_pchol does) stays at round-off, like the Householder triangle itself.
The device kernels are held to the same two numbers in tests/test_closed_loop.py::test_mode_a_closed_loop (GPU)."""
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from dev_mode_a_numerics import chol_pivoted, chol_zero_rows  # noqa: E402  (numpy restatements of the two factorisations)

from open_vins_amd import capi, closed_loop  # noqa: E402
from oracle import pyoracle  # noqa: E402


def _loop(stream, opts, factor):
    worst = [0.0]

    def upd(prob):
        v = capi.Views(prob)
        ref = pyoracle.msckf_update(opts, v, want_compressed=True)
        R, rc, cols = ref["H_comp"], ref["r_comp"], pyoracle.column_map(opts, v)
        D = R.shape[1]
        L = np.linalg.cholesky(prob.P[np.ix_(cols, cols)])
        A = np.hstack([R @ L, rc[:, None]])
        Rw = factor(A.T @ A, D)[:D]
        Hc = sla.solve_triangular(L, Rw[:, :D].T, lower=True, trans="T").T
        st, P1, dx = pyoracle.ekf_update(prob.P, Hc, Rw[:, D], cols, opts.sigma_pix ** 2)
        assert st == 0
        worst[0] = max(worst[0], np.linalg.norm(dx - ref["dx"]) / np.linalg.norm(ref["dx"]))
        out = pyoracle.apply_dx(opts, v, dx)
        out.update(P=P1, feat_status=ref["feat_status"])
        return out

    return closed_loop.run(stream, upd), worst[0]


def test_pivoting_is_what_keeps_mode_a_on_the_oracle_trajectory():
    opts = capi.default_options(chi2_multipler=1.0)
    stream = closed_loop.Stream(C=12, feats_per_frame=50, seed=7)
    base = closed_loop.run(stream, lambda prob: pyoracle.msckf_update(opts, capi.Views(prob)))
    piv, dx_piv = _loop(stream, opts, lambda G, D: chol_pivoted(G, D, 1e-15))
    unp, dx_unp = _loop(stream, opts, lambda G, D: chol_zero_rows(G))
    dev_piv, dev_unp = np.abs(piv["est"] - base["est"]).max(), np.abs(unp["est"] - base["est"]).max()
    print(f"closed loop, 52 frames: pivoted {dev_piv:.1e} (one-step dx {dx_piv:.1e}), unpivoted {dev_unp:.1e} (one-step dx {dx_unp:.1e})")
    assert dev_piv < 1e-11 and dx_piv < 1e-10
    assert dev_unp > 1e-8 and dx_unp > 1e-9
