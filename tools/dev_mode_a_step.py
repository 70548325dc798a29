"""Developer check (GPU): one-step error of mode A (Gram-route factor) along the oracle-driven closed loop."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from open_vins_amd import capi, closed_loop
from open_vins_amd.updater import UpdaterMSCKF
from oracle import pyoracle
stream = closed_loop.Stream(C=12, feats_per_frame=50, seed=7)
opts = capi.default_options(chi2_multipler=1.0)
ups = {r: UpdaterMSCKF(capi.default_options(chi2_multipler=1.0, compress_route=r)) for r in (capi.COMPRESS_GRAM, capi.COMPRESS_TSQR)}
n = [0]
def upd(prob):
    v = capi.Views(prob)
    ref = pyoracle.msckf_update(opts, v, want_compressed=True)
    G, g = ref["H_comp"].T @ ref["H_comp"], ref["H_comp"].T @ ref["r_comp"]
    line = f"frame {n[0]:2d} cond(P_DD) {np.linalg.cond(prob.P[30:, 30:]):.1e}"
    for r, up in ups.items():
        up.set_problem(prob)
        cmp = up.compress()
        H, rr = cmp["H"], cmp["r"]
        st, P1, dx = pyoracle.ekf_update(prob.P, H, rr, cmp["col_cov_id"], 1.0)
        nz = int((np.abs(H).sum(axis=1) == 0).sum())
        line += f" | route {r}: G {np.linalg.norm(H.T @ H - G) / np.linalg.norm(G):.1e} g {np.linalg.norm(H.T @ rr - g) / np.linalg.norm(g):.1e} P {np.linalg.norm(P1 - ref['P']) / np.linalg.norm(ref['P']):.1e} dx {np.linalg.norm(dx - ref['dx']) / np.linalg.norm(ref['dx']):.1e} zero rows {nz} max|H| {np.abs(H).max():.1e}"
    if n[0] < 12 or n[0] % 10 == 0:
        print(line, flush=True)
    n[0] += 1
    return ref
closed_loop.run(stream, upd)
