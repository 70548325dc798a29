"""Developer check of the fused per-feature kernel (k_featy.h) against the oracle and against the legacy kernels."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
from oracle import pyoracle

def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

for kw in (dict(cfg=2, F=300), dict(cfg=2, F=200, track="ragged", outlier_frac=0.3), dict(cfg=4, F=100), dict(cfg=2, F=150, C=11, K=1)):
    kw = dict(kw)
    prob = synth.make_problem(kw.pop("cfg"), **kw)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    tri = pyoracle.triangulate(opts, v)
    ref = pyoracle.msckf_update(opts, v, given=tri)
    for legacy in (0, 1):
        up = UpdaterMSCKF(opts)
        up.debug_option("legacy_feature_kernel", legacy)
        up.set_problem(prob)
        up.set_triangulation(tri["p_FinG"], tri["p_FinA"], tri["anchor_meas"], tri["status"])
        out = up.update()
        gate = np.isfinite(ref["chi2"])
        same = np.array_equal(out["feat_status"], ref["feat_status"])
        print(kw, "legacy" if legacy else "fused ", "status same", same, "chi2", np.abs(out["chi2"][gate] / ref["chi2"][gate] - 1).max(),
              "dx", rel(out["dx"], ref["dx"]), "P", rel(out["P"], ref["P"]), "used", out["stats"]["n_used"], ref["stats"]["n_used"], flush=True)
        up.close()
