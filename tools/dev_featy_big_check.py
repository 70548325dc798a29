"""Developer check of the multi-pass per-feature kernel (k_featy_big.h): against the one-pass kernel on batches both hold (debug option
"featy_big": 1 = same tile budget, 2 = 5 tiles per wavefront, i.e. several passes on short tracks), and against the oracle at the
configs[4] geometry."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
from oracle import pyoracle

def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

opts = capi.default_options(chi2_multipler=1.0)
for kw in (dict(cfg=2, F=300), dict(cfg=2, F=200, track="ragged", outlier_frac=0.3), dict(cfg=4, F=100), dict(cfg=2, F=150, C=11, K=1)):
    kw = dict(kw)
    prob = synth.make_problem(kw.pop("cfg"), **kw)
    outs = {}
    for big in (0, 1, 2):
        up = UpdaterMSCKF(opts)
        up.debug_option("featy_big", big)
        up.set_problem(prob)
        outs[big] = up.update()
        up.close()
    for big in (1, 2):
        o, r = outs[big], outs[0]
        gate = np.isfinite(r["chi2"])
        print(kw, "big", big, "status same", np.array_equal(o["feat_status"], r["feat_status"]), "chi2", np.abs(o["chi2"][gate] / r["chi2"][gate] - 1).max(),
              "dx", rel(o["dx"], r["dx"]), "P", rel(o["P"], r["P"]), "used", o["stats"]["n_used"], r["stats"]["n_used"], flush=True)

F5 = int(sys.argv[1]) if len(sys.argv) > 1 else 120
prob = synth.make_problem(5, F=F5)
v = capi.Views(prob)
t = time.time()
tri = pyoracle.triangulate(opts, v)
ref = pyoracle.msckf_update(opts, v, given=tri)
print(f"oracle at cfg5, {F5} features: {time.time() - t:.1f} s", flush=True)
up = UpdaterMSCKF(opts)
up.set_problem(prob)
up.set_triangulation(tri["p_FinG"], tri["p_FinA"], tri["anchor_meas"], tri["status"])
out = up.update()
gate = np.isfinite(ref["chi2"])
print("cfg5", F5, "status same", np.array_equal(out["feat_status"], ref["feat_status"]), "chi2", np.abs(out["chi2"][gate] / ref["chi2"][gate] - 1).max(),
      "dx", rel(out["dx"], ref["dx"]), "P", rel(out["P"], ref["P"]), "used", out["stats"]["n_used"], ref["stats"]["n_used"], flush=True)
for _ in range(3):
    up.reset_state(); up.update_async()
up.synchronize()
t = time.time()
for _ in range(5):
    up.reset_state(); up.update_async()
up.synchronize()
print(f"cfg5 {F5} features: {(time.time() - t) / 5 * 1e3:.2f} ms / update")
up.close()
