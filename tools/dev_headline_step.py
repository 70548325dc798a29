import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
from open_vins_amd import capi, closed_loop
from open_vins_amd.updater import UpdaterMSCKF
from oracle import pyoracle
from test_gpu_fullsize import extended_precision_update, _rel, LD
F = int(sys.argv[1]) if len(sys.argv) > 1 else 800
stream = closed_loop.Stream(C=31, feats_per_frame=F, seed=3, K=2)
opts = capi.default_options(chi2_multipler=1.0)
up = UpdaterMSCKF(opts)
n = [0]
def upd(prob):
    v = capi.Views(prob)
    ref = pyoracle.msckf_update(opts, v, want_compressed=True)
    cols = pyoracle.column_map(opts, v)
    up.set_problem(prob)
    out = up.update()
    if n[0] % 3 == 0:
        P_true, dx_true = extended_precision_update(prob.P, cols, ref["H_comp"], ref["r_comp"], 1.0)
        ev = np.linalg.eigvalsh(prob.P[np.ix_(cols, cols)])
        same = np.array_equal(out["feat_status"], ref["feat_status"])
        print(f"frame {n[0]:2d} cond {ev[-1] / ev[0]:.1e} same gate {same} | P: gpu {_rel(out['P'].astype(LD), P_true):.1e} oracle {_rel(ref['P'].astype(LD), P_true):.1e} | dx: gpu {_rel(out['dx'].astype(LD), dx_true):.1e} "
              f"oracle {_rel(ref['dx'].astype(LD), dx_true):.1e} | gpu vs oracle dx {_rel(out['dx'], ref['dx']):.1e} |dx| {np.linalg.norm(ref['dx']):.1e} clone dev {np.abs(out['clone_q_p'] - ref['clone_q_p']).max():.1e}", flush=True)
    n[0] += 1
    return ref
closed_loop.run(stream, upd)
