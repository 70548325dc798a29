"""Developer check (GPU): the update through two shapes of the per-feature kernel (ovgpu_debug_option "featy_shape") on the same batches."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
sa, sb = int(sys.argv[1]), int(sys.argv[2])
for kw in (dict(cfg=3, F=400), dict(cfg=2, F=300, track="ragged", outlier_frac=0.3), dict(cfg=2, F=200, K=1, C=12), dict(cfg=2, F=64), dict(cfg=3, F=2000)):
    prob = synth.make_problem(kw.pop("cfg"), **kw)
    outs = []
    for shape in (sa, sb):
        up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0)); up.debug_option("featy_shape", shape); up.set_problem(prob); o = up.update(); outs.append(o); up.close()
    a, b = outs
    gm = np.isfinite(a["chi2"]) & np.isfinite(b["chi2"])
    print(kw, "status equal", np.array_equal(a["feat_status"], b["feat_status"]), "n_used", a["stats"]["n_used"], b["stats"]["n_used"],
          "chi2 rel %.1e" % np.max(np.abs(a["chi2"][gm] - b["chi2"][gm]) / np.abs(a["chi2"][gm])),
          "dx rel %.1e" % (np.linalg.norm(a["dx"] - b["dx"]) / np.linalg.norm(a["dx"])), "P rel %.1e" % (np.linalg.norm(a["P"] - b["P"]) / np.linalg.norm(a["P"])))
