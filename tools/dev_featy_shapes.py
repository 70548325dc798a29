import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
prob = synth.make_problem(3)
ref = None
for shape in (0, 1, 2):
    up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
    up.debug_option("featy_shape", shape)
    up.set_problem(prob)
    out = up.update()
    if ref is None: ref = out
    for _ in range(3):
        up.reset_state(); up.update_async()
    up.synchronize(); up.kernel_times(reset=True)
    for _ in range(20):
        up.reset_state(); up.update_async()
    up.synchronize()
    kt = up.kernel_times(reset=True)
    print("shape", shape, "stage us", kt["ms_system"] * 1e3, "update us", kt["ms_update"] * 1e3, "status same", np.array_equal(out["feat_status"], ref["feat_status"]),
          "chi2 diff", np.nanmax(np.abs(out["chi2"] / ref["chi2"] - 1)), "dx", np.linalg.norm(out["dx"] - ref["dx"]) / np.linalg.norm(ref["dx"]), flush=True)
    up.close()
