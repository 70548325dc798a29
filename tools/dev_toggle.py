import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
seq = [int(x) for x in sys.argv[1].split(",")]
prob = synth.make_problem(3)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
up.set_problem(prob)
for v in seq:
    up.debug_option("chol_flag_sync", v)
    for i in range(3):
        up.reset_state(); up.update_async(); up.synchronize()
        print("flag_sync", v, "update", i, "ok", flush=True)
up.close()
