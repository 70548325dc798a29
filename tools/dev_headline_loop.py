import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from open_vins_amd import capi, closed_loop
from open_vins_amd.updater import UpdaterMSCKF
from oracle import pyoracle
F = int(sys.argv[1]) if len(sys.argv) > 1 else 800
stream = closed_loop.Stream(C=31, feats_per_frame=F, seed=3, K=2)
opts = capi.default_options(chi2_multipler=1.0)
ref = closed_loop.run(stream, lambda prob: pyoracle.msckf_update(opts, capi.Views(prob)))
def report(name, res):
    fr = sorted(ref["used"])
    diff = [t for t in fr if res["used"].get(t) != ref["used"][t]]
    dev = np.abs(res["est"] - ref["est"]).max(axis=1)
    print(name, "first differing frame", diff[:3], "used there", [(res["used"].get(t), ref["used"][t]) for t in diff[:3]], "dev per frame", " ".join(f"{d:.0e}" for d in dev[::4]), flush=True)
up = UpdaterMSCKF(opts)
def gpu_update(prob):
    up.set_problem(prob)
    return up.update()
report("host-fed GPU", closed_loop.run(stream, gpu_update))
up.close()
up = UpdaterMSCKF(opts)
report("resident", closed_loop.run_resident(stream, up))
up.close()
up = UpdaterMSCKF(opts)
report("resident + track store", closed_loop.run_resident(stream, up, track_store=True))
up.close()
