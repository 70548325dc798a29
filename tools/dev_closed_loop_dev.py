"""Development check: deviation of the closed loops (host-fed and resident) from the oracle-driven loop, per compression route."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime first)
from open_vins_amd import capi, closed_loop  # noqa: E402
from open_vins_amd.updater import UpdaterMSCKF  # noqa: E402
from oracle import pyoracle  # noqa: E402

OPTS = dict(chi2_multipler=1.0)
stream = closed_loop.Stream(C=12, feats_per_frame=50, seed=7)
opts = capi.default_options(**OPTS)
ref = closed_loop.run(stream, lambda prob: pyoracle.msckf_update(opts, capi.Views(prob)))
for mode in ("tsqr", "gram", "cholqr"):
    os.environ["OVGPU_COMPRESS"] = mode
    up = UpdaterMSCKF(capi.default_options(**OPTS))

    def gpu_update(prob):
        up.set_problem(prob)
        return up.update()
    host = closed_loop.run(stream, gpu_update)
    up.close()
    up = UpdaterMSCKF(capi.default_options(**OPTS))
    res = closed_loop.run_resident(stream, up)
    up.close()
    dev_h = np.abs(host["est"] - ref["est"]).max(axis=tuple(range(1, host["est"].ndim)))
    dev_r = np.abs(res["est"] - ref["est"]).max(axis=tuple(range(1, res["est"].ndim)))
    print(mode, "host loop max dev %.2e used equal %s | resident max dev %.2e used equal %s" % (dev_h.max(), host["used"] == ref["used"], dev_r.max(), res["used"] == ref["used"]))
    print("   resident dev per frame:", " ".join("%.0e" % x for x in dev_r[::4]))
    print("   host     dev per frame:", " ".join("%.0e" % x for x in dev_h[::4]))
