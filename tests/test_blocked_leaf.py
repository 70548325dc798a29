"""The experimental compact-WY TSQR leaf (OVGPU_TSQR_LEAF=blocked, k_tsqr_blk.h) against the default leaf kernel.
The switch is read once per process, so the comparison runs in a child interpreter."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import numpy as np, sys
sys.path.insert(0, %r)
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
out = {}
for name, kw in (("cfg2", dict(cfg=2, F=300)), ("d86", dict(cfg=2, F=200, K=1, C=12)), ("cfg4", dict(cfg=4, F=120))):
    kw = dict(kw)
    prob = synth.make_problem(kw.pop("cfg"), **kw)
    up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
    up.set_problem(prob)
    c = up.compress()
    o = up.update()
    up.close()
    H, r = c["H"], c["r"]
    assert np.abs(np.tril(H, -1)).max() == 0.0
    out[name] = (H.T @ H, H.T @ r, o["dx"], o["P"])
np.savez(sys.argv[1], **{k + "_" + str(i): v for k, t in out.items() for i, v in enumerate(t)})
""" % ROOT


@pytest.mark.gpu
def test_blocked_leaf_gives_the_same_compressed_system(tmp_path):
    import numpy as np
    res = {}
    for mode in ("default", "blocked"):
        env = dict(os.environ)
        env.pop("OVGPU_TSQR_LEAF", None)
        if mode == "blocked":
            env["OVGPU_TSQR_LEAF"] = "blocked"
        f = str(tmp_path / (mode + ".npz"))
        subprocess.check_call([sys.executable, "-c", CHILD, f], env=env, cwd=ROOT)
        res[mode] = np.load(f)
    for k in res["default"].files:
        a, b = res["default"][k], res["blocked"][k]
        assert np.linalg.norm(a - b) <= 1e-11 * np.linalg.norm(a), k
