"""The experimental compact-WY TSQR leaf (options.tsqr_leaf_blocked, k_tsqr_blk.h) against the default leaf kernel."""
import numpy as np
import pytest

from open_vins_amd import capi, synth


@pytest.mark.gpu
def test_blocked_leaf_gives_the_same_compressed_system():
    from open_vins_amd.updater import UpdaterMSCKF
    for name, kw in (("cfg2", dict(cfg=2, F=300)), ("d86", dict(cfg=2, F=200, K=1, C=12)), ("cfg4", dict(cfg=4, F=120))):
        kw = dict(kw)
        prob = synth.make_problem(kw.pop("cfg"), **kw)
        res = {}
        for blocked in (0, 1):
            up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0, tsqr_leaf_blocked=blocked, compress_route=capi.COMPRESS_TSQR))
            up.set_problem(prob)
            c = up.compress()
            o = up.update()
            up.close()
            H, r = c["H"], c["r"]
            assert np.abs(np.tril(H, -1)).max() == 0.0
            res[blocked] = (H.T @ H, H.T @ r, o["dx"], o["P"])
        for a, b in zip(res[0], res[1]):
            assert np.linalg.norm(a - b) <= 1e-11 * np.linalg.norm(a), name
