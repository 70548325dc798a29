"""BASELINE.json configs[0] (rpng_sim mono, 11-clone window + the new clone, ~50 MSCKF features per update) as a closed
loop: a sliding-window filter over the committed trajectory fixture, once with the CPU oracle and once with the GPU
library as the updater (open_vins_amd/closed_loop.py).  "ATE parity" = the two runs see identical measurement and
noise streams and must produce the same trajectory."""
import numpy as np
import pytest

from open_vins_amd import capi, closed_loop
from oracle import pyoracle

OPTS = dict(chi2_multipler=1.0)


@pytest.fixture(scope="module")
def stream():
    return closed_loop.Stream(C=12, feats_per_frame=50, seed=7)


@pytest.fixture(scope="module")
def oracle_run(stream):
    opts = capi.default_options(**OPTS)
    return closed_loop.run(stream, lambda prob: pyoracle.msckf_update(opts, capi.Views(prob)))


def test_oracle_filter_beats_dead_reckoning(stream, oracle_run):
    dead = closed_loop.run(stream, None)
    ate_dead, ate_filt = closed_loop.ate(dead), closed_loop.ate(oracle_run)
    assert len(oracle_run["used"]) == stream.T - stream.C
    assert np.mean(list(oracle_run["used"].values())) > 30          # most of the ~50 tracks pass the gate
    assert ate_filt[1] < 0.5 * ate_dead[1] and ate_filt[0] < 0.6 * ate_dead[0]
    assert ate_filt[1] < 0.05                                       # metres, over a 5 s window at 10 Hz


@pytest.mark.gpu
def test_ate_parity_gpu_vs_oracle(stream, oracle_run):
    from open_vins_amd.updater import UpdaterMSCKF
    up = UpdaterMSCKF(capi.default_options(**OPTS))

    def gpu_update(prob):
        up.set_problem(prob)
        return up.update()

    gpu = closed_loop.run(stream, gpu_update)
    up.close()
    assert gpu["used"] == oracle_run["used"]                        # identical accept sets in every frame
    assert np.abs(gpu["est"] - oracle_run["est"]).max() < 1e-8      # 52 consecutive updates, posterior fed back each time
    a_g, a_o = closed_loop.ate(gpu), closed_loop.ate(oracle_run)
    assert abs(a_g[0] - a_o[0]) < 1e-7 and abs(a_g[1] - a_o[1]) < 1e-8
