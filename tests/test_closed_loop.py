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


@pytest.mark.gpu
def test_resident_window_matches_the_host_loop(stream, oracle_run):
    """cloning, propagation of the new block, marginalisation and the update all on the RESIDENT covariance
    (uploaded once, 52 frames): same accept sets and trajectory as the oracle-driven host loop."""
    from open_vins_amd.updater import UpdaterMSCKF
    up = UpdaterMSCKF(capi.default_options(**OPTS))
    res = closed_loop.run_resident(stream, up)
    post = up.get_state(P=True)
    up.close()
    assert res["used"] == oracle_run["used"]
    assert np.abs(res["est"] - oracle_run["est"]).max() < 1e-8
    # Phi P Phi^T + Q is formed entry by entry like the reference's (StateHelper.cpp:85-90): symmetric up to rounding only
    assert post["P"].shape == (16 + 14 + 6 * stream.C,) * 2
    np.testing.assert_allclose(post["P"], post["P"].T, rtol=0, atol=1e-18)


@pytest.mark.gpu
def test_resident_window_and_track_store(stream, oracle_run):
    """SURVEY 8f N2 + N3 together: observations appended per frame to the device-resident feature database, lost tracks
    queried, their batch assembled on the device, window bookkeeping and update on the resident covariance — the trajectory
    of the oracle-driven host loop again."""
    from open_vins_amd.updater import UpdaterMSCKF
    up = UpdaterMSCKF(capi.default_options(**OPTS))
    res = closed_loop.run_resident(stream, up, track_store=True)
    assert up.tracks_count() == 0           # every track was used once and erased
    up.close()
    assert res["used"] == oracle_run["used"]
    assert np.abs(res["est"] - oracle_run["est"]).max() < 1e-8


@pytest.mark.gpu
def test_long_loop_with_imu_level_process_noise():
    """500 consecutive updates with the posterior fed back, process noise at IMU level (5e-5 rad, 1e-5 m per clone step): the window's
    clones are then correlated to 1e-5 of their prior sigma and cond(P_DD) grows over the run to the values of the conditioning sweep
    (tests/test_gpu_fullsize.py).  Same stream through the oracle and through the GPU's default (Gram / prior-whitened) route.
    A gate decision that falls within rounding of its threshold may differ; from there on the two filters see different measurements,
    so the comparison runs up to the first such frame (and requires it to be late) and bounds the divergence after it."""
    from open_vins_amd.updater import UpdaterMSCKF
    stream = closed_loop.Stream(C=12, feats_per_frame=40, seed=11, q_theta=5e-5, q_p=1e-5, T=512)
    opts = capi.default_options(**OPTS)
    ref = closed_loop.run(stream, lambda prob: pyoracle.msckf_update(opts, capi.Views(prob)))
    up = UpdaterMSCKF(opts)
    routes = []

    def gpu_update(prob):
        up.set_problem(prob)
        out = up.update()
        routes.append(out["route"])
        return out

    gpu = closed_loop.run(stream, gpu_update)
    up.close()
    assert len(ref["used"]) == 500 and np.mean(list(ref["used"].values())) > 25
    assert all(r == capi.COMPRESS_GRAM for r in routes)            # the prior never failed the pivot test: no Householder fall-back
    frames = sorted(ref["used"])
    same = [gpu["used"][t] == ref["used"][t] for t in frames]
    first_diff = same.index(False) if False in same else len(frames)
    assert first_diff >= 100, first_diff
    k = first_diff + (stream.C - 1)                                # index into est (est starts at the last initial clone)
    assert np.abs(gpu["est"][:k] - ref["est"][:k]).max() < 1e-8
    assert np.abs(gpu["est"] - ref["est"]).max() < 1e-4            # after a differing gate decision: same filter, one measurement apart
    a_g, a_o = closed_loop.ate(gpu), closed_loop.ate(ref)
    assert abs(a_g[0] - a_o[0]) < 1e-4 and abs(a_g[1] - a_o[1]) < 1e-3
