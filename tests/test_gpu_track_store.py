"""The device track store (ovgpu_tracks_*, csrc/k_tracks.h) as a whole FeatureDatabase: the queries
(features_not_containing_newer / _containing_older / _containing, get_oldest_timestamp), the clean-ups (cleanup_measurements /
_exact), get_feature_clone — through the C ABI on an MI355X against the REFERENCE'S OWN ov_core::FeatureDatabase
(oracle/_ref/libov_ref.so, oracle/ref/ref_featdb.cpp; the Python model of oracle/featdb_oracle.py, pinned to that class by
tests/test_track_store_cpu.py, where the library is missing).  Index and byte work: every answer identical, every stored byte identical."""
import numpy as np
import pytest

from open_vins_amd import capi
from open_vins_amd.updater import UpdaterMSCKF
from oracle import featdb_oracle as fo

pytestmark = pytest.mark.gpu


class DeviceStore:
    """The interface of oracle/featdb_oracle.py on the library's track store."""

    def __init__(self, max_tracks, max_obs):
        self.up = UpdaterMSCKF(capi.default_options(), device=0)
        self.up.tracks_create(max_tracks, max_obs)

    def close(self):
        self.up.close()

    def append_frame(self, t, obs):  # one ovgpu_tracks_append per frame time, all cameras (FeatureDatabase::update_feature per observation)
        if not obs:
            return
        self.up.tracks_append(t, [o[0] for o in obs], [o[1] for o in obs], np.array([[o[2], o[3]] for o in obs], np.float32),
                              np.array([[o[4], o[5]] for o in obs], np.float32))

    def size(self):
        return self.up.tracks_count()

    def not_containing_newer(self, t):
        return self.up.tracks_not_containing_newer(t)

    def containing_older(self, t):
        return self.up.tracks_containing_older(t)

    def containing(self, t):
        return self.up.tracks_containing(t)

    def oldest(self):
        return self.up.tracks_oldest_timestamp()

    def cleanup_measurements(self, t, exact=False):
        return self.up.tracks_cleanup_measurements(t, exact)

    def erase(self, ids):
        self.up.tracks_erase(ids)

    def get_feature(self, fid):
        return fo.camera_major(self.up.tracks_get_feature(fid))


@pytest.mark.parametrize("seed,cams,ooo", [(0, 1, False), (1, 2, False), (2, 4, False), (3, 2, True), (4, 3, True)])
def test_track_store_answers_like_the_reference_database(seed, cams, ooo):
    rng = np.random.default_rng(seed)
    ops = fo.random_script(rng, n_frames=45, n_cams=cams, n_ids=25, out_of_order=ooo)
    want = fo.replay(ops, fo.make_checker())
    dev = DeviceStore(max_tracks=64, max_obs=cams * 50)
    try:
        got = fo.replay(ops, dev)
    finally:
        dev.close()
    fo.assert_same_log(want, got, f"seed {seed}")
    assert any(e[0] == "cleanup" and e[3] > 0 for e in got) and any(e[0] == "query" and len(e[4]) for e in got)


def test_cleanup_keeps_a_long_lived_track_inside_max_obs():
    """VioManager.cpp:589: cleanup_measurements(time of the clone about to leave) once per frame.  A track observed for 300 frames lives
    in a store of 12 observations per camera with it, and overflows without it (OVGPU_ERR_CAPACITY, nothing appended)."""
    cams, window, frames = 2, 10, 300
    chk = fo.make_checker()
    dev = DeviceStore(max_tracks=8, max_obs=cams * (window + 2))
    try:
        times = [round(5.0 + 0.05 * f, 6) for f in range(frames)]
        for f, t in enumerate(times):
            obs = [(fid, cam, np.float32(f + cam), np.float32(fid), np.float32(0.01 * f), np.float32(0.1 * cam)) for fid in (1, 2, 3) for cam in range(cams)
                   if not (fid == 3 and f % 2)]
            dev.append_frame(t, obs)
            for o in obs:
                chk.update_feature(o[0], t, *o[1:])
            if f >= window:
                assert dev.cleanup_measurements(times[f - window]) == chk.cleanup_measurements(times[f - window])
        assert dev.size() == chk.size() == 3 and dev.oldest() == chk.oldest() == times[frames - window]
        for fid in (1, 2, 3):
            a, b = dev.get_feature(fid), chk.get_feature(fid)
            for k in ("timestamps", "cam_id", "uv", "uvn"):
                assert np.array_equal(a[k], b[k]), (fid, k)
        np.testing.assert_array_equal(dev.containing(times[frames - window]), chk.containing(times[frames - window]))
        # without the clean-up the same track fills its slot: the append is refused as a whole
        for f in range(2):
            dev.append_frame(times[-1] + 0.05 * (f + 1), [(1, cam, 0, 0, 0, 0) for cam in range(cams)])
        with pytest.raises(capi.OvgpuError) as e:
            dev.append_frame(times[-1] + 1.0, [(1, cam, 0, 0, 0, 0) for cam in range(cams)])
        assert e.value.code == capi.ERR_CAPACITY
        assert len(dev.get_feature(1)["timestamps"]) == cams * (window + 2)
    finally:
        dev.close()


def test_track_store_queries_on_an_empty_store_and_unknown_ids():
    dev = DeviceStore(max_tracks=4, max_obs=6)
    try:
        assert dev.size() == 0 and dev.oldest() == -1.0 and dev.get_feature(5) is None
        assert len(dev.containing(1.0)) == 0 and len(dev.containing_older(1.0)) == 0 and len(dev.not_containing_newer(1.0)) == 0
        assert dev.cleanup_measurements(10.0) == 0 and dev.cleanup_measurements(10.0, exact=True) == 0
        dev.append_frame(2.0, [(9, 0, 1, 2, 3, 4)])
        assert dev.oldest() == 2.0 and list(dev.containing(2.0)) == [9] and list(dev.not_containing_newer(2.5)) == [9]
        assert dev.cleanup_measurements(2.0, exact=True) == 1 and dev.size() == 0 and dev.oldest() == -1.0
        dev.append_frame(3.0, [(9, 1, 5, 6, 7, 8)])  # the id starts a fresh track in the slot that was freed
        f = dev.get_feature(9)
        assert list(f["timestamps"]) == [3.0] and list(f["cam_id"]) == [1] and f["uv"].tolist() == [[5.0, 6.0]] and f["uvn"].tolist() == [[7.0, 8.0]]
    finally:
        dev.close()
