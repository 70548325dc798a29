"""The checker of the device track store (oracle/featdb_oracle.py) against the REFERENCE'S OWN ov_core::FeatureDatabase
(oracle/_ref/libov_ref.so: FeatureDatabase.cpp / Feature.cpp compiled from /root/reference, driven through oracle/ref/ref_featdb.cpp):
random operation sequences of a front end's life — frames of observations, the three queries, get_oldest_timestamp, the two
clean-ups, used features leaving — answer for answer, byte for byte.  CPU only; the device legs are in test_gpu_parity.py."""
import numpy as np
import pytest

from oracle import featdb_oracle as fo

needs_ref = pytest.mark.skipif(not fo.ref_available(), reason="oracle/_ref/libov_ref.so cannot be built or found here")


@needs_ref
@pytest.mark.parametrize("seed,cams,ooo", [(0, 1, False), (1, 2, False), (2, 3, False), (3, 2, True), (4, 4, True)])
def test_model_equals_the_reference_class(seed, cams, ooo):
    rng = np.random.default_rng(seed)
    ops = fo.random_script(rng, n_frames=45, n_cams=cams, n_ids=25, out_of_order=ooo)
    ref = fo.replay(ops, fo.RefFeatureDatabase())
    mod = fo.replay(ops, fo.FeatureDatabaseModel())
    fo.assert_same_log(ref, mod, f"seed {seed}")
    # the script does exercise what it is meant to: every query answers non-trivially somewhere, clean-ups drop features
    q = [e for e in ref if e[0] == "query"]
    assert any(len(e[2]) for e in q) and any(len(e[3]) for e in q) and any(len(e[4]) for e in q)
    assert any(e[0] == "cleanup" and e[3] > 0 for e in ref)


@needs_ref
def test_reference_semantics_first_and_last_are_positions_not_extremes():
    """FeatureDatabase.cpp:105 / :146 read at(size - 1) / at(0): with times appended out of order the answers follow the POSITION."""
    for db in (fo.RefFeatureDatabase(), fo.FeatureDatabaseModel()):
        for t in (10.2, 10.0, 10.1):  # one camera: first = 10.2, last = 10.1, smallest = 10.0
            db.update_feature(7, t, 0, 1, 2, .1, .2)
        assert list(db.containing_older(10.15)) == []      # first (10.2) is not < 10.15 although 10.0 is stored
        assert list(db.containing_older(10.25)) == [7]
        assert list(db.not_containing_newer(10.15)) == [7]  # last (10.1) is not >= 10.15 although 10.2 is stored
        assert list(db.not_containing_newer(10.05)) == []
        assert db.oldest() == 10.2
        assert list(db.containing(10.0)) == [7] and list(db.containing(10.05)) == []
        assert db.cleanup_measurements(10.1, False) == 0 and list(db.get_feature(7)["timestamps"]) == [10.2]
        assert db.cleanup_measurements(10.2, True) == 1 and db.size() == 0 and db.oldest() == -1.0


def test_model_alone_on_an_empty_database():
    db = fo.FeatureDatabaseModel()
    assert db.size() == 0 and db.oldest() == -1.0 and db.get_feature(3) is None
    assert len(db.containing(1.0)) == 0 and len(db.containing_older(1.0)) == 0 and len(db.not_containing_newer(1.0)) == 0
    assert db.cleanup_measurements(5.0) == 0
    db.erase([1, 2])
