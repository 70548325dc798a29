"""ov_core::FeatureDatabase for the tests of the device track store (ovgpu_tracks_*, csrc/k_tracks.h).  TEST INFRASTRUCTURE ONLY.

Two implementations of one interface:

* `RefFeatureDatabase` — the REFERENCE'S OWN class (ov_core/src/feat/FeatureDatabase.cpp, Feature.cpp, compiled from /root/reference
  into oracle/_ref/libov_ref.so) behind the C driver oracle/ref/ref_featdb.cpp.  Available wherever that library is (built here,
  shipped to the GPU box with the snapshot).
* `FeatureDatabaseModel` — a pure-Python restatement of the same functions (each citing the lines it follows), pinned against the
  class above by tests/test_track_store_cpu.py on random operation sequences; the fallback where the library is missing.

Interface (ids come back ascending; a feature's observations camera by camera in ascending camera id, each camera's list as stored):
  update_feature(id, t, cam, u, v, un, vn)      FeatureDatabase.cpp:59-85
  not_containing_newer(t)                       :87-126    no camera whose LAST observation is >= t
  containing_older(t)                           :128-167   a camera whose FIRST observation is < t
  containing(t)                                 :169-209   an observation with time == t
  oldest()                                      :265-276   min over the cameras' FIRST observations, -1 if none
  cleanup_measurements(t, exact)                :226-263   drop time <= t (exact: == t); features left empty leave
  erase(ids)                                    :211-224   to_delete + cleanup()
  get_feature(id)                               :41-57
"""
from __future__ import annotations

import ctypes as C

import numpy as np


class FeatureDatabaseModel:
    def __init__(self):
        self.feats = {}  # id -> {cam: [(t, u, v, un, vn), ...]}   (a camera key stays when its list empties, as in Feature::timestamps)

    def update_feature(self, fid, t, cam, u, v, un, vn):
        # FeatureDatabase.cpp:66-84: append to the feature's vectors of that camera, or create the feature
        self.feats.setdefault(int(fid), {}).setdefault(int(cam), []).append((float(t), np.float32(u), np.float32(v), np.float32(un), np.float32(vn)))

    def size(self):
        return len(self.feats)

    def not_containing_newer(self, t):
        out = []
        for fid, cams in self.feats.items():
            newer = any(len(obs) > 0 and obs[-1][0] >= t for obs in cams.values())  # :103-108
            if not newer:
                out.append(fid)
        return np.array(sorted(out), np.int64)

    def containing_older(self, t):
        out = [fid for fid, cams in self.feats.items() if any(len(obs) > 0 and obs[0][0] < t for obs in cams.values())]  # :144-149
        return np.array(sorted(out), np.int64)

    def containing(self, t):
        out = [fid for fid, cams in self.feats.items() if any(o[0] == t for obs in cams.values() for o in obs)]  # :186-191
        return np.array(sorted(out), np.int64)

    def oldest(self):
        firsts = [obs[0][0] for cams in self.feats.values() for obs in cams.values() if len(obs) > 0]  # :268-273
        return min(firsts) if firsts else -1.0

    def cleanup_measurements(self, t, exact=False):
        erased = 0
        for fid in list(self.feats):
            cams = self.feats[fid]
            for cam in cams:  # Feature.cpp:84-110 (time <= t goes) / :55-82 (time == t goes)
                cams[cam] = [o for o in cams[cam] if not ((o[0] == t) if exact else (o[0] <= t))]
            if sum(len(obs) for obs in cams.values()) < 1:  # FeatureDatabase.cpp:236-238, :255-257
                del self.feats[fid]
                erased += 1
        return erased

    def erase(self, ids):
        for fid in ids:
            self.feats.pop(int(fid), None)

    def get_feature(self, fid):
        cams = self.feats.get(int(fid))
        if cams is None:
            return None
        rows = [(o[0], cam, o[1], o[2], o[3], o[4]) for cam in sorted(cams) for o in cams[cam]]
        return _pack(rows)


def _pack(rows):
    return dict(timestamps=np.array([r[0] for r in rows], np.float64), cam_id=np.array([r[1] for r in rows], np.int32),
                uv=np.array([[r[2], r[3]] for r in rows], np.float32).reshape(-1, 2), uvn=np.array([[r[4], r[5]] for r in rows], np.float32).reshape(-1, 2))


def camera_major(feat):
    """A device track (observations in append order, cameras interleaved) in the form get_feature() returns: camera by camera in
    ascending camera id, each camera's observations in their stored order (a stable sort by camera)."""
    if feat is None:
        return None
    order = np.argsort(feat["cam_id"], kind="stable")
    return {k: v[order] for k, v in feat.items()}


def ref_available():
    from . import pyref
    return pyref.available()


class RefFeatureDatabase:
    """The reference's ov_core::FeatureDatabase through oracle/ref/ref_featdb.cpp."""

    def __init__(self):
        from . import pyref
        lib = pyref.load()
        i64p = C.POINTER(C.c_int64)
        lib.ref_featdb_create.restype = C.c_void_p
        lib.ref_featdb_create.argtypes = []
        lib.ref_featdb_destroy.restype = None
        lib.ref_featdb_destroy.argtypes = [C.c_void_p]
        lib.ref_featdb_update_feature.restype = None
        lib.ref_featdb_update_feature.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        lib.ref_featdb_size.restype = C.c_int
        lib.ref_featdb_size.argtypes = [C.c_void_p]
        lib.ref_featdb_query.restype = C.c_int
        lib.ref_featdb_query.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, i64p]
        lib.ref_featdb_oldest.restype = C.c_double
        lib.ref_featdb_oldest.argtypes = [C.c_void_p]
        lib.ref_featdb_cleanup.restype = None
        lib.ref_featdb_cleanup.argtypes = [C.c_void_p, C.c_double, C.c_int]
        lib.ref_featdb_erase.restype = None
        lib.ref_featdb_erase.argtypes = [C.c_void_p, C.c_int, i64p]
        lib.ref_featdb_get_feature.restype = C.c_int
        lib.ref_featdb_get_feature.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        self.lib = lib
        self.h = C.c_void_p(lib.ref_featdb_create())

    def __del__(self):
        try:
            if self.h:
                self.lib.ref_featdb_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def update_feature(self, fid, t, cam, u, v, un, vn):
        self.lib.ref_featdb_update_feature(self.h, int(fid), float(t), int(cam), float(u), float(v), float(un), float(vn))

    def size(self):
        return self.lib.ref_featdb_size(self.h)

    def _query(self, mode, t):
        n = self.lib.ref_featdb_query(self.h, mode, float(t), 0, None)
        ids = np.zeros(max(n, 1), np.int64)
        n = self.lib.ref_featdb_query(self.h, mode, float(t), n, ids.ctypes.data_as(C.POINTER(C.c_int64)))
        return ids[:n]

    def not_containing_newer(self, t):
        return self._query(0, t)

    def containing_older(self, t):
        return self._query(1, t)

    def containing(self, t):
        return self._query(2, t)

    def oldest(self):
        return self.lib.ref_featdb_oldest(self.h)

    def cleanup_measurements(self, t, exact=False):
        before = self.size()
        self.lib.ref_featdb_cleanup(self.h, float(t), 1 if exact else 0)
        return before - self.size()

    def erase(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int64)
        self.lib.ref_featdb_erase(self.h, len(a), a.ctypes.data_as(C.POINTER(C.c_int64)))

    def get_feature(self, fid):
        n = self.lib.ref_featdb_get_feature(self.h, int(fid), 0, None, None, None, None)
        if n < 0:
            return None
        out = dict(timestamps=np.zeros(n), cam_id=np.zeros(n, np.int32), uv=np.zeros((n, 2), np.float32), uvn=np.zeros((n, 2), np.float32))
        self.lib.ref_featdb_get_feature(self.h, int(fid), n, out["timestamps"].ctypes.data_as(C.POINTER(C.c_double)),
                                        out["cam_id"].ctypes.data_as(C.POINTER(C.c_int32)), out["uv"].ctypes.data_as(C.POINTER(C.c_float)),
                                        out["uvn"].ctypes.data_as(C.POINTER(C.c_float)))
        return out


def make_checker():
    """The reference's class where its library exists, the model otherwise."""
    return RefFeatureDatabase() if ref_available() else FeatureDatabaseModel()


def random_script(rng, n_frames=40, n_cams=2, n_ids=30, p_seen=0.6, window=8, out_of_order=False):
    """A front end's life as a list of operations, for both the checker and the device store:
      ("frame", t, [(id, cam, u, v, un, vn), ...])   the observations of one camera frame time (all cameras), FeatureDatabase::update_feature each
      ("query", t) / ("oldest",)                     every query at time t
      ("cleanup", t, exact)                          cleanup_measurements(_exact)
      ("erase", [ids])                               used features leave
    Times are 0.1 s apart; the clean-ups follow VioManager.cpp:589 (the time leaving a window of `window` frames) and now and then an
    exact one (UpdaterZeroVelocity.cpp:257).  out_of_order: some frames carry a time OLDER than the previous one, so that "first" /
    "last" of a camera's vector differ from its smallest / largest time."""
    ops = []
    alive = list(range(100, 100 + n_ids))
    next_id = 100 + n_ids
    times = []
    for f in range(n_frames):
        t = round(10.0 + 0.1 * f, 6)
        if out_of_order and f > 2 and rng.random() < 0.2:
            t = round(times[-2] - 0.05, 6)
        times.append(t)
        obs = []
        for fid in alive:
            for cam in range(n_cams):
                if rng.random() < (p_seen if fid % 4 else 0.1):  # every fourth track is seen rarely: the clean-ups empty and drop it
                    u, v = rng.uniform(0, 752), rng.uniform(0, 480)
                    obs.append((fid, cam, np.float32(u), np.float32(v), np.float32((u - 367) / 458), np.float32((v - 248) / 457)))
        ops.append(("frame", t, obs))
        qt = [t, times[max(0, f - window)], t + 0.05, times[max(0, f - 3)]]
        for q in qt:
            ops.append(("query", q))
        ops.append(("oldest",))
        if f >= window:
            ops.append(("cleanup", times[f - window], False))
            ops.append(("query", times[f - window + 1]))
        if f % 7 == 5:
            ops.append(("cleanup", times[f - 1], True))
            ops.append(("oldest",))
        if f % 5 == 4 and alive:  # some tracks are used up / lost, new ones start (ids are never reused by a front end)
            k = max(1, len(alive) // 6)
            gone = [alive[i] for i in sorted(rng.choice(len(alive), size=k, replace=False))]
            ops.append(("erase", gone))
            alive = [a for a in alive if a not in gone] + list(range(next_id, next_id + k))
            next_id += k
    return ops


def replay(ops, db, dump_every_cleanup=True):
    """Runs a random_script on any implementation of the interface; returns the log of everything it answered."""
    log = []
    known = set()

    def dump():
        out = {}
        for fid in sorted(known):
            f = db.get_feature(fid)
            out[fid] = None if f is None else {k: np.array(v) for k, v in f.items()}
        return out

    for op in ops:
        if op[0] == "frame":
            _, t, obs = op
            if hasattr(db, "append_frame"):
                db.append_frame(t, obs)
            else:
                for (fid, cam, u, v, un, vn) in obs:
                    db.update_feature(fid, t, cam, u, v, un, vn)
            known.update(o[0] for o in obs)
            log.append(("size", db.size()))
        elif op[0] == "query":
            t = op[1]
            log.append(("query", t, np.array(db.not_containing_newer(t)), np.array(db.containing_older(t)), np.array(db.containing(t))))
        elif op[0] == "oldest":
            log.append(("oldest", float(db.oldest())))
        elif op[0] == "cleanup":
            erased = db.cleanup_measurements(op[1], op[2])
            log.append(("cleanup", op[1], op[2], int(erased), db.size()))
            if dump_every_cleanup:
                log.append(("dump", dump()))
        elif op[0] == "erase":
            db.erase(op[1])
            log.append(("size", db.size()))
    log.append(("dump", dump()))
    return log


def assert_same_log(a, b, what=""):
    assert len(a) == len(b), what
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0], (what, i)
        if x[0] == "query":
            assert x[1] == y[1]
            for k, name in ((2, "not_containing_newer"), (3, "containing_older"), (4, "containing")):
                assert np.array_equal(x[k], y[k]), (what, i, name, x[1], x[k], y[k])
        elif x[0] == "dump":
            assert x[1].keys() == y[1].keys(), (what, i)
            for fid in x[1]:
                fx, fy = x[1][fid], y[1][fid]
                assert (fx is None) == (fy is None), (what, i, fid)
                if fx is not None:
                    for k in ("timestamps", "cam_id", "uv", "uvn"):
                        assert np.array_equal(fx[k], fy[k]), (what, i, fid, k)  # bit-exact: bytes are moved, never computed
        else:
            assert x == y, (what, i, x, y)
