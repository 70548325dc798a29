"""Times of the device FeatureDatabase's per-frame operations at configs[3]'s scale: 10 000 live tracks, a 30-frame stereo window
(600 k observations resident).  Per frame of a live filter: append of the newest stereo frame, the two selection queries
(VioManager.cpp:366-378), cleanup_measurements of the time leaving the window (:589).  One JSON line."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from open_vins_amd import capi  # noqa: E402
from open_vins_amd.updater import UpdaterMSCKF  # noqa: E402

T, K, W = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 2, 30
up = UpdaterMSCKF(capi.default_options(), device=0)
up.tracks_create(T + 16, K * (W + 2))
rng = np.random.default_rng(0)
ids = np.arange(1, T + 1, dtype=np.int64)
fid = np.repeat(ids, K)
cam = np.tile(np.arange(K, dtype=np.int32), T)
times = [round(100.0 + 0.1 * f, 6) for f in range(W + 12)]


def frame(t, uv=None):
    uv = rng.uniform(0, 480, (T * K, 2)).astype(np.float32) if uv is None else uv
    up.tracks_append(t, fid, cam, uv, uvn_of(uv))


def uvn_of(uv):
    return (uv / 460).astype(np.float32)


for f in range(W):
    frame(times[f])
res = {"tracks": T, "cameras": K, "window_frames": W, "observations_resident": T * K * W}
acc = {"append_ms": [], "not_containing_newer_ms": [], "containing_ms": [], "cleanup_measurements_ms": [], "oldest_timestamp_ms": []}
for f in range(W, W + 10):
    uv_f = rng.uniform(0, 480, (T * K, 2)).astype(np.float32)
    t0 = time.perf_counter()
    frame(times[f], uv_f)
    t1 = time.perf_counter()
    a = up.tracks_not_containing_newer(times[f])
    t2 = time.perf_counter()
    b = up.tracks_containing(times[f - W])
    t3 = time.perf_counter()
    n = up.tracks_cleanup_measurements(times[f - W])
    t4 = time.perf_counter()
    o = up.tracks_oldest_timestamp()
    t5 = time.perf_counter()
    assert len(a) == 0 and len(b) == T and n == 0 and o == times[f - W + 1], (len(a), len(b), n, o)
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[k].append(1e3 * v)
res.update({k: round(float(np.median(v)), 4) for k, v in acc.items()})
res["note"] = "host to host through the Python mirror (20 B per observation in, id arrays out); queries return id arrays"
print(json.dumps(res))
up.close()
