#!/usr/bin/env python3
"""Developer script: assembling the batch of a 10k-feature update on the device (ovgpu_tracks_to_features) vs uploading the
same batch from host arrays (ovgpu_set_features; the host-side walk over the per-feature maps is NOT included in that)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
prob = synth.make_problem(2, F=F)
feat_of = np.repeat(np.arange(prob.F), np.diff(prob.meas_offsets))
perm = np.lexsort((prob.clone_idx, prob.cam_idx, feat_of))
prob.uv, prob.uvn = prob.uv.reshape(-1, 2)[perm].reshape(-1), prob.uvn.reshape(-1, 2)[perm].reshape(-1)
prob.clone_idx, prob.cam_idx = prob.clone_idx[perm], prob.cam_idx[perm]
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
up.set_problem(prob)
up.tracks_create(F + 16, 64)
clone_times = 50.0 + 0.1 * np.arange(prob.C)
uv2, uvn2 = prob.uv.reshape(-1, 2), prob.uvn.reshape(-1, 2)
t0 = time.perf_counter()
for cl in range(prob.C):
    for cam in range(prob.K):
        idx = np.flatnonzero((prob.clone_idx == cl) & (prob.cam_idx == cam))
        if len(idx):
            up.tracks_append(clone_times[cl], feat_of[idx], np.full(len(idx), cam), uv2[idx], uvn2[idx])
print("appended %d observations of %d tracks in %d calls: %.1f ms total (incl. the numpy selection)" % (len(feat_of), F, prob.C * prob.K, 1e3 * (time.perf_counter() - t0)))
ids = np.arange(F)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); up.tracks_to_features(ids, clone_times); up.synchronize(); ts.append(time.perf_counter() - t0)
print("ovgpu_tracks_to_features: %.2f ms" % (1e3 * min(ts)))
ts = []
for _ in range(5):
    t0 = time.perf_counter(); up.set_features(prob); up.synchronize(); ts.append(time.perf_counter() - t0)
print("ovgpu_set_features (flat host arrays, %.1f MB over PCIe): %.2f ms" % (len(feat_of) * 18 / 1e6, 1e3 * min(ts)))
