#!/usr/bin/env python3
"""Developer script: wall time of UpdaterSLAM::delayed_init (50 new features, cfg-2 state) on the GPU and in the oracle."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
from oracle import pyoracle

F = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 0
prob = synth.make_problem(2, F=F)
opts = capi.default_options(chi2_multipler=1.0)
up = UpdaterMSCKF(opts)
ts = []
for it in range(4):
    up.set_problem(prob)
    up.synchronize()
    t0 = time.perf_counter()
    out = up.delayed_init(rep)
    ts.append(time.perf_counter() - t0)
print("GPU  delayed_init F=%d rep=%d: %.2f ms (first %.2f), accepted %d, N %d -> %d" % (F, rep, 1e3 * min(ts[1:]), 1e3 * ts[0], (out["lm_cov_id"] >= 0).sum(), prob.N, out["N"]))
v = capi.Views(prob)
t0 = time.perf_counter()
ref = pyoracle.slam_delayed_init(opts, v, feat_rep=rep)
print("oracle (1 core): %.1f ms, accepted %d" % (1e3 * (time.perf_counter() - t0), (ref["lm_cov_id"] >= 0).sum()))
