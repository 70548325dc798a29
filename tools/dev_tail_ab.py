"""Developer A/B (GPU): the factorisations reading their inputs at the source / follow-first tail (debug option "fuse_chol_inputs") and the
stage events' period, same process, interleaved repeats; plus the results of both forms side by side."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
F = int(sys.argv[2]) if len(sys.argv) > 2 else None
prob = synth.make_problem(cfg, F=F) if F else synth.make_problem(cfg)
opts = capi.default_options(chi2_multipler=1.0)
up = UpdaterMSCKF(opts)
up.set_problem(prob)

def run(fuse, period, steps=40):
    up.debug_option("fuse_chol_inputs", fuse)
    up.debug_option("stage_timing_period", period)
    for _ in range(5):
        up.reset_state(); up.update_async()
    up.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        up.reset_state(); up.update_async()
    up.synchronize()
    return (time.perf_counter() - t) / steps * 1e3

res = {}
for rep in range(5):
    for key in ((0, 1), (1, 1), (0, 4), (1, 4), (1, 1000000)):
        res.setdefault(key, []).append(run(*key))
for key, v in res.items():
    print(f"fuse {key[0]} period {key[1]:>7}: median {np.median(v):.4f} ms  min {min(v):.4f}  all {' '.join(f'{x:.4f}' for x in v)}")
outs = []
for fuse in (0, 1):
    up.debug_option("fuse_chol_inputs", fuse)
    up.reset_state()
    outs.append(up.update())
a, b = outs
rel = lambda x, y: np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300)
print("fused vs assembled: status same", np.array_equal(a["feat_status"], b["feat_status"]), "dx", rel(b["dx"], a["dx"]), "P", rel(b["P"], a["P"]),
      "P symmetric", np.array_equal(b["P"], b["P"].T))
up.close()
