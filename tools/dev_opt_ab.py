"""Developer A/B (GPU) of ovgpu_debug_option switches, same process, interleaved repeats: python tools/dev_opt_ab.py cfg F name=v0,v1 [name=...]"""
import sys, os, time, itertools
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF

cfg = int(sys.argv[1]); F = int(sys.argv[2]) or None
axes = [(a.split("=")[0], [int(x) for x in a.split("=")[1].split(",")]) for a in sys.argv[3:]]
prob = synth.make_problem(cfg, F=F) if F else synth.make_problem(cfg)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
up.set_problem(prob)
up.debug_option("stage_timing_period", 4)
combos = list(itertools.product(*[v for _, v in axes]))

def run(combo, steps=40):
    for (name, _), v in zip(axes, combo):
        up.debug_option(name, v)
    for _ in range(5):
        up.reset_state(); up.update_async()
    up.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        up.reset_state(); up.update_async()
    up.synchronize()
    return (time.perf_counter() - t) / steps * 1e3

res = {c: [] for c in combos}
outs = {}
for rep in range(5):
    for c in combos:
        res[c].append(run(c))
for c in combos:
    for (name, _), v in zip(axes, c):
        up.debug_option(name, v)
    up.reset_state()
    outs[c] = up.update()
rel = lambda x, y: np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300)
for c in combos:
    o, r = outs[c], outs[combos[0]]
    print(" ".join(f"{n}={v}" for (n, _), v in zip(axes, c)), f": median {np.median(res[c]):.4f} ms  min {min(res[c]):.4f} | vs first: status same",
          np.array_equal(o["feat_status"], r["feat_status"]), "dx", rel(o["dx"], r["dx"]), "P", rel(o["P"], r["P"]), flush=True)
up.close()
