"""Developer script (GPU): do two half-batches on two contexts (two streams) finish sooner than the whole batch on one?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from open_vins_amd import synth, capi
from open_vins_amd.updater import UpdaterMSCKF
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
F = int(sys.argv[2]) if len(sys.argv) > 2 else None
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
prob = synth.make_problem(cfg, F=F)
opts = capi.default_options(chi2_multipler=1.0, no_timing=1)
def bench(ups, n=20):
    for u in ups: u.update_async()
    for u in ups: u.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for u in ups: u.reset_state()
        for u in ups: u.update_async()
    for u in ups: u.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
one = UpdaterMSCKF(opts); one.set_problem(prob)
print("one context, all features: %.3f ms" % bench([one]))
ups = []
for k in range(parts):
    u = UpdaterMSCKF(opts); u.set_problem(prob.subset(np.arange(k, prob.F, parts))); ups.append(u)
print("%d contexts, 1/%d of the features each, concurrently: %.3f ms" % (parts, parts, bench(ups)))
print("one of them alone: %.3f ms" % bench(ups[:1]))
