"""Developer probe (GPU): does a context's update time depend on how many contexts / streams the process already has?  (bench.py's extras run in
second and third contexts of the process.)"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF

prob = synth.make_problem(3, imu_intrinsics=True)
opts = capi.default_options(chi2_multipler=1.0)


def timed(up, steps=300):
    up.debug_option("stage_timing_period", 1000000)
    for _ in range(10):
        up.reset_state(); up.update_async()
    up.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        up.reset_state(); up.update_async()
    up.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


ups = []
for i in range(4):
    up = UpdaterMSCKF(opts)
    up.set_problem(prob)
    ups.append(up)
    print(f"context {i} (with {i} older contexts alive): {timed(up):.4f} ms/update")
print("again, oldest first:", " ".join(f"{timed(u):.4f}" for u in ups))
for u in ups[:-1]:
    u.close()
print(f"the youngest alone: {timed(ups[-1]):.4f}")
ups[-1].close()
up = UpdaterMSCKF(opts)
up.set_problem(prob)
print(f"a fresh one after all were closed: {timed(up):.4f}")
up.close()
