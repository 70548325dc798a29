"""Developer probe (GPU): where the host-to-host time of one frame goes (bench.py's pcie_inclusive_ms): ovgpu_set_state + ovgpu_set_features +
ovgpu_msckf_update on pageable host memory.  (a) the host-side return time of each call in the pipelined sequence, (b) each call followed by a
synchronisation (its isolated cost), (c) the asynchronous update alone (enqueue time / device time)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from open_vins_amd import capi, synth  # noqa: E402
from open_vins_amd.updater import UpdaterMSCKF  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
prob = synth.make_problem(cfg, imu_intrinsics=(cfg == 3))
opts = capi.default_options(chi2_multipler=1.0)
up = UpdaterMSCKF(opts)
up.debug_option("stage_timing_period", 1000000)
v = capi.Views(prob)
F, N = v.features.F, v.state.N
st, chi2, thr = np.zeros(F, np.int32), np.zeros(F), np.zeros(F)
pG, dx, P = np.zeros((F, 3)), np.zeros(N), np.zeros((N, N))
stats = capi.UpdateStats()
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
args = (st.ctypes.data_as(C.POINTER(C.c_int32)), dp(chi2), dp(thr), dp(pG), dp(dx), dp(P), C.byref(stats))
lib, ctx = up.lib, up._ctx
pc = time.perf_counter


def med(xs):
    return 1e3 * sorted(xs)[len(xs) // 2]


for upk in (1, 0, 1, 0):
  up.debug_option("upload_kernel", upk)
  rows = {k: [] for k in ("state", "feats", "update", "total")}
  for it in range(45):
    t0 = pc(); lib.ovgpu_set_state(ctx, C.byref(v.state))
    t1 = pc(); lib.ovgpu_set_features(ctx, C.byref(v.features))
    t2 = pc(); rc = lib.ovgpu_msckf_update(ctx, *args)
    t3 = pc()
    assert rc == 0
    if it >= 5:
        rows["state"].append(t1 - t0), rows["feats"].append(t2 - t1), rows["update"].append(t3 - t2), rows["total"].append(t3 - t0)
  print(f"upload_kernel={upk} (a) pipelined, host return times [ms]: " + "  ".join(f"{k} {med(x):.4f}" for k, x in rows.items()))
up.debug_option("upload_kernel", 1)
rows = {k: [] for k in ("state", "feats", "update", "total")}
for it in range(25):
    t0 = pc(); lib.ovgpu_set_state(ctx, C.byref(v.state))
    t1 = pc(); lib.ovgpu_set_features(ctx, C.byref(v.features))
    t2 = pc(); rc = lib.ovgpu_msckf_update(ctx, *args)
    t3 = pc()
    assert rc == 0
    if it >= 5:
        rows["state"].append(t1 - t0), rows["feats"].append(t2 - t1), rows["update"].append(t3 - t2), rows["total"].append(t3 - t0)
print("(a) pipelined, host return times [ms]: " + "  ".join(f"{k} {med(x):.4f}" for k, x in rows.items()))

rows = {k: [] for k in ("state", "feats", "update")}
for it in range(25):
    lib.ovgpu_synchronize(ctx)
    t0 = pc(); lib.ovgpu_set_state(ctx, C.byref(v.state)); ta = pc(); lib.ovgpu_synchronize(ctx)
    t1 = pc(); lib.ovgpu_set_features(ctx, C.byref(v.features)); tb = pc(); lib.ovgpu_synchronize(ctx)
    t2 = pc(); lib.ovgpu_msckf_update(ctx, *args)
    t3 = pc()
    if it >= 5:
        rows["state"].append((ta - t0, t1 - t0)), rows["feats"].append((tb - t1, t2 - t1)), rows["update"].append((t3 - t2, t3 - t2))
print("(b) isolated [ms] (host return / with sync): " + "  ".join(f"{k} {med([a for a, _ in x]):.4f} / {med([b for _, b in x]):.4f}" for k, x in rows.items()))

enq, tot = [], []
for it in range(25):
    lib.ovgpu_reset_state(ctx)
    lib.ovgpu_synchronize(ctx)
    t0 = pc(); lib.ovgpu_msckf_update_async(ctx); t1 = pc(); lib.ovgpu_synchronize(ctx); t2 = pc()
    if it >= 5:
        enq.append(t1 - t0), tot.append(t2 - t0)
print(f"(c) asynchronous update alone [ms]: enqueue {med(enq):.4f}  enqueue + device {med(tot):.4f}")
# results without P (a resident-covariance caller): dx only
args2 = (st.ctypes.data_as(C.POINTER(C.c_int32)), dp(chi2), dp(thr), dp(pG), dp(dx), None, C.byref(stats))
ts = []
for it in range(25):
    lib.ovgpu_reset_state(ctx)
    lib.ovgpu_synchronize(ctx)
    t0 = pc(); lib.ovgpu_msckf_update(ctx, *args2); t1 = pc()
    if it >= 5:
        ts.append(t1 - t0)
print(f"(d) synchronous update alone, dx without P' [ms]: {med(ts):.4f}")
ts = []
for it in range(25):
    lib.ovgpu_reset_state(ctx)
    lib.ovgpu_synchronize(ctx)
    t0 = pc(); lib.ovgpu_msckf_update(ctx, *args); t1 = pc()
    if it >= 5:
        ts.append(t1 - t0)
print(f"(e) synchronous update alone, every output [ms]: {med(ts):.4f}")
up.close()
