"""Developer probe (GPU): N whole frames host to host (ovgpu_set_state + ovgpu_set_features + synchronous ovgpu_msckf_update) for a kernel trace
(tools/gpu_frame_timeline.sh)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from open_vins_amd import capi, synth  # noqa: E402
from open_vins_amd.updater import UpdaterMSCKF  # noqa: E402

prob = synth.make_problem(3, imu_intrinsics=True)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
up.debug_option("stage_timing_period", 1000000)
v = capi.Views(prob)
F, N = v.features.F, v.state.N
st, chi2, thr = np.zeros(F, np.int32), np.zeros(F), np.zeros(F)
pG, dx, P = np.zeros((F, 3)), np.zeros(N), np.zeros((N, N))
stats = capi.UpdateStats()
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
args = (st.ctypes.data_as(C.POINTER(C.c_int32)), dp(chi2), dp(thr), dp(pG), dp(dx), dp(P), C.byref(stats))
ts = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    t = time.perf_counter()
    up.lib.ovgpu_set_state(up._ctx, C.byref(v.state))
    up.lib.ovgpu_set_features(up._ctx, C.byref(v.features))
    assert up.lib.ovgpu_msckf_update(up._ctx, *args) == 0
    ts.append(time.perf_counter() - t)
print("frame host to host, median [ms]:", 1e3 * sorted(ts)[len(ts) // 2])
up.close()
