"""Host-buffers-in, host-buffers-out time of one update (the PCIe-inclusive figure DESIGN.md section 6 quotes next to the
resident-input benchmark): ovgpu_set_state + ovgpu_set_features + ovgpu_msckf_update with dx / P' read back, per call."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from open_vins_amd import capi, synth  # noqa: E402
from open_vins_amd.updater import UpdaterMSCKF  # noqa: E402

prob = synth.make_problem(2)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
for _ in range(3):
    up.set_problem(prob)
    up.update()
ts, tu = [], []
for _ in range(20):
    t0 = time.perf_counter()
    up.set_problem(prob)
    t1 = time.perf_counter()
    out = up.update()
    t2 = time.perf_counter()
    ts.append(t1 - t0), tu.append(t2 - t1)
ts.sort(), tu.sort()
print("cfg-2, 800 features: upload (set_state + set_features) median %.3f ms, update incl. read-back of status / chi2 / p_FinG / dx / P' median %.3f ms, device %.3f ms -> %.0f features/s host to host"
      % (1e3 * ts[10], 1e3 * tu[10], out["stats"]["ms_total"], prob.F / (ts[10] + tu[10])))
up.close()
