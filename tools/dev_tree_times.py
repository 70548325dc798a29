#!/usr/bin/env python3
"""Developer script: when did the merge-tree nodes start / get their first inputs (needs tools/_prof/libovgpu_prof.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from open_vins_amd import capi
capi.LIB_PATH = os.path.join(os.path.dirname(__file__), "_prof", "libovgpu_prof.so")
import numpy as np
from open_vins_amd import synth
from open_vins_amd.updater import UpdaterMSCKF
prob = synth.make_problem(2)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
up.set_problem(prob)
for _ in range(3):
    up.reset_state(); up.update_async()
up.synchronize()
lib = capi.load()
buf = (ctypes.c_longlong * 1024)()
lib.ovgpu_debug_tree_times.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
lib.ovgpu_debug_tree_times(buf)
a = np.array(buf[:1024]).reshape(512, 2).astype(np.float64) / 100.0  # us
n = int((a[:, 0] > 0).sum())
print("nodes:", n)
a = a[:n]
t0 = a[:, 0].min()
st, rd = a[:, 0] - t0, a[:, 1] - t0
print("node start  (us after the first): min %.0f  median %.0f  p90 %.0f  max %.0f" % (st.min(), np.median(st), np.percentile(st, 90), st.max()))
print("level-1 nodes (0..121) first inputs: min %.0f median %.0f max %.0f" % (rd[:n // 2].min(), np.median(rd[:n // 2]), rd[:n // 2].max()))
print("nodes started within 50 us:", int((st < 50).sum()), "of", n)
print("start histogram (100 us bins):", np.histogram(st, bins=np.arange(0, 1600, 100))[0])
print("first-input histogram:", np.histogram(rd[rd > 0], bins=np.arange(0, 1600, 100))[0])
