#!/usr/bin/env python3
"""Developer script: per-phase cycle counts of the TSQR node kernel (needs tools/_prof/libovgpu_prof.so, built with -DQR_PROFILE)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from open_vins_amd import capi
capi.LIB_PATH = os.path.join(os.path.dirname(__file__), "_prof", "libovgpu_prof.so")
import numpy as np
from open_vins_amd import synth
from open_vins_amd.updater import UpdaterMSCKF

prob = synth.make_problem(2, F=int(sys.argv[1]) if len(sys.argv) > 1 else 800)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
up.set_problem(prob)
for _ in range(3):
    up.reset_state(); up.update_async()
up.synchronize()
print(up.kernel_times(reset=True))
lib = capi.load()
buf = (ctypes.c_longlong * 128)()
lib.ovgpu_debug_qr_cycles.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
lib.ovgpu_debug_qr_cycles(buf)
a = np.array(buf[:]).reshape(2, 16, 4)
for name, blk in (("leaf (dense, node 0)", a[0]), ("last merge (node 0)", a[1])):
    print(name, "cycles per wave (last wave = panel wave): [loop top, barrier wait, apply, publish]")
    for w in range(8):
        print("  wave", w, blk[w], "sum", blk[w].sum())
s = np.array(buf[100:110])
names = ["loop top/wait", "(a0) representation", "(a) rows", "(c) T=HP + S0 chunks", "RHS", "Cholesky", "chi2 reduce", "(b) Householder H_f", "(d) stack/output", "(c) phase 1 only (T = H P)"]
print("k_system block 0 cycles by phase (sum over its features):")
for nme, v in zip(names, s):
    print("  %-24s %10d  %5.1f%%" % (nme, v, 100.0 * v / max(s.sum(), 1)))
