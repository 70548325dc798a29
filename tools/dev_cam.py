import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, ctypes as C
from open_vins_amd import synth, capi
from open_vins_amd.updater import UpdaterMSCKF
from oracle import pyoracle
up = UpdaterMSCKF(capi.default_options())
lib = pyoracle.load(); dp = capi.c_double_p
rng = np.random.default_rng(0)
for fish in (0, 1):
    cam = np.array((synth._INTRINSICS_EQUI if fish else synth._INTRINSICS)[0])
    n = 20000
    uvn = rng.uniform(-0.6, 0.6, (n, 2))
    uv = np.zeros((n, 2)); a = np.zeros((n, 4)); b = np.zeros((n, 16))
    capi.check(up.lib.ovgpu_cam_distort(up._ctx, fish, cam.ctypes.data_as(dp), n, uvn.ctypes.data_as(dp), uv.ctypes.data_as(dp), a.ctypes.data_as(dp), b.ctypes.data_as(dp)), "cam")
    uv2 = np.zeros((n, 2)); a2 = np.zeros((n, 4)); b2 = np.zeros((n, 16))
    for i in range(n):
        lib.oracle_cam_distort(cam.ctypes.data_as(dp), fish, uvn[i].ctypes.data_as(dp), uv2[i].ctypes.data_as(dp), a2[i].ctypes.data_as(dp), b2[i].ctypes.data_as(dp))
    print("fish", fish, "uv mismatches", (uv != uv2).sum(), "of", uv.size, "max", np.abs(uv - uv2).max(), "dzn rel", np.abs(a - a2).max() / np.abs(a2).max(), "dze", np.abs(b - b2).max())
