"""Developer probe (GPU): cycle counters of workgroup 0 of k_feat_y, phase by phase (FEAT_T slots 220..225), per update."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
prob = synth.make_problem(cfg)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
if len(sys.argv) > 2:
    up.debug_option("raw_stack", int(sys.argv[2]))  # 0: projected stack, 2: one region (developer experiments)
up.set_problem(prob)
lib = up.lib
lib.ovgpu_debug_cycles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for _ in range(3):
    up.reset_state(); up.update_async()
up.synchronize()
lib.ovgpu_debug_cycles(up._ctx, 1, None)
reps = 20
for _ in range(reps):
    up.reset_state(); up.update_async()
up.synchronize()
buf = (C.c_longlong * 512)()
lib.ovgpu_debug_cycles(up._ctx, 1, buf)
a = np.array(buf[220:227], dtype=np.float64) / reps
names = ["prologue", "sweep", "store (V^T Y is the last line)", "SYRK", "S0 setup", "Cholesky+verdict", "V^T Y partial sums + barrier"]
tot = a.sum()
kt = up.kernel_times(reset=True)
print("workgroup 0 of k_feat_y, cycles per update (100 MHz clock64 ticks?):")
for n, v in zip(names, a):
    print(f"  {n:20s} {v:10.0f}  {100 * v / tot:5.1f} %")
print("  total", tot, " stage ms", kt["ms_system"])
up.close()
