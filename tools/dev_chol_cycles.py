"""Developer aid (GPU): in-kernel cycle counters of k_chol_factor2 (chain wavefront; tile wavefronts 1 and 7), both factorisations of an update summed."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
prob = synth.make_problem(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
up.set_problem(prob)
for _ in range(3):
    up.reset_state(); up.update_async()
up.synchronize()
up.lib.ovgpu_debug_cycles(up._ctx, 1, None)
n = 20
for _ in range(n):
    up.reset_state(); up.update_async()
up.synchronize()
cyc = (C.c_longlong * 512)()
up.lib.ovgpu_debug_cycles(up._ctx, 0, cyc)
k = max(cyc[313], 1)
print(f"chain (per factorisation, {cyc[313]} runs): total {cyc[310] / k:.0f}  wait for the diagonal tile {cyc[311] / k:.0f}  factor {cyc[312] / k:.0f}")
for name, o in (("tile wavefront 1", 320), ("tile wavefront 7", 330)):
    k = max(cyc[o + 6], 1)
    print(f"{name}: total {cyc[o] / k:.0f}  wait U^-1 {cyc[o + 1] / k:.0f}  stores {cyc[o + 2] / k:.0f}  panel {cyc[o + 3] / k:.0f}  counting barrier {cyc[o + 4] / k:.0f}  trailing {cyc[o + 5] / k:.0f}")

k = max(cyc[342], 1)
print(f"pair owner, per step ({cyc[342]} steps): U^-1 seen -> next diagonal tile handed over {cyc[340] / k:.0f} cycles; waited for U^-1 {cyc[341] / k:.0f}")
up.close()
