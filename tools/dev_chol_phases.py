"""Developer probe (GPU): cycle counters of the single-launch Cholesky (k_chol.h, -DOVG_CHOL_PROF build) per update at a bench configuration:
the chain wavefront (total / waiting for the next diagonal tile / factoring), two tile wavefronts (waiting for U_kk^-1, panel solve, counting
barrier, trailing update, stores), the pair owner's hand-over, a follower.  usage: dev_chol_phases.py [--lib path/to/libovgpu_prof.so] [cfg]"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from open_vins_amd import capi, synth
args = sys.argv[1:]
if args and args[0] == "--lib":
    capi.LIB_PATH = os.path.abspath(args[1])
    args = args[2:]
from open_vins_amd.updater import UpdaterMSCKF
cfg = int(args[0]) if args else 3
kw = dict(imu_intrinsics=True) if cfg == 3 else {}
prob = synth.make_problem(cfg, **kw)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
up.set_problem(prob)
lib = up.lib
lib.ovgpu_debug_cycles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
def sync():
    try:
        up.synchronize()
    except capi.OvgpuError as e:  # (a timing-experiment build whose results are garbage reports "not positive definite")
        print("  (", str(e)[:90], ")")
for _ in range(3):
    up.reset_state(); up.update_async()
sync()
lib.ovgpu_debug_cycles(up._ctx, 1, None)
reps = 20
for _ in range(reps):
    up.reset_state(); up.update_async()
sync()
buf = (C.c_longlong * 512)()
lib.ovgpu_debug_cycles(up._ctx, 1, buf)
a = np.array(buf[:], dtype=np.float64)
n = max(a[313], 1.0)  # factorisations counted by the chain wavefront (two per update)
print(f"cfg {cfg}: N = {prob.N}, {int(n)} factorisations in {reps} updates; cycles per factorisation")
print(f"  chain wavefront   total {a[310] / n:9.0f} | waiting for tile (0, 0) {a[311] / n:9.0f} | factoring {a[312] / n:9.0f} | publish + pair (solve, update, panel) {a[314] / n:9.0f}")
for base, name in ((320, "tile wavefront 1"), (330, "tile wavefront 7")):
    m = max(a[base + 6], 1.0)
    print(f"  {name}  total {a[base] / m:9.0f} | wait U_kk^-1 {a[base + 1] / m:9.0f} | panel solve {a[base + 3] / m:9.0f} | counting barrier {a[base + 4] / m:9.0f} | "
          f"trailing {a[base + 5] / m:9.0f} | stores {a[base + 2] / m:9.0f}")
print(f"  follower 0        total {a[303] / n:9.0f} | waiting {a[304] / n:9.0f}")
print(f"  start-up          entry -> first barrier {a[373] / n:9.0f} | wavefront 0: tile indices {a[370] / n:9.0f} | loads issued {a[371] / n:9.0f} | loads waited for + deposit {a[372] / n:9.0f}")
print("  SIMD of wavefronts 0..15 (HW_ID[5:4]):", [int(x) for x in a[350:366]])
kt = up.kernel_times(reset=True)
print("  stage ms:", {k: round(v, 4) for k, v in kt.items()} if isinstance(kt, dict) else kt)
up.close()
