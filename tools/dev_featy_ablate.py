"""Developer timing ablation of k_feat_y (GPU): which phase costs what when it is really removed (results are garbage)."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
F = int(sys.argv[2]) if len(sys.argv) > 2 else None
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
prob = synth.make_problem(cfg, F=F)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
up.set_problem(prob)
names = {0: "full", 1: "-sweep", 2: "-out", 4: "-syrk", 8: "-chol", 3: "-sweep-out", 12: "-syrk-chol", 15: "-all", 7: "chol only", 11: "syrk only", 14: "sweep only", 13: "out only"}
for mask in (0, 1, 2, 4, 8, 3, 12, 15, 7, 11, 14, 13):
    up.debug_option("featy_skip", mask)
    for _ in range(3):
        up.reset_state(); up.update_async()
    try:
        up.synchronize()
    except Exception:
        pass
    up.kernel_times(reset=True)
    for _ in range(reps):
        up.reset_state(); up.update_async()
    try:
        up.synchronize()
    except Exception:
        pass
    kt = up.kernel_times(reset=True)
    print(f"{names[mask]:12s} stage {kt['ms_system'] * 1e3:7.1f} us", flush=True)
up.close()
