"""Developer A/B (GPU, under rocprofv3 --kernel-trace): 12 updates with the legacy per-feature kernels, 12 with the fused one."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
F = int(sys.argv[2]) if len(sys.argv) > 2 else None
prob = synth.make_problem(cfg, F=F)
for legacy in (1, 0):
    up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0))
    up.debug_option("legacy_feature_kernel", legacy)
    up.set_problem(prob)
    for _ in range(3):
        up.reset_state(); up.update_async()
    up.synchronize(); up.kernel_times(reset=True)
    for _ in range(12):
        up.reset_state(); up.update_async()
    up.synchronize()
    kt = up.kernel_times(reset=True)
    print("legacy" if legacy else "fused ", f"update {kt['ms_update'] * 1e3:.0f} us, stage {kt['ms_system'] * 1e3:.0f} us", flush=True)
    up.close()
