"""Developer script (GPU): N default updates of a BASELINE config, for rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from open_vins_amd import synth, capi
from open_vins_amd.updater import UpdaterMSCKF
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
F = int(sys.argv[2]) if len(sys.argv) > 2 else None
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
kw = {}
dbg = {}
for a in sys.argv[4:]:
    k, v = a.split("=")
    if k.startswith("dbg:"):
        dbg[k[4:]] = int(v)
    else:
        kw[k] = int(v)
prob = synth.make_problem(cfg, F=F)
up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0, **kw))
for k, v in dbg.items():
    up.debug_option(k, v)
up.set_problem(prob)
try:
    up.update()
except Exception as e:
    print("first update:", e)
import ctypes as C
up.lib.ovgpu_debug_cycles(up._ctx, 1, None)
for _ in range(n):
    up.reset_state()
    up.update_async()
try:
    up.synchronize()
except Exception as e:
    print("synchronize:", e)
print(up.kernel_times())
cyc = (C.c_longlong * 512)()
up.lib.ovgpu_debug_cycles(up._ctx, 0, cyc)
c = list(cyc[300:305])
if c[2]:
    print(f"k_chol_pipe per call: factor WG {c[0] / c[2] / 1e3:.1f} kcyc (diag phase {c[1] / c[2] / 1e3:.1f}), follower WG 1 {c[3] / c[2] / 1e3:.1f} kcyc (waiting {c[4] / c[2] / 1e3:.1f})")
ph = list(cyc[200:211])
if sum(ph):
    tot = sum(ph)
    names = {0: "copy", 3: "T sweep", 4: "S0 tiles", 5: "diag factor", 6: "row panel", 7: "trailing", 8: "chi2", 10: "tail"}
    print("k_feat workgroup 0 phases (% of its cycles):", ", ".join(f"{names.get(i, i)} {100 * v / tot:.1f}" for i, v in enumerate(ph) if v), f"| total {tot / (n + 1) / 1e3:.0f} kcyc per update")
ph = list(cyc[220:230])
if sum(ph):
    tot = sum(ph)
    names = {0: "prologue", 1: "sweep", 2: "out", 3: "SYRK", 4: "finalize", 5: "cholesky+chi2", 6: "diag", 7: "panel", 8: "trailing"}
    print("k_feat_y workgroup 0 phases (% of its cycles):", ", ".join(f"{names.get(i, i)} {100 * v / tot:.1f}" for i, v in enumerate(ph) if v), f"| total {tot / (n + 1) / 1e3:.0f} kcyc per update")
up.close()
