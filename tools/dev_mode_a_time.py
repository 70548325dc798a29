"""Developer timing (GPU): mode A host to host — ovgpu_msckf_compress (upload excluded: the problem is resident; the call returns
(H, r) on the host) per route, and the device-side stage times the library reports."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
F = int(sys.argv[2]) if len(sys.argv) > 2 else None
prob = synth.make_problem(cfg, F=F) if F else synth.make_problem(cfg)
res = {}
for name, code in (("tsqr (Householder)", capi.COMPRESS_TSQR), ("pcholqr (pivoted)", capi.COMPRESS_PCHOLQR),
                   ("default", capi.COMPRESS_GRAM)):
    up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0, compress_route=code))
    up.set_problem(prob)
    for _ in range(3):
        out = up.compress()
    ts = []
    for _ in range(15):
        t = time.perf_counter()
        out = up.compress()
        ts.append((time.perf_counter() - t) * 1e3)
    st = out["stats"]
    res[name] = out
    print(f"cfg {cfg} F {prob.F} | {name:20s} route {up.lib.ovgpu_last_update_route(up._ctx)}  host to host median {np.median(ts):.3f} ms  min {min(ts):.3f}   stats {({k: round(v, 4) for k, v in st.items() if k.startswith('ms')})}", flush=True)
    up.close()
a, b = res["tsqr (Householder)"], res["pcholqr (pivoted)"]
Ga, Gb = a["H"].T @ a["H"], b["H"].T @ b["H"]
print("pivoted vs Householder: |G_p - G_h| / |G_h| =", np.linalg.norm(Gb - Ga) / np.linalg.norm(Ga), " zero rows", int((np.abs(b["H"]).sum(axis=1) == 0).sum()))
