"""Developer script (GPU): single-launch Cholesky-with-carry (k_chol.h) against the step-wise kernels, through complete updates."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from open_vins_amd import synth, capi
from open_vins_amd.updater import UpdaterMSCKF

def run(prob, pipe, steps=20, **kw):
    up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0, no_single_launch_cholesky=0 if pipe else 1, **kw))
    up.set_problem(prob)
    out = up.update()
    up.kernel_times(reset=True)
    for _ in range(steps):
        up.reset_state()
        up.update_async()
    up.synchronize()
    out["kt"] = up.kernel_times(reset=True)
    up.close()
    return out

rel = lambda x, y: np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300)
for name, kw in (("cfg2 F=800", dict(cfg=2, F=800)), ("K1 C12 (D=86)", dict(cfg=2, F=100, K=1, C=12)), ("cfg4 F=300 (D=236)", dict(cfg=4, F=300)),
                 ("C=5 K=1 (D=44)", dict(cfg=2, F=30, K=1, C=5)), ("cfg2 F=800 tsqr", dict(cfg=2, F=800, route=1))):
    kw = dict(kw)
    route = kw.pop("route", 0)
    prob = synth.make_problem(kw.pop("cfg"), **kw)
    a = run(prob, False, compress_route=route)
    b = run(prob, True, compress_route=route)
    print(f"{name}: status same {np.array_equal(a['feat_status'], b['feat_status'])} rc {b['stats']['status']}, dx {rel(b['dx'], a['dx']):.2e}, P {rel(b['P'], a['P']):.2e}, "
          f"sym {np.array_equal(b['P'], b['P'].T)}; update ms step-wise {a['kt']['ms_update']:.3f} -> single launch {b['kt']['ms_update']:.3f}")
