"""Developer script (GPU): the MSCKF fast path of the per-feature stage (k_feat.h) against the general kernel (k_system.h)."""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from open_vins_amd import synth, capi
from open_vins_amd.updater import UpdaterMSCKF

def run(prob, fast, steps=10, **kw):
    up = UpdaterMSCKF(capi.default_options(chi2_multipler=1.0, no_fast_feature_kernel={False: 1, True: 0, 2: 2}[fast], **kw))
    up.set_problem(prob)
    out = up.update()
    up.lib.ovgpu_debug_cycles(up._ctx, 1, None)
    up.kernel_times(reset=True)
    for _ in range(steps):
        up.reset_state()
        up.update_async()
    up.synchronize()
    ms = C.c_double(0); n = C.c_int64(0)
    up.lib.ovgpu_system_time(up._ctx, C.byref(ms), C.byref(n))
    kt = up.kernel_times(reset=True)
    cyc = (C.c_longlong * 512)()
    up.lib.ovgpu_debug_cycles(up._ctx, 0, cyc)
    out["ms_system"] = ms.value
    out["kt"] = kt
    out["cyc"] = np.array(cyc[200:212], dtype=np.float64)
    up.close()
    return out

for name, kw in (("cfg2 F=64", dict(cfg=2, F=64)), ("cfg2 F=800", dict(cfg=2, F=800)), ("ragged F=300", dict(cfg=2, F=300, track="ragged")),
                 ("K1 C12", dict(cfg=2, F=100, K=1, C=12)), ("cfg4 F=300", dict(cfg=4, F=300)), ("outliers", dict(cfg=2, F=200, outlier_frac=0.3)),
                 ("cfg2 10k", dict(cfg=2, F=10000))):
    kw = dict(kw)
    prob = synth.make_problem(kw.pop("cfg"), **kw)
    a = run(prob, False)
    b = run(prob, True)
    same = np.array_equal(a["feat_status"], b["feat_status"])
    gate = np.isfinite(a["chi2"])
    dchi = np.nanmax(np.abs(a["chi2"][gate] - b["chi2"][gate]) / np.abs(a["chi2"][gate])) if gate.any() else 0.0
    rel = lambda x, y: np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300)
    print(f"{name}: status same {same} (used {a['stats']['n_used']}/{b['stats']['n_used']}), chi2 rel {dchi:.2e}, dx {rel(b['dx'], a['dx']):.2e}, P {rel(b['P'], a['P']):.2e}; "
          f"system ms general {a['ms_system']:.3f} fast {b['ms_system']:.3f}; update ms {a['kt']} -> {b['kt']}")
    cyc = b["cyc"]
    tot = cyc.sum()
    if tot > 0:
        names = ["rows", "qr", "-", "Tsweep", "S0", "diag", "trsm", "trail", "chi2", "wvz", "Ysweep", "-"]
        print("   wg0 cycles: " + ", ".join(f"{n} {100 * v / tot:.0f}%" for n, v in zip(names, cyc) if v > 0) + f"  total {tot / 1e3:.0f} kcyc")
    if not same:
        d = np.nonzero(a["feat_status"] != b["feat_status"])[0][:10]
        print("   differing:", d, a["feat_status"][d], b["feat_status"][d], a["chi2"][d], b["chi2"][d])
