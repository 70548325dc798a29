#!/usr/bin/env python3
"""Developer script: one update on the GPU next to the oracle, prints the parity deltas."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from open_vins_amd import synth, capi
from open_vins_amd.updater import UpdaterMSCKF
from oracle import pyoracle

def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    track = sys.argv[3] if len(sys.argv) > 3 else "full"
    rep = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    prob = synth.make_problem(cfg, F=F, track=track)
    opts = capi.default_options(chi2_multipler=1.0, feat_rep_msckf=rep)
    v = capi.Views(prob)
    t = time.time(); ref = pyoracle.msckf_update(opts, v, want_compressed=True); t_cpu = time.time() - t
    up = UpdaterMSCKF(opts)
    up.set_problem(prob)
    tri = up.triangulate()
    rt = pyoracle.triangulate(opts, v)
    ok = (rt["status"] == 0)
    print("tri status equal:", np.array_equal(tri["status"], rt["status"]), "n_ok", ok.sum(),
          "max |dpG|", np.abs(tri["p_FinG"][ok] - rt["p_FinG"][ok]).max() if ok.any() else None)
    out = up.update(check=False)
    print("rc", out["rc"], out["stats"])
    print("oracle stats", ref["stats"], "cpu s", t_cpu, ref["stage_seconds"])
    print("status equal:", np.array_equal(out["feat_status"], ref["feat_status"]), np.bincount(out["feat_status"], minlength=5), np.bincount(ref["feat_status"], minlength=5))
    g = np.isfinite(ref["chi2"])
    print("chi2 rel err max", np.nanmax(np.abs(out["chi2"][g] - ref["chi2"][g]) / ref["chi2"][g]))
    print("dx rel", np.linalg.norm(out["dx"] - ref["dx"]) / np.linalg.norm(ref["dx"]))
    print("P rel fro", np.linalg.norm(out["P"] - ref["P"]) / np.linalg.norm(ref["P"]))
    print("clone max", np.abs(out["clone_q_p"] - ref["clone_q_p"]).max(), "calib", np.abs(out["calib_q_p"] - ref["calib_q_p"]).max(), "intr", np.abs(out["intrinsics"] - ref["intrinsics"]).max())
    up.reset_state()
    cmp = up.compress()
    H, r = cmp["H"], cmp["r"]
    Hr, rr = ref["H_comp"], ref["r_comp"]
    print("R^T R rel", np.linalg.norm(H.T @ H - Hr.T @ Hr) / np.linalg.norm(Hr.T @ Hr), "R^T c rel", np.linalg.norm(H.T @ r - Hr.T @ rr) / np.linalg.norm(Hr.T @ rr))
    # timing
    for i in range(3):
        up.reset_state(); up.update_async()
    up.synchronize(); up.kernel_times(reset=True)
    n = 10
    t = time.time()
    for i in range(n):
        up.reset_state(); up.update_async()
    up.synchronize(); wall = (time.time() - t) / n
    print("wall ms/update", wall * 1e3, up.kernel_times())

if __name__ == "__main__":
    main()
