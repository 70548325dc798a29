"""Developer check (GPU): k_gram_wide (two passes over the stack at 23 tile columns) against the block variant k_gram_blk — results and time."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from open_vins_amd import capi, synth
from open_vins_amd.updater import UpdaterMSCKF

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
prob = synth.make_problem(5, F=F)
opts = capi.default_options(chi2_multipler=1.0)
outs = {}
for blocks in (1, 0):
    up = UpdaterMSCKF(opts)
    up.debug_option("gram_blocks_only", blocks)
    up.set_problem(prob)
    outs[blocks] = up.update()
    for _ in range(3):
        up.reset_state(); up.update_async()
    up.synchronize()
    up.kernel_times(reset=True)
    t = time.perf_counter()
    for _ in range(10):
        up.reset_state(); up.update_async()
    up.synchronize()
    dt = (time.perf_counter() - t) / 10 * 1e3
    kt = up.kernel_times(reset=True)
    print(f"gram_blocks_only {blocks}: {dt:.3f} ms / update, compression {kt['ms_compress']:.3f} ms, system {kt['ms_system']:.3f} ms", flush=True)
    up.close()
a, b = outs[1], outs[0]
rel = lambda x, y: np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300)
print("wide vs blocks: status same", np.array_equal(a["feat_status"], b["feat_status"]), "dx", rel(b["dx"], a["dx"]), "P", rel(b["P"], a["P"]))
