"""bench.py --gpus N must start its own N ranks when no launcher did (CPU dry run of the launcher: the ranks themselves need GPUs)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_flag_spawns_ranks(monkeypatch):
    import bench
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert e.value.code == 0
    cmd = calls["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and os.path.basename(cmd[-7]) == "bench.py"
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_single_gpu_line_needs_a_gpu(monkeypatch):
    """Without a GPU the bench refuses to run (no CPU fallback) instead of printing a number."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--steps", "1", "--warmup", "0"])
    assert "no CPU fallback" in str(e.value)


def test_gated_flops_counts_only_features_that_reach_the_gate():
    import numpy as np
    import bench
    from open_vins_amd import capi, synth
    prob = synth.make_problem(2, F=20)
    st = np.full(prob.F, capi.FEAT_USED, dtype=np.int32)
    all_s, all_c = bench.gated_flops(prob, st, capi, synth)
    st[:5] = capi.FEAT_TRI_FAILED
    st[5:8] = capi.FEAT_CHI2_REJECTED
    s2, c2 = bench.gated_flops(prob, st, capi, synth)
    one = [bench.gated_flops(prob.subset([f]), np.array([capi.FEAT_USED], dtype=np.int32), capi, synth) for f in range(prob.F)]
    assert abs(s2 - sum(o[0] for o in one[5:])) < 1e-6 * all_s       # failed triangulations never enter the kernel
    assert abs(c2 - sum(o[1] for o in one[8:])) < 1e-6 * all_c       # rejected features are gated but not stacked
    assert abs(all_s - synth.algorithmic_flops_system(prob)) < 1e-9 * all_s
