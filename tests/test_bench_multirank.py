"""bench.py --gpus 2 END TO END on a machine without GPUs: the launcher spawns two ranks with torch.distributed.run, the ranks
rendezvous over gloo, fail to set up the library's RCCL communicator, agree on the host-driven fall-back (open_vins_amd.parallel: one
all-reduce of the Gram matrix), run the timed loops with their fences and rank 0 prints the JSON line -- with the exchange kind, the
scaling model's prediction for this run and every rank's own time, i.e. what the first real multi-GPU line will be read against.  The
arithmetic is the oracle's through the OVGPU_BENCH_TEST_HOOK seam (tests/bench_cpu_hook.py): this checks control flow, not speed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_runs_end_to_end_over_gloo():
    env = dict(os.environ)
    env["OVGPU_BENCH_TEST_HOOK"] = "tests.bench_cpu_hook"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cfg", "4", "--features", "14"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert "test_hook" in d and "NOT a measurement" in d["test_hook"]
    assert "host-driven fallback" in d["exchange"]["kind"]          # both ranks agreed after the native communicator failed
    assert [r["rank"] for r in d["exchange"]["ranks"]] == [0, 1] and [r["features_this_rank"] for r in d["exchange"]["ranks"]] == [7, 7]
    assert d["exchange"]["rccl_ranks"] == [-1]                       # no RCCL communicator on this host: the line says so
    assert "native RCCL exchange unavailable" in p.stderr
    assert len(d["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_ms_per_step"])
    assert max(d["per_rank_ms_per_step"]) <= d["ms_per_step_timed_loops"][-1] * 1.0001  # the line's time is the max over ranks
    assert d["predicted_ms"] > 0 and "weak" in d and d["weak"]["features_total"] == 12
    # the preflight: on stderr before anything is timed, and in the line
    pre = [l for l in p.stderr.splitlines() if l.startswith("[bench preflight] ")]
    assert len(pre) == 1
    pf, _ = json.JSONDecoder().raw_decode(pre[0][len("[bench preflight] "):])  # (the ranks share the pipe: another rank's line may follow on the same one)
    assert [r["rank"] for r in pf["ranks"]] == [0, 1] and all(r["local_ms_per_step"] > 0 and r["features_this_rank"] == 7 for r in pf["ranks"])
    assert pf["predicted_ms"] > 0 and d["preflight"]["predicted_ms"] == pf["predicted_ms"]
    assert d["config"]["features_total"] == 14 and d["config"]["features_this_rank"] == 7
    assert "feature-shard x2" in d["config"]["parallelism"]
    assert d["vs_baseline"] is None and d["metric"].startswith("MSCKF features/sec")
