#!/bin/bash
# the gate's residual bound: parity (both modes), then the bench lines with and without it at configs[2], [3] on one GPU, one rank's share of [4]
set -u
TAG=${1:-gate}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_known_answer.py tests/test_gpu_fullsize.py tests/test_ref_fixtures.py -q -m gpu -x -p no:cacheprovider -k "not 10k_features and not fp32_gram_variant and not conditioning and not sharded" 2>&1 | tail -12 > $OUT/pytest.txt
B="python bench.py --no-cpu-baseline --no-extras"
timeout 200 $B --steps 200 --warmup 10 > $OUT/bench_cfg3.json 2> $OUT/bench.err
timeout 200 $B --steps 200 --warmup 10 --gate-always-factor > $OUT/bench_cfg3_full.json 2>> $OUT/bench.err
timeout 200 $B --cfg 2 --steps 200 --warmup 10 > $OUT/bench_cfg2.json 2>> $OUT/bench.err
timeout 200 $B --cfg 2 --steps 200 --warmup 10 --gate-always-factor > $OUT/bench_cfg2_full.json 2>> $OUT/bench.err
timeout 200 $B --cfg 4 --steps 20 --warmup 3 > $OUT/bench_cfg4.json 2>> $OUT/bench.err
timeout 200 $B --cfg 4 --steps 20 --warmup 3 --gate-always-factor > $OUT/bench_cfg4_full.json 2>> $OUT/bench.err
timeout 200 $B --cfg 5 --features 2500 --steps 10 --warmup 2 > $OUT/bench_cfg5.json 2>> $OUT/bench.err
timeout 200 $B --cfg 5 --features 2500 --steps 10 --warmup 2 --gate-always-factor > $OUT/bench_cfg5_full.json 2>> $OUT/bench.err
timeout 200 $B --cfg 5 --features 2500 --steps 10 --warmup 2 --gram-fp32 > $OUT/bench_cfg5_fp32.json 2>> $OUT/bench.err
cat $OUT/pytest.txt
for f in bench_cfg3 bench_cfg3_full bench_cfg2 bench_cfg2_full bench_cfg4 bench_cfg4_full bench_cfg5 bench_cfg5_full bench_cfg5_fp32; do python - $OUT/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["roofline"]["compression"]; g=d["roofline"]["gate"]
    print(sys.argv[1].split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "system %.4f frac %.3f"%(d["roofline"]["avg_ms_per_launch"], d["roofline"]["frac"]), "compress %.4f"%c["avg_ms_per_launch"], "gate", g["features_passed_by_the_bound"], "/", g["features_reaching_the_gate"], "used", d["config"]["features_used_rank0"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -3 $OUT/bench.err
