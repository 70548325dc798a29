#!/bin/bash
# fp32 variant check: parity tests, then the one-GPU share of configs[4] in f64 and with --gram-fp32, and configs[2] with --gram-fp32
set -u
TAG=${1:-fp32}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_ref_fixtures.py -q -m gpu -s -p no:cacheprovider -k "fp32_gram_variant or cfg5_geometry or ref" 2>&1 | grep "fp32 Gram\|passed\|failed\|Error\|error" | tail -15 > $OUT/pytest.txt
B="python bench.py --no-cpu-baseline --no-extras"
timeout 200 $B --cfg 5 --features 2500 --steps 5 --warmup 1 > $OUT/bench_cfg5_f64.json 2> $OUT/bench.err
timeout 200 $B --cfg 5 --features 2500 --gram-fp32 --steps 5 --warmup 1 > $OUT/bench_cfg5_fp32.json 2>> $OUT/bench.err
timeout 120 $B --steps 50 --warmup 5 > $OUT/bench_cfg3.json 2>> $OUT/bench.err
timeout 120 $B --gram-fp32 --steps 50 --warmup 5 > $OUT/bench_cfg3_fp32.json 2>> $OUT/bench.err
cat $OUT/pytest.txt
for f in bench_cfg5_f64 bench_cfg5_fp32 bench_cfg3 bench_cfg3_fp32; do python - $OUT/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["roofline"]["compression"]
    print(sys.argv[1].split("/")[-1], "ms/step %.3f"%d["ms_per_step"], "system ms %.3f frac %.3f"%(d["roofline"]["avg_ms_per_launch"], d["roofline"]["frac"]), "compress ms %.3f frac %.3f peak %s"%(c["avg_ms_per_launch"], c["frac"], c.get("peak")), "update %.3f"%d["roofline"]["update_ms_device"], "whole frac %.3f"%d["roofline"]["whole_update"]["frac"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -3 $OUT/bench.err
