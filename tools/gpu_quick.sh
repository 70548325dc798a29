#!/bin/bash
# Quick GPU check between edits: the parity suite without its five slowest cases, then the bench lines at 2000 / 800 features.
set -u
TAG=${1:-quick}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "not cfg5_geometry and not headline_closed_loop and not 10k_features and not fp32_gram_variant and not cfg4_shard and not long_loop" 2>&1 | tail -8 > $OUT/pytest_gpu.txt
B="python bench.py --no-cpu-baseline --no-extras"
timeout 120 $B --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout 120 $B --cfg 2 --steps 50 --warmup 5 > $OUT/bench_cfg2.json 2>> $OUT/bench.err
cat $OUT/pytest_gpu.txt; cut -c1-330 $OUT/bench.json; echo; cut -c1-330 $OUT/bench_cfg2.json; echo; tail -3 $OUT/bench.err
