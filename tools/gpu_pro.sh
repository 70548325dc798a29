#!/bin/bash
# after a change of the per-feature kernels: parity subset, then a same-box A/B of the bench line against ab_old/*.so
set -u
TAG=${1:-pro}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_known_answer.py tests/test_gpu_fullsize.py tests/test_ref_fixtures.py -q -m gpu -x -p no:cacheprovider -k "not 10k_features and not conditioning and not sharded" 2>&1 | tail -12 > $OUT/pytest.txt
cat $OUT/pytest.txt
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/cur.so
B="python bench.py --no-cpu-baseline --no-extras"
run() { # name, args
  timeout 300 $B $2 > $OUT/$1.json 2>> $OUT/err
  python - $OUT/$1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["roofline"]["compression"]
    print(sys.argv[1].split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "system %.4f frac %.3f"%(d["roofline"]["avg_ms_per_launch"], d["roofline"]["frac"]), "compress %.4f"%c["avg_ms_per_launch"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for rep in 1 2; do
  cp /tmp/cur.so $CUR; run new_cfg3_$rep "--steps 300 --warmup 10"
  for o in ab_old/*.so; do cp $o $CUR; run old_cfg3_$rep "--steps 300 --warmup 10"; done
done
cp /tmp/cur.so $CUR; run new_cfg2 "--cfg 2 --steps 300 --warmup 10"
for o in ab_old/*.so; do cp $o $CUR; run old_cfg2 "--cfg 2 --steps 300 --warmup 10"; done
cp /tmp/cur.so $CUR; run new_cfg4 "--cfg 4 --steps 30 --warmup 3"
for o in ab_old/*.so; do cp $o $CUR; run old_cfg4 "--cfg 4 --steps 30 --warmup 3"; done
cp /tmp/cur.so $CUR; run new_cfg5 "--cfg 5 --features 2500 --steps 15 --warmup 2"
for o in ab_old/*.so; do cp $o $CUR; run old_cfg5 "--cfg 5 --features 2500 --steps 15 --warmup 2"; done
cp /tmp/cur.so $CUR
tail -3 $OUT/err
