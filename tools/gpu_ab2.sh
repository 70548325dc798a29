#!/bin/bash
# same-box A/B of the tree's libovgpu.so against every ab_old/*.so: bench lines at configs[2], [1], [3] on one GPU
set -u
TAG=${1:-ab}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
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
  cp /tmp/cur.so $CUR; run cur_cfg3_$rep "--steps 300 --warmup 10"
  for o in ab_old/*.so; do cp $o $CUR; run $(basename $o .so)_cfg3_$rep "--steps 300 --warmup 10"; done
done
cp /tmp/cur.so $CUR; run cur_cfg4 "--cfg 4 --steps 30 --warmup 3"
for o in ab_old/*.so; do cp $o $CUR; run $(basename $o .so)_cfg4 "--cfg 4 --steps 30 --warmup 3"; done
cp /tmp/cur.so $CUR
tail -3 $OUT/err
