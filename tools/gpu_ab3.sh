#!/bin/bash
# same-box A/B of the tree's libovgpu.so against every ab_old/*.so: ONE short bench line each at configs[2] (cur first and last)
set -u
TAG=${1:-ab3}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/cur.so
B="python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-300} --warmup 10"
run() { # name
  timeout 60 $B > $OUT/$1.json 2>> $OUT/err
  python - $OUT/$1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["roofline"]["compression"]
    print(sys.argv[1].split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "min %.4f"%d["ms_per_step_min"], "system %.4f"%d["roofline"]["avg_ms_per_launch"], "compress %.4f"%c["avg_ms_per_launch"], "update %.4f"%d["roofline"]["update_ms_device"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
cp /tmp/cur.so $CUR; run cur_a
for o in ab_old/*.so; do cp $o $CUR; run $(basename $o .so); done
cp /tmp/cur.so $CUR; run cur_b
cp /tmp/cur.so $CUR
tail -3 $OUT/err
