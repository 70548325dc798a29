#!/bin/bash
# same-box A/B of two builds of libovgpu.so: ab_old/<name>.so against the tree's (bench line of configs[2], alternating)
set -u
TAG=${1:-ab}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/cur.so
B="python bench.py --no-cpu-baseline --no-extras"
for rep in 1 2; do
  for steps in 50 400; do
    cp /tmp/cur.so $CUR
    timeout 200 $B --steps $steps --warmup 10 > $OUT/new_${steps}_$rep.json 2>> $OUT/err
    for o in ab_old/*.so; do
      cp $o $CUR
      timeout 200 $B --steps $steps --warmup 10 > $OUT/old_${steps}_$rep.json 2>> $OUT/err
    done
  done
done
cp /tmp/cur.so $CUR
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["roofline"]["compression"]
    print(sys.argv[1].split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "system %.4f"%d["roofline"]["avg_ms_per_launch"], "compress %.4f"%c["avg_ms_per_launch"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
rocm-smi --showclocks 2>/dev/null | head -20
tail -3 $OUT/err
