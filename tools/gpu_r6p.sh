#!/bin/bash
# Round 6: phase counters of k_feat_y (developer builds of the base and of the tree) + alternating bench lines, one box.  Usage: gpu_r6p.sh TAG [cfg3] [cfg4] [cfg2]
set -u
TAG=${1:-r6p}
shift || true
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/new.so
for w in "$@"; do
  n=${w#cfg}
  for which in base new; do
    if [ $which = base ]; then cp tools/_prof/libovgpu_dev_base.so $CUR; else cp tools/_prof/libovgpu_dev.so $CUR; fi
    timeout 100 python tools/dev_featy_phases.py $n 2>&1 | grep -v amdgpu.ids | tee $OUT/phases_${which}_$w.txt
  done
done
B="python bench.py --no-cpu-baseline --no-extras"
run() {
  timeout 90 $B $2 > $OUT/$1.json 2>> $OUT/err
  python - $OUT/$1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["roofline"]["compression"]
    print(sys.argv[1].split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "system %.4f"%d["roofline"]["avg_ms_per_launch"], "compress %.4f"%c["avg_ms_per_launch"], "update %.4f"%d["roofline"]["update_ms_device"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for rep in 1 2; do
for w in "$@"; do
  case $w in
    cfg3) A="--steps 300 --warmup 10";;
    cfg2) A="--cfg 2 --steps 300 --warmup 10";;
    cfg4) A="--cfg 4 --steps 30 --warmup 3";;
    *) continue;;
  esac
  cp /tmp/new.so $CUR; run new_${w}_$rep "$A"
  cp ab_old/base.so $CUR; run base_${w}_$rep "$A"
done
done
cp /tmp/new.so $CUR
tail -2 $OUT/err
