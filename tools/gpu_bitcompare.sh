#!/bin/bash
# The tree's libovgpu.so against ab_old/base.so on ONE box: outputs bit for bit over tools/dev_bitcompare.py's shapes (with the base
# build against itself as the determinism control), then alternating bench lines at configs[2] and one at configs[3] on one GPU.
set -u
TAG=${1:-bitcmp}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/new.so
P=ab_old/problems.pkl
cp ab_old/base.so $CUR
timeout 60 python tools/dev_bitcompare.py dump /tmp/base_a.npz $P > $OUT/dump_base_a.txt 2>&1
timeout 60 python tools/dev_bitcompare.py dump /tmp/base_b.npz $P > $OUT/dump_base_b.txt 2>&1
cp /tmp/new.so $CUR
timeout 60 python tools/dev_bitcompare.py dump /tmp/new.npz $P > $OUT/dump_new.txt 2>&1
tail -3 $OUT/dump_new.txt
echo "== base against itself (determinism control)" | tee $OUT/compare.txt
python tools/dev_bitcompare.py compare /tmp/base_a.npz /tmp/base_b.npz 2>&1 | tee -a $OUT/compare.txt
echo "== new build against base" | tee -a $OUT/compare.txt
python tools/dev_bitcompare.py compare /tmp/base_a.npz /tmp/new.npz 2>&1 | tee -a $OUT/compare.txt
B="python bench.py --no-cpu-baseline --no-extras --warmup 10"
run() { # name, args
  timeout 60 $B $2 > $OUT/$1.json 2>> $OUT/err
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
for rep in 1; do
  cp /tmp/new.so $CUR; run new_cfg3_$rep "--steps 300"
  cp ab_old/base.so $CUR; run base_cfg3_$rep "--steps 300"
done
cp /tmp/new.so $CUR; run new_cfg2 "--cfg 2 --steps 300"
cp ab_old/base.so $CUR; run base_cfg2 "--cfg 2 --steps 300"
cp /tmp/new.so $CUR
