#!/bin/bash
# as tools/gpu_bitcompare.sh (one dump per build), bench lines at configs[3] on one GPU and at one rank's share of configs[4]
set -u
TAG=${1:-bitcmp2}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/new.so
P=ab_old/problems.pkl
cp ab_old/base.so $CUR
timeout 60 python tools/dev_bitcompare.py dump /tmp/base_a.npz $P > $OUT/dump_base_a.txt 2>&1
cp /tmp/new.so $CUR
timeout 60 python tools/dev_bitcompare.py dump /tmp/new.npz $P > $OUT/dump_new.txt 2>&1
tail -2 $OUT/dump_new.txt
echo "== new build against base" | tee $OUT/compare.txt
python tools/dev_bitcompare.py compare /tmp/base_a.npz /tmp/new.npz 2>&1 | tee -a $OUT/compare.txt
B="python bench.py --no-cpu-baseline --no-extras"
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
cp /tmp/new.so $CUR; run new_cfg4 "--cfg 4 --steps 30 --warmup 3"
cp ab_old/base.so $CUR; run base_cfg4 "--cfg 4 --steps 30 --warmup 3"
cp /tmp/new.so $CUR; run new_cfg3 "--steps 300 --warmup 10"
cp ab_old/base.so $CUR; run base_cfg3 "--steps 300 --warmup 10"
cp /tmp/new.so $CUR
tail -2 $OUT/err
