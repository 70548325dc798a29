#!/bin/bash
# long-track kernels (k_feat_y<8,17,1>, k_feat_y_big): parity on the shapes that reach them, then same-box A/B against ab_old/*.so
set -u
TAG=${1:-long}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -p no:cacheprovider -k "long_tracks or block_row or cfg4 or (cfg5_geometry and 240) or random_shapes or fp32_gram_variant" 2>&1 | tail -4 > $OUT/pytest.txt
cat $OUT/pytest.txt
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/cur.so
B="python bench.py --no-cpu-baseline --no-extras"
run() {
  timeout 300 $B $2 > $OUT/$1.json 2>> $OUT/err
  python - $OUT/$1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "system %.4f frac %.3f"%(d["roofline"]["avg_ms_per_launch"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for rep in 1 2; do
  cp /tmp/cur.so $CUR; run new_cfg4_$rep "--cfg 4 --steps 30 --warmup 3"
  for o in ab_old/*.so; do cp $o $CUR; run old_cfg4_$rep "--cfg 4 --steps 30 --warmup 3"; done
done
cp /tmp/cur.so $CUR; run new_cfg5 "--cfg 5 --features 2500 --steps 15 --warmup 2"
for o in ab_old/*.so; do cp $o $CUR; run old_cfg5 "--cfg 5 --features 2500 --steps 15 --warmup 2"; done
cp /tmp/cur.so $CUR
tail -2 $OUT/err
