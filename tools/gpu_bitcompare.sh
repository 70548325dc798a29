#!/bin/bash
# The tree's libovgpu.so against ab_old/base.so on ONE box: every output of tools/dev_bitcompare.py's 17 shapes bit for bit, then alternating
# bench lines.  Usage: gpu_bitcompare.sh TAG [control] [cfg3] [cfg2] [cfg4] [cfg5]
#   control   also the base build against itself (the determinism control)
#   cfgN      bench lines to alternate: cfg3 = configs[2] (the headline), cfg2 = configs[1], cfg4 = configs[3] on one GPU, cfg5 = one rank's share of configs[4]
# (Since k_feat_anchor, round 5, k_triangulate compiles to differently paired multiply-adds: against a base older than that commit p_FinG and what follows
# from it differ at 1e-13; take a base from that commit or later for a bit-for-bit answer.)
# ab_old/problems.pkl: the problems, generated here on the CPU (`python tools/dev_bitcompare.py prepare ab_old/problems.pkl`).
# Late round 4 this was run as: `bitcmp control cfg3 cfg2` (the 4-wavefront kernel in its own translation unit), `bitcmp2 cfg4 cfg5` (the 8-wavefront and
# block-row kernels), `prio cfg4 cfg3` (the gate chain's wave priority): profiles/r04_late_*.
set -u
TAG=${1:-bitcmp}
shift || true
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/new.so
P=ab_old/problems.pkl
cp ab_old/base.so $CUR
timeout 60 python tools/dev_bitcompare.py dump /tmp/base_a.npz $P > $OUT/dump_base_a.txt 2>&1
case " $* " in *" control "*)
  timeout 60 python tools/dev_bitcompare.py dump /tmp/base_b.npz $P > $OUT/dump_base_b.txt 2>&1
  echo "== base against itself (determinism control)" | tee $OUT/compare.txt
  python tools/dev_bitcompare.py compare /tmp/base_a.npz /tmp/base_b.npz 2>&1 | tee -a $OUT/compare.txt;;
esac
cp /tmp/new.so $CUR
timeout 60 python tools/dev_bitcompare.py dump /tmp/new.npz $P > $OUT/dump_new.txt 2>&1
tail -2 $OUT/dump_new.txt
echo "== new build against base" | tee -a $OUT/compare.txt
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
for w in "$@"; do
  case $w in
    cfg3) A="--steps 300 --warmup 10";;
    cfg2) A="--cfg 2 --steps 300 --warmup 10";;
    cfg4) A="--cfg 4 --steps 30 --warmup 3";;
    cfg5) A="--cfg 5 --features 2500 --steps 15 --warmup 2";;
    *) continue;;
  esac
  cp /tmp/new.so $CUR; run new_$w "$A"
  cp ab_old/base.so $CUR; run base_$w "$A"
done
cp /tmp/new.so $CUR
tail -2 $OUT/err
