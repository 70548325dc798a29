#!/bin/bash
# round 5 experiment loop: smoke + a parity subset through the tree's library, then bench lines for a list of debug-option settings,
# with ab_old/base.so (the round's starting build) first and last.  usage: gpu_r5.sh TAG "opt1 opt2 ..." [pytest -k expression]
set -u
TAG=${1:-r5}
OPTS=${2:-}
KEXPR=${3:-}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
if [ -n "$KEXPR" ]; then
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_known_answer.py tests/test_ref_fixtures.py -q -m gpu -x -p no:cacheprovider -k "$KEXPR" 2>&1 | tail -15 > $OUT/pytest.txt
  cat $OUT/pytest.txt
fi
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/cur.so
B="python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-300} --warmup 10 ${BARGS:-}"
run() { # name, extra args
  timeout 120 $B $2 > $OUT/$1.json 2>> $OUT/err
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
[ -f ab_old/base.so ] && { cp ab_old/base.so $CUR; run base_a ""; }
cp /tmp/cur.so $CUR; run cur ""
for o in $OPTS; do run "cur_$o" "--debug-option $o"; done
[ -f ab_old/base.so ] && { cp ab_old/base.so $CUR; run base_b ""; }
cp /tmp/cur.so $CUR
tail -3 $OUT/err
