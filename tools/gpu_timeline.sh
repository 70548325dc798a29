#!/bin/bash
# kernel timeline of one update (rocprofv3 --kernel-trace of a short bench run -> tools/prof_timeline.py)
set -u
TAG=${1:-tl}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s -- python /root/repo/bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 4 --stage-events-every 1000 > /dev/null 2>&1
cd /root/repo
f=$(find $OUT/prof -name "*.db" | head -1)
python tools/prof_timeline.py $f 20 > $OUT/timeline.txt
python tools/prof_timeline.py $f 30 >> $OUT/timeline.txt
rm -rf $OUT/prof
cat $OUT/timeline.txt
