#!/bin/bash
# kernel statistics (rocprofv3 --kernel-trace --stats) of the bench command at configs[2] and configs[1]
set -u
TAG=${1:-prof}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extras"
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o s -- $B --steps 20 --warmup 5 > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats2 -o s -- $B --cfg 2 --steps 20 --warmup 5 > /dev/null 2>&1
cd /root/repo
for d in prof_stats prof_stats2; do f=$(find $OUT/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $d > $OUT/${d}.txt; done
rm -rf $OUT/prof_stats $OUT/prof_stats2
head -30 $OUT/prof_stats.txt | cut -c1-60,72-128; head -30 $OUT/prof_stats2.txt | cut -c1-60,72-128
