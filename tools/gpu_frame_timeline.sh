#!/bin/bash
# kernel timeline of one whole FRAME (set_state + set_features + update + read-back): rocprofv3 --kernel-trace of tools/dev_frame.py
set -u
TAG=${1:-frame}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s -- python /root/repo/tools/dev_frame.py 30 > $OUT/frame_traced.txt 2>&1
cd /root/repo
f=$(find $OUT/prof -name "*.db" | head -1)
python tools/prof_timeline.py $f 20 k_build_tables > $OUT/frame_timeline.txt
python tools/prof_summary.py $f frame > $OUT/frame_stats.txt
rm -rf $OUT/prof
cat $OUT/frame_timeline.txt; tail -1 $OUT/frame_traced.txt
