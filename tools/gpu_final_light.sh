#!/bin/bash
# The round's last call, sized for a few GPU-minutes: the default bench line, configs[1], kernel statistics + timeline of the headline
# (one rocprofv3 run), smoke(), then the GPU suite on six workers.  Everything lands under gpurun_out/<tag>/.
set -u
TAG=${1:-final_light}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 60 python bench.py --cfg 2 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_cfg2.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof -o s -- python /root/repo/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 --stage-events-every 1000 > /dev/null 2>&1
cd /root/repo
f=$(find $OUT/prof -name "*.db" | head -1)
python tools/prof_summary.py $f prof_stats > $OUT/prof_stats.txt
python tools/prof_timeline.py $f 20 > $OUT/timeline.txt
rm -rf $OUT/prof
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
cut -c1-300 $OUT/bench.json; echo; cut -c1-160 $OUT/bench_cfg2.json; echo; head -14 $OUT/timeline.txt; tail -1 $OUT/smoke.txt
( time timeout ${PYTEST_LIMIT:-200} python -m pytest tests -q -m gpu -n 6 -p no:cacheprovider --durations=5 2>&1 | tail -14 ) > $OUT/pytest_gpu.txt 2>&1
cat $OUT/pytest_gpu.txt
