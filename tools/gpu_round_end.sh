#!/bin/bash
# Round-end measurement on the GPU box: full GPU test suite, smoke, bench (1 GPU), rocprofv3 kernel stats and the two PMC
# passes of the same bench command.  Everything lands under gpurun_out/final/ (copied to profiles/ by hand afterwards).
set -u
OUT=/root/repo/gpurun_out/final
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $OUT/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
timeout 120 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 60 python bench.py --features 10000 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_10k.json 2>> $OUT/bench.err
timeout 60 python bench.py --features 2000 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_2k.json 2>> $OUT/bench.err
OVGPU_COMPRESS=tsqr timeout 60 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_tsqr.json 2>> $OUT/bench.err
OVGPU_COMPRESS=cholqr timeout 60 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_cholqr.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o s -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -o f -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -o w -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cat $OUT/pytest_gpu.txt; cat $OUT/smoke.txt | tail -2; cut -c1-400 $OUT/bench.json; echo; cut -c1-250 $OUT/bench_10k.json; echo; cut -c1-250 $OUT/bench_tsqr.json; echo; ls -R $OUT | head -30
