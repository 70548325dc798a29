#!/bin/bash
# Round-end measurement on the GPU box: full GPU test suite, smoke, bench (1 GPU), rocprofv3 kernel stats and the PMC passes of
# the same bench command.  Everything lands under gpurun_out/<tag>/; tools/collect_profiles.sh copies the summaries to profiles/.
set -u
TAG=${1:-final}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12 > $OUT/pytest_gpu.txt
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
fi
timeout 300 python -m pytest tests/test_rpng_sim_loop.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep "GPU- vs reference\|passed\|failed" > $OUT/rpng_sim_loop.txt
timeout 120 open_vins_amd/shim/selftest --time 2000 9 > $OUT/shim_time.json 2>/dev/null
timeout 120 open_vins_amd/shim/selftest --time 800 9 >> $OUT/shim_time.json 2>/dev/null
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -s -p no:cacheprovider -k "conditioning_sweep or round1_route" 2>&1 | grep "cond(P_DD)\|gram then\|passed\|failed" > $OUT/conditioning_sweep.txt
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 120 python bench.py --cfg 2 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_cfg2.json 2>> $OUT/bench.err
timeout 120 python bench.py --cfg 2 --features 10000 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_stereo_10k.json 2>> $OUT/bench.err
timeout 120 python bench.py --cfg 4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_cfg4_one_gpu.json 2>> $OUT/bench.err
timeout 120 python bench.py --route tsqr --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_tsqr.json 2>> $OUT/bench.err
timeout 200 python bench.py --cfg 5 --features 2500 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_cfg5_share_f64.json 2>> $OUT/bench.err
timeout 200 python bench.py --cfg 5 --features 2500 --gram-fp32 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_cfg5_share_fp32_gram.json 2>> $OUT/bench.err
timeout 120 python tools/dev_mode_a_time.py 3 2>/dev/null > $OUT/mode_a_times.txt
timeout 120 python tools/dev_mode_a_time.py 2 2>/dev/null >> $OUT/mode_a_times.txt
timeout 120 python tools/dev_mode_a_time.py 4 1250 2>/dev/null >> $OUT/mode_a_times.txt
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extras"
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/prof_modea -o s -- python /root/repo/tools/dev_mode_a_time.py 3 > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o s -- $B --steps 20 --warmup 5 > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats2 -o s -- $B --cfg 2 --steps 20 --warmup 5 > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats3 -o s -- $B --cfg 4 --steps 6 --warmup 2 > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -o f -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -o w -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/prof_sq1 -o q -- $B --steps 5 --warmup 2 > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS -d $OUT/prof_sq2 -o q -- $B --steps 5 --warmup 2 > /dev/null 2>&1
cd /root/repo
for d in prof_stats prof_stats2 prof_stats3 prof_modea; do f=$(find $OUT/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $d > $OUT/${d}.txt; done
for d in prof_fetch prof_write prof_sq1 prof_sq2; do f=$(find $OUT/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $d > $OUT/${d}.txt; done
rm -rf $OUT/prof_modea $OUT/prof_stats $OUT/prof_stats2 $OUT/prof_stats3 $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq1 $OUT/prof_sq2
cat $OUT/pytest_gpu.txt 2>/dev/null; tail -2 $OUT/smoke.txt 2>/dev/null; cut -c1-400 $OUT/bench.json; echo; for f in bench_cfg2 bench_stereo_10k bench_cfg4_one_gpu bench_tsqr; do cut -c1-200 $OUT/$f.json; echo; done; for f in bench_cfg5_share_f64 bench_cfg5_share_fp32_gram; do cut -c1-200 $OUT/$f.json; echo; done; cat $OUT/conditioning_sweep.txt | cut -c1-200; head -24 $OUT/prof_stats.txt | cut -c1-60,72-128; cat $OUT/mode_a_times.txt
