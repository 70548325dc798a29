#!/bin/bash
# Copies the summaries of a tools/gpu_round_end.sh run (gpurun_out/<tag>) into profiles/ under the round's prefix.
# usage: tools/collect_profiles.sh <tag> <prefix, e.g. r02>
set -eu
SRC=gpurun_out/$1; P=profiles/$2
cp $SRC/prof_stats.txt ${P}_kernel_stats_cfg3.txt
cp $SRC/prof_stats2.txt ${P}_kernel_stats_cfg2.txt
[ -f $SRC/prof_stats3.txt ] && cp $SRC/prof_stats3.txt ${P}_kernel_stats_cfg4_one_gpu.txt
cp $SRC/prof_fetch.txt ${P}_pmc_fetch_size.txt
cp $SRC/prof_write.txt ${P}_pmc_write_size.txt
cp $SRC/prof_sq1.txt ${P}_pmc_sq_waits.txt
cp $SRC/prof_sq2.txt ${P}_pmc_sq_insts.txt
cp $SRC/conditioning_sweep.txt ${P}_conditioning_sweep.txt
cp $SRC/bench.json ${P}_bench.json
cp $SRC/bench_cfg2.json ${P}_bench_cfg2_800_features.json
cp $SRC/bench_stereo_10k.json ${P}_bench_stereo_10k_features.json
cp $SRC/bench_cfg4_one_gpu.json ${P}_bench_cfg4_one_gpu.json
cp $SRC/bench_tsqr.json ${P}_bench_tsqr_route.json
cp $SRC/bench_cfg5_share_f64.json ${P}_bench_cfg5_one_gpu_share.json
cp $SRC/bench_cfg5_share_fp32_gram.json ${P}_bench_cfg5_one_gpu_share_fp32_gram.json
[ -f $SRC/rpng_sim_loop.txt ] && cp $SRC/rpng_sim_loop.txt ${P}_rpng_sim_closed_loop.txt
[ -f $SRC/shim_time.json ] && cp $SRC/shim_time.json ${P}_shim_dropin_times.json
[ -f $SRC/pytest_gpu.txt ] && cp $SRC/pytest_gpu.txt ${P}_pytest_gpu.txt
[ -f $SRC/mode_a_times.txt ] && cp $SRC/mode_a_times.txt ${P}_mode_a_times.txt
[ -f $SRC/prof_modea.txt ] && cp $SRC/prof_modea.txt ${P}_kernel_stats_mode_a.txt
python tools/make_pmc_json.py $SRC/prof_fetch.txt $SRC/prof_write.txt 3 ${P}_pmc.json
ls -la profiles | grep $2
