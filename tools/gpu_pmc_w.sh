#!/bin/bash
# PMC passes (instruction mix, waits) of a short bench run with a debug option; usage: gpu_pmc_w.sh TAG "featy_shape=3"
set -u
TAG=${1:-pmcw}; OPT=${2:-featy_shape=3}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --debug-option $OPT"
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/p1 -o q -- $B > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS -d $OUT/p2 -o q -- $B > /dev/null 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD -d $OUT/p3 -o q -- $B > /dev/null 2>&1
cd /root/repo
for d in p1 p2 p3; do f=$(find $OUT/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f $d > $OUT/$d.txt; rm -rf $OUT/$d; done
grep -h "k_feat_w\|k_feat_y\|^kernel\|^#" $OUT/p1.txt $OUT/p2.txt $OUT/p3.txt | cut -c1-110
