#!/bin/bash
# Developer script (GPU): several rocprofv3 --pmc passes over a few default updates, to see what the per-feature kernels wait for.
# usage: tools/pmc_deep.sh <tag> <cfg> <features>
TAG=${1:-pmc}; CFG=${2:-3}; F=${3:-2000}
OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
CMD="python /root/repo/tools/dev_prof_update.py $CFG $F 4"
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $SET -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*.db" | head -1)
  [ -n "$f" ] && python /root/repo/tools/pmc_summary.py $f "$SET" > $OUT/p$i.txt
  rm -rf $OUT/p$i
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL
SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES
SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F64
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
SETS
ls $OUT; for f in $OUT/p*.txt; do echo "== $f"; grep "k_feat_y\|k_gram<\|k_chol_factor" $f | head -30; done
