#!/bin/bash
# developer build (tools/_prof/libovgpu_dev.so: -DOVG_FEAT_PROF -DOVG_FEAT_ABLATE) in place of the tree's library: phase counters and ablation of k_feat_y
set -u
TAG=${1:-dev}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
CUR=open_vins_amd/csrc/libovgpu.so
cp $CUR /tmp/cur.so
cp tools/_prof/libovgpu_dev.so $CUR
timeout 120 python tools/dev_featy_phases.py 3 2>&1 | tee $OUT/phases.txt
timeout 300 python tools/dev_featy_ablate.py 3 2>&1 | tee $OUT/ablate.txt
cp /tmp/cur.so $CUR
