#!/bin/bash
# Round 6's developer call: bit comparison of the tree's library against ab_old/base.so (17 shapes), a parity subset, alternating bench lines and
# the host-side breakdown.  Usage: gpu_r6.sh TAG [bitcompare args...]
set -u
TAG=${1:-r6}
shift || true
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
bash tools/gpu_bitcompare.sh $TAG "$@" 2>&1 | tee $OUT/bitcompare_log.txt
timeout 100 python tools/dev_host_breakdown.py 3 2>&1 | tee $OUT/host_breakdown.txt
( timeout ${PYTEST_LIMIT:-240} python -m pytest tests -q -m gpu -n 6 -p no:cacheprovider -x ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -8 ) | tee $OUT/pytest_gpu.txt
