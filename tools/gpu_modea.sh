#!/bin/bash
# mode A beyond 255 columns: parity shapes, then host-to-host times at one rank's share of configs[4] per route
set -u
TAG=${1:-modea}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_closed_loop.py -q -m gpu -x -s -p no:cacheprovider -k "mode_a" 2>&1 | grep "mode A\|passed\|failed\|Error\|error" | tail -30 > $OUT/pytest.txt
cat $OUT/pytest.txt
timeout 300 python tools/dev_mode_a_time.py 5 2500 2>/dev/null > $OUT/mode_a_cfg5.txt
timeout 120 python tools/dev_mode_a_time.py 5 240 2>/dev/null >> $OUT/mode_a_cfg5.txt
timeout 120 python tools/dev_mode_a_time.py 3 2>/dev/null >> $OUT/mode_a_cfg5.txt
cat $OUT/mode_a_cfg5.txt
