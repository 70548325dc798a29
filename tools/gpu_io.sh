#!/bin/bash
# after the staging / refactor changes: quick parity subset, the bench line with its host-to-host extras, the C++ drop-in timing
set -u
TAG=${1:-io}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_shim.py tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "shim or track" 2>&1 | tail -6 > $OUT/pytest_gpu.txt
timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 10 --no-extras > $OUT/bench.json 2> $OUT/bench.err
timeout 120 open_vins_amd/shim/selftest --time 2000 9 > $OUT/shim_time.json 2>> $OUT/bench.err
timeout 120 open_vins_amd/shim/selftest --time 800 9 >> $OUT/shim_time.json 2>> $OUT/bench.err
cat $OUT/pytest_gpu.txt
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step %.4f loops %s" % (d["ms_per_step"], d["ms_per_step_timed_loops"]))
print("pcie_inclusive_ms", d.get("pcie_inclusive_ms"), "mode_a", {k:v["ms_host_to_host"] for k,v in d.get("mode_a",{}).items()})
print("shim", d.get("shim"))
print("configs3_single_gpu", d.get("configs3_single_gpu",{}).get("ms_per_step"), "scaling_model", d.get("scaling_model",{}).get("predicted_ms"))
PY
cat $OUT/shim_time.json; tail -3 $OUT/bench.err
timeout 120 open_vins_amd/shim/selftest --time-resident 2000 5 2>> $OUT/bench.err | tee -a $OUT/shim_time.json
timeout 120 open_vins_amd/shim/selftest --time-resident 800 5 2>> $OUT/bench.err | tee -a $OUT/shim_time.json
