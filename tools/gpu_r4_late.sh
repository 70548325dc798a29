#!/bin/bash
# late round 4 (6 GPU-minutes left): the device FeatureDatabase's new entry points against the reference's class, the track-store
# tests that exercise the re-routed not_containing_newer, the IMU-intrinsics fixture; then the driver's bench command; then smoke
set -u
TAG=${1:-r4late}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 170 python -m pytest tests/test_gpu_track_store.py tests/test_gpu_parity.py tests/test_ref_fixtures.py -q -m gpu -x -p no:cacheprovider \
  -k "track_store or long_lived or imu_intrinsics_state" 2>&1 | tail -15 > $OUT/pytest.txt
cat $OUT/pytest.txt
timeout 110 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms/step", d["ms_per_step"], "imu_intrinsics_state", d.get("imu_intrinsics_state"), "frac", d["roofline"]["frac"])
except Exception as e:
    print("bench line ERR", e)
PY
timeout 40 python tools/dev_track_store_time.py 2>&1 | tail -2 | tee $OUT/track_store_times.json
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
