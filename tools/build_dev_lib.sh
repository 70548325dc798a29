#!/bin/bash
# developer build of the library (phase counters + ablation switches of the per-feature kernels): tools/_prof/libovgpu_dev.so (git-ignored)
set -eu
cd "$(dirname "$0")/../open_vins_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -DOVG_FEAT_PROF -DOVG_FEAT_ABLATE ${EXTRA:-}"
O=../../tools/_prof
/opt/rocm/bin/hipcc $F -c ovgpu_api.hip -o $O/dev_api.o &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=iterative-ilp -c ovgpu_featy_tu.hip -o $O/dev_featy.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $O/dev_api.o $O/dev_featy.o -o $O/libovgpu_dev.so
ls -la $O/libovgpu_dev.so
