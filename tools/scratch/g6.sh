cd /root/repo
mkdir -p gpurun_out/g6
timeout 400 python -m pytest tests -q -m gpu -x -s -p no:cacheprovider -k "mode_a or semi_definite or long_loop or conditioning_sweep or slam_mode_a or shim" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/g6/pytest.txt
timeout 120 python tools/dev_mode_a_time.py 3 > gpurun_out/g6/time3.txt 2>&1
grep -v "^\.mode A\|^\.\?cond(P_DD).*route \(gram\|tsqr\)" gpurun_out/g6/pytest.txt | cut -c1-260 | tail -22; cat gpurun_out/g6/time3.txt
