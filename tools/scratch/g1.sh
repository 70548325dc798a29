cd /root/repo
timeout 300 python -m pytest tests -q -m gpu -x -s -p no:cacheprovider -k "mode_a" 2>&1 | grep -v "^$" | tail -25 > gpurun_out/g1/pytest.txt
timeout 120 python tools/dev_mode_a_time.py 3 > gpurun_out/g1/time3.txt 2>&1
timeout 120 python tools/dev_mode_a_time.py 2 > gpurun_out/g1/time2.txt 2>&1
cat gpurun_out/g1/pytest.txt gpurun_out/g1/time3.txt gpurun_out/g1/time2.txt
