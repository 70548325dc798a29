cd /root/repo
mkdir -p gpurun_out/g4
timeout 300 python -m pytest tests -q -m gpu -x -s -p no:cacheprovider -k "mode_a or known_answer or cfg4_full_size" 2>&1 | grep -v "^$" | tail -25 > gpurun_out/g4/pytest.txt
timeout 120 python tools/dev_mode_a_time.py 3 > gpurun_out/g4/time3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/g4/prof -o s -- python /root/repo/tools/dev_mode_a_time.py 3 > /dev/null 2>&1
cd /root/repo
f=$(find gpurun_out/g4/prof -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f modea > gpurun_out/g4/prof_modea.txt
rm -rf gpurun_out/g4/prof
cat gpurun_out/g4/pytest.txt gpurun_out/g4/time3.txt; grep "pchol\|unwhiten\|gram_chol" gpurun_out/g4/prof_modea.txt | cut -c1-150
