cd /root/repo
mkdir -p gpurun_out/g2
bash tools/gpu_quick.sh g2 > gpurun_out/g2/quick.txt 2>&1
timeout 120 python tools/dev_mode_a_time.py 3 > gpurun_out/g2/time3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/g2/prof -o s -- python /root/repo/tools/dev_mode_a_time.py 3 > /dev/null 2>&1
cd /root/repo
f=$(find gpurun_out/g2/prof -name "*.db" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f modea > gpurun_out/g2/prof_modea.txt
rm -rf gpurun_out/g2/prof
cat gpurun_out/g2/quick.txt gpurun_out/g2/time3.txt; head -30 gpurun_out/g2/prof_modea.txt | cut -c1-150
