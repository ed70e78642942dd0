cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_ekf -- python3 /root/repo/bench.py --workload door_ekf --steps 32 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
cd /root/repo
head -14 gpurun_out/prof_ekf/*/*kernel_stats.csv | cut -c1-160
