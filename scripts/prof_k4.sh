cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_k4 -- python3 /root/repo/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-f32-mode > /dev/null 2>&1
cd /root/repo
grep "conv_\|fc_\|traj_prog" gpurun_out/prof_k4/*/*kernel_stats.csv | cut -d, -f1-4,6-7 | cut -c60-220
