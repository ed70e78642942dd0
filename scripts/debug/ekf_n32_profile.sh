R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs"
rm -rf /tmp/ekf32
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ekf32 -- python3 $R/bench.py --workload door_ekf --batch 32 --steps 32 --warmup 4 $LEAN > /dev/null 2>&1
cd $R
python scripts/gpu_busy.py $(find /tmp/ekf32 -name "*kernel_trace.csv" | head -1) --top 12 --kernels 14
head -25 $(find /tmp/ekf32 -name "*kernel_stats.csv" | head -1) | cut -c1-200
