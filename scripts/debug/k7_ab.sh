# traj_program_kernel in both regimes: many rows (the EKF bench's programs) and the 512 rows of a training step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/k7a; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k7a -- python3 $R/bench.py --workload door_ekf --steps 32 --warmup 4 --no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs > /tmp/k7a.json 2>/dev/null
grep traj_program_kernel $(find /tmp/k7a -name "*kernel_stats.csv" | head -1) | cut -c1-160
python3 -c "import json; j=json.loads([l for l in open('/tmp/k7a.json') if l.startswith('{')][-1]); print('ekf', j['value'], j['ms_per_step'])"
rm -rf /tmp/k7b; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k7b -- python3 $R/scripts/bench_reference_sizes.py --only train --backends hip --no-cpu --train-iters 10 > /dev/null 2>&1
grep traj_program_kernel $(find /tmp/k7b -name "*kernel_stats.csv" | head -1) | cut -c1-160
cd $R; python scripts/bench_reference_sizes.py --only train --backends hip --no-cpu 2>&1 | grep "^{" | cut -c100-230
