# The EKF step loop as ONE persistent launch (default) against the loop of 2 T launches (MMF_EKF_PERSISTENT=0): bench lines.
#   bash scripts/debug/ekf_persistent_ab.sh   (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs"
line() { python3 -c "
import json,sys
l=[json.loads(x) for x in sys.stdin if x.startswith('{')][-1]
print('  $1: %.4e trajectory-steps/s  %.2f us/step' % (l['value'], 1e3*l['ms_per_step']))"; }
for N in 1024 32 256 8192; do
  echo "# door crossmodal EKF, $N trajectories, 32 steps"
  for rep in 1 2; do
    MMF_EKF_PERSISTENT=1 python3 $R/bench.py --workload door_ekf --batch $N --steps 32 --warmup 4 $LEAN 2>/dev/null | line "persistent"
    MMF_EKF_PERSISTENT=0 python3 $R/bench.py --workload door_ekf --batch $N --steps 32 --warmup 4 $LEAN 2>/dev/null | line "launches  "
  done
done
