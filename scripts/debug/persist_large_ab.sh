# The persistent step loop beyond one round of tiles per wave: bench.py's door PF at small batches, ONE launch against the
# loop of launches.   bash scripts/debug/persist_large_ab.sh   (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs --no-f32-mode --no-kernel-timers"  # (a kernel timer records events between the launches: the loop of launches)
run() {  # batch particles persistent
  MMF_PF_PERSISTENT=$3 python3 $R/bench.py --workload door_pf --batch $1 --particles $2 --steps 64 --warmup 8 $LEAN 2>/dev/null | python3 -c "
import json,sys
l=[json.loads(x) for x in sys.stdin if x.startswith('{')][-1]
print('  %4d x %5d persistent=$3: %.4e particle-steps/s  %.2f us/step' % ($1, $2, l['value'], 1e3*l['ms_per_step']))"
}
for shape in ${SHAPES:-16x4096 8x4096 32x2048 24x3000 64x1024 40x1500 32x4096 64x4096}; do
  set -- ${shape%x*} ${shape#*x}
  run $1 $2 1
  run $1 $2 0
done
