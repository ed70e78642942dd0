# EKF bench (door crossmodal EKF, 1024 trajectories): images per K4 launch sequence x linear-layer form.
#   bash scripts/debug/ekf_chunk_ab.sh   (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs"
for rep in 1 2; do
for chunk in 4096 8192; do
  for two in 1 0; do
    echo "# MMF_IMAGE_CHUNK=$chunk MMF_K4_FC_TWO_LAUNCHES=$two"
    MMF_IMAGE_CHUNK=$chunk MMF_K4_FC_TWO_LAUNCHES=$two python3 $R/bench.py --workload door_ekf --steps 32 --warmup 4 $LEAN 2>/dev/null | python3 -c "
import json,sys
l=[json.loads(x) for x in sys.stdin if x.startswith('{')][-1]
print(' value %.4e  ms_per_step %.4f  frac %.4f  achieved %.1f' % (l['value'], l['ms_per_step'], l['roofline']['frac'], l['roofline']['achieved']))"
  done
done
done
