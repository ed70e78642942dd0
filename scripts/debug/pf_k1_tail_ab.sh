# K1 in the tail of the step's last measurement launch (default) against K1 as its own launch (MMF_PF_FUSE_K1=0): bench lines,
# alternating, and the kernel list of one fused run.   bash scripts/debug/pf_k1_tail_ab.sh   (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs --no-f32-mode --no-kernel-timers"
line() { python3 -c "
import json,sys
l=[json.loads(x) for x in sys.stdin if x.startswith('{')][-1]
print('  $1: %.4e particle-steps/s  %.2f us/step' % (l['value'], 1e3*l['ms_per_step']))"; }
for shape in ${SHAPES:-32x4096 64x4096 256x4096 256x1024 1024x4096}; do
  N=${shape%x*}; M=${shape#*x}
  echo "# door_pf $N x $M, 64 steps"
  for rep in 1 2; do
    MMF_PF_FUSE_K1=1 python3 $R/bench.py --workload door_pf --batch $N --particles $M --steps 64 --warmup 8 $LEAN 2>/dev/null | line "K1 in the tail "
    MMF_PF_FUSE_K1=0 python3 $R/bench.py --workload door_pf --batch $N --particles $M --steps 64 --warmup 8 $LEAN 2>/dev/null | line "K1 its own launch"
  done
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/k1tail
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1tail -- python3 $R/bench.py --workload door_pf --batch 32 --particles 4096 --steps 32 --warmup 4 $LEAN > /dev/null 2>&1
echo "# kernels of a fused 32 x 4096 run (rocprofv3 --stats): calls, average us"
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/k1tail/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("  %-90s %6s %9.1f" % (r["Name"].replace("(anonymous namespace)::", "")[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
