# Socket power and shader clock (rocm-smi, sampled every ~0.5 s) while (a) the EKF bench's K4-dominated loop and (b) the PF
# bench's K2-dominated loop run.  bash scripts/debug/power_probe.sh  (GPU box) -> stdout
R=${GRAFT_REPO_ROOT:-/root/repo}
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs --no-f32-mode --preroll-seconds 4"
for W in door_ekf door_pf; do
  python3 $R/bench.py --workload $W $LEAN > /tmp/pp_$W.json 2>/dev/null &
  BP=$!
  for i in $(seq 1 60); do   # 30 s of samples; the report keeps the busiest ones
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket|sclk" | tr '\n' ' ' | sed 's/  */ /g'
    echo
    sleep 0.45
    kill -0 $BP 2>/dev/null || break
  done > /tmp/pp_$W.txt
  wait $BP
  echo "== $W: $(tail -1 /tmp/pp_$W.json | cut -c1-120)"
  python3 - /tmp/pp_$W.txt <<'PY'
import re, sys
rows = []
for l in open(sys.argv[1]):
    m, p = re.search(r"\((\d+)Mhz\)", l), re.search(r"\(W\): ([0-9.]+)", l)
    if m and p:
        rows.append((float(p.group(1)), int(m.group(1))))
busy = sorted(rows, reverse=True)[:8]
print("   samples", len(rows), "| the eight highest-power samples (W, sclk MHz):", busy)
PY
done
