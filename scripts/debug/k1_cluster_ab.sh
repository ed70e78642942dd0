R=${GRAFT_REPO_ROOT:-/root/repo}
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs --no-f32-mode"
for rep in 1 2; do
for c in 1 0; do  # 1: cluster, 0: one workgroup per trajectory (the default)
  echo "# MMF_K1_CLUSTER=$c: bench.py --workload door_pf --batch 32 --steps 64"
  MMF_K1_CLUSTER=$c python $R/bench.py --workload door_pf --batch 32 --steps 64 $LEAN 2>/dev/null | grep "^{" | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e9,4),'e9', round(j['ms_per_step']*1e3,2),'us/step', {k:round(v['avg_ms']*1e3,2) for k,v in j['kernels'].items()})"
done
done
cd /tmp && export TMPDIR=/tmp
for c in 1 0; do  # 1: cluster, 0: one workgroup per trajectory (the default)
  cat > /tmp/k1run.py <<PY
import os, runpy, sys
os.environ["MMF_K1_CLUSTER"] = "$c"
sys.argv = ["$R/scripts/bench_k1.py"]
runpy.run_path("$R/scripts/bench_k1.py", run_name="__main__")
PY
  rm -rf /tmp/k1prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1prof_$c -- python3 /tmp/k1run.py > /dev/null 2>&1
  echo "# rocprofv3 kernel durations, MMF_K1_CLUSTER=$c (scripts/bench_k1.py shapes: 256x4096, 256x1024, 1024x4096, 32x4096, 32x300, 256x8192)"
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/k1prof_$c/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pf_res" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"].split("(")[0][-60:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    by[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    v = sorted(v)
    print(k, "launches", len(v), "median us", round(v[len(v)//2], 2), "min", round(v[0], 2))
PY
done
