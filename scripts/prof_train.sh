cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_train -- python3 /root/repo/scripts/bench_train.py --steps 4 --backends hip > /dev/null 2>&1
cd /root/repo
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_train/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {r["Calls"]:>6} calls  {r["Name"][:110]}')
PY
