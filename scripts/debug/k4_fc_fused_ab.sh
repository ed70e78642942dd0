# fc_fused_f16x3_kernel (one launch) against fc_partial_f16x3 + fc_tail: time per encoder launch sequence, kernel
# durations (rocprofv3), and the features compared bit for bit.   bash scripts/debug/k4_fc_fused_ab.sh   (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/k4fc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  MMF_K4_FC_TWO_LAUNCHES=1 python3 $R/scripts/debug/k4_fc_probe.py /tmp/fc_two.pt 2>&1 | grep "ms per"
  MMF_K4_FC_TWO_LAUNCHES=0 python3 $R/scripts/debug/k4_fc_probe.py /tmp/fc_one.pt 2>&1 | grep "ms per"
done
python3 - <<'PY'
import torch
a, b = torch.load("/tmp/fc_two.pt"), torch.load("/tmp/fc_one.pt")
for k in a:
    print(k, "features bit-identical:", bool(torch.equal(a[k].view(torch.int32), b[k].view(torch.int32))), "finite:", bool(torch.isfinite(b[k]).all()))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc_prof_one -- python3 $R/scripts/debug/k4_fc_probe.py /tmp/x.pt > /dev/null 2>&1
cat > /tmp/fc_two.py <<'PY'
import os, runpy, sys
os.environ["MMF_K4_FC_TWO_LAUNCHES"] = "1"
sys.argv = [sys.argv[1]] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc_prof_two -- python3 /tmp/fc_two.py $R/scripts/debug/k4_fc_probe.py /tmp/x.pt > /dev/null 2>&1
for d in one two; do
  echo "# kernel stats, $d"
  f=$(find /tmp/fc_prof_$d -name "*kernel_stats.csv" | head -1)
  grep -E "fc_|resident" $f | cut -c1-400
  f=$(find /tmp/fc_prof_$d -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "fc_" in r["Kernel_Name"]]
# the largest problem's launches: the longest ones
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"].split("(")[0][-40:], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    v.sort()
    print(f"  {k}: {len(v)} launches, longest quarter mean {sum(v[-len(v)//4:]) / max(1, len(v)//4):.1f} us (4096 x 2)")
PY
done
