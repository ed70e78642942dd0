R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6n
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tr_h
cat > /tmp/newp.py <<PY
import os, runpy, sys
os.environ["MMF_K4_PRECISION"] = "f16x3"
sys.argv = ["$R/scripts/bench_reference_sizes.py", "--only", "train", "--backends", "hip", "--no-cpu", "--train-iters", "10"]
runpy.run_path(sys.argv[0], run_name="__main__")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_h -o tr -- python3 /tmp/newp.py > /dev/null 2>&1
cp $(find /tmp/tr_h -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r6n/train_refsize_kernel_stats_f16x3_convs.csv
