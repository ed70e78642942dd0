R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6j
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tr_new /tmp/tr_old
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_new -o tr -- python3 $R/scripts/bench_reference_sizes.py --only train --backends hip --no-cpu --train-iters 10 > /dev/null 2>&1
cat > /tmp/old.py <<PY
import os, runpy, sys
os.environ["MMF_K4_PRECISION"] = "f32"
sys.argv = ["$R/scripts/bench_reference_sizes.py", "--only", "train", "--backends", "hip", "--no-cpu", "--train-iters", "10"]
runpy.run_path(sys.argv[0], run_name="__main__")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_old -o tr -- python3 /tmp/old.py > /dev/null 2>&1
cp $(find /tmp/tr_new -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r6j/train_refsize_kernel_stats_resident_fwd.csv
cp $(find /tmp/tr_old -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r6j/train_refsize_kernel_stats_f32_fwd.csv
cd $R
python scripts/gpu_busy.py $(find /tmp/tr_new -name "*kernel_trace.csv" | head -1) --gap-ms 1.0 --top 10 --kernels 12 > $R/gpurun_out/r6j/train_refsize_busy_resident_fwd.txt 2>&1
python scripts/gpu_busy.py $(find /tmp/tr_old -name "*kernel_trace.csv" | head -1) --gap-ms 1.0 --top 10 --kernels 12 > $R/gpurun_out/r6j/train_refsize_busy_f32_fwd.txt 2>&1
for i in 1 2 3; do
python scripts/bench_reference_sizes.py --only train --backends hip --no-cpu 2>&1 | grep "^{" | cut -c1-220
MMF_K4_PRECISION=f32 python scripts/bench_reference_sizes.py --only train --backends hip --no-cpu 2>&1 | grep "^{" | cut -c1-220
done > $R/gpurun_out/r6j/train_refsize_wall_ab.txt
