set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "training_step_matches_oracle" 2>&1 | tail -3 > gpurun_out/trainref_tests.txt
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "on_a_shard" 2>&1 | tail -3 >> gpurun_out/trainref_tests.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o tr -- python3 $GRAFT_REPO_ROOT/scripts/bench_reference_sizes.py --only train --backends hip --no-cpu --train-iters 10 > $GRAFT_REPO_ROOT/gpurun_out/trainref_bench.txt 2>&1
cd $GRAFT_REPO_ROOT
T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
S=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
python scripts/gpu_busy.py $T --gap-ms 1.0 --top 14 --kernels 40 > gpurun_out/trainref_busy.txt
cp $S gpurun_out/trainref_kernel_stats.csv
