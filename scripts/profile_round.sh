# Round profile (run on the GPU box): kernel-trace stats + HBM traffic counters of the default bench, the
# EKF / f32 / philox / training variants, the other SURVEY 8d configurations and the reference-sized regimes.
#   bash scripts/profile_round.sh [out-dir under gpurun_out/]   then: python scripts/collect_profiles.py gpurun_out/<dir> profiles/rNN
# rocprofv3 writes under /tmp (a kernel trace of a long run is tens of MB; gpurun_out/ returns <= 64 MiB):
# only the summaries (stats CSVs, this repo's kernels' trace / counter rows) are copied into $OUT.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof_r03}
P=/tmp/mmf_prof
rm -rf $P; mkdir -p $OUT $P
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 $R/bench.py --steps 32 --warmup 4 $LEAN --no-f32-mode > $OUT/bench_under_rocprof.json 2> $P/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_f32 -- python3 $R/bench.py --steps 32 --warmup 4 $LEAN --precision f32 > $OUT/bench_under_rocprof_f32.json 2> $P/stats_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_ekf -- python3 $R/bench.py --workload door_ekf --steps 32 --warmup 4 $LEAN > $OUT/bench_under_rocprof_ekf.json 2> $P/stats_ekf.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_philox -- python3 $R/bench.py --steps 32 --warmup 4 $LEAN --no-f32-mode --noise philox > $OUT/bench_under_rocprof_philox.json 2> $P/stats_philox.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_$C -- python3 $R/bench.py --steps 4 --warmup 1 $LEAN --no-kernel-timers --no-f32-mode --preroll-seconds 0 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_f32_$C -- python3 $R/bench.py --steps 4 --warmup 1 $LEAN --no-kernel-timers --precision f32 --preroll-seconds 0 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_philox_$C -- python3 $R/bench.py --steps 4 --warmup 1 $LEAN --no-kernel-timers --no-f32-mode --noise philox --preroll-seconds 0 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_ekf_$C -- python3 $R/bench.py --workload door_ekf --steps 8 --warmup 0 $LEAN --no-kernel-timers --preroll-seconds 0 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_k4_$C -- python3 $R/scripts/bench_k4.py > /dev/null 2>&1
done
# training (config 5 shape) and the reference-sized regimes: kernel stats + GPU-busy fraction of the timed regions
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_train -- python3 $R/scripts/bench_train.py --steps 3 --backends hip > /dev/null 2> $P/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_ref -- python3 $R/scripts/bench_reference_sizes.py --no-cpu --eval-repeats 1 --train-iters 3 > /dev/null 2> $P/ref.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_trainref -- python3 $R/scripts/bench_reference_sizes.py --only train --backends hip --no-cpu --train-iters 10 > /dev/null 2> $P/trainref.err
cd $R
cp $(find $P/stats_trainref -name "*kernel_stats.csv" | head -1) $OUT/train_refsize_kernel_stats.csv
python scripts/gpu_busy.py $(find $P/stats_trainref -name "*kernel_trace.csv" | head -1) --gap-ms 1.0 --top 4 --kernels 12 > $OUT/train_refsize_gpu_busy.txt 2>&1
python scripts/gpu_busy.py $(find $P/stats_ref -name "*kernel_trace.csv" | head -1) --top 8 --kernels 4 > $OUT/reference_sizes_gpu_busy.txt 2>&1
python scripts/gpu_busy.py $(find $P/stats_ekf -name "*kernel_trace.csv" | head -1) --top 3 --kernels 12 > $OUT/door_ekf_gpu_busy.txt 2>&1
python scripts/gpu_busy.py $(find $P/stats -name "*kernel_trace.csv" | head -1) --top 3 --kernels 8 > $OUT/door_pf_gpu_busy.txt 2>&1
cp $(find $P/stats_train -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats.csv
cp $(find $P/stats_ref -name "*kernel_stats.csv" | head -1) $OUT/reference_sizes_kernel_stats.csv
cp $(find $P/stats_philox -name "*kernel_stats.csv" | head -1) $OUT/door_pf_philox_kernel_stats.csv
python scripts/collect_profiles.py $P $OUT/collected > $OUT/collect.log 2>&1
# plain bench lines
python bench.py > $OUT/bench_door_pf_n1.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags_door_pf.json 2>> $OUT/bench.err
python bench.py --noise philox $LEAN > $OUT/bench_door_pf_philox.json 2>> $OUT/bench.err
python bench.py --workload push_pf --no-reference-sizes > $OUT/bench_push_pf_n1.json 2>> $OUT/bench.err
python bench.py --workload door_ekf > $OUT/bench_door_ekf_n1.json 2>> $OUT/bench.err
python bench.py --workload door_ekf --steps 20 --warmup 5 $LEAN > $OUT/bench_driver_flags_door_ekf.json 2>> $OUT/bench.err
MMF_BENCH_TIMECOURSE=p,p,p,s1,p,f,p,s5,p MMF_BENCH_TIMECOURSE_STEPS=3 python bench.py --steps 20 --warmup 5 2>> $OUT/bench.err | grep "^{" > $OUT/bench_pass_timecourse_now.txt
python scripts/debug/rccl_probe.py 2>&1 | grep -E "^rank|^world|Duplicate GPU" | sort -u > $OUT/bench_rccl_probe.txt
python scripts/bench_k4.py > $OUT/bench_k4.txt 2>> $OUT/bench.err
python scripts/bench_k1.py > $OUT/bench_k1.txt 2>> $OUT/bench.err
python bench.py --workload door_pf --particles 1024 $LEAN --no-f32-mode > $OUT/bench_c2_door_pf_n256_m1024.json 2>> $OUT/bench.err
python bench.py --workload push_pf --batch 1024 --steps 64 $LEAN --no-f32-mode > $OUT/bench_c3_push_pf_n1024_m4096.json 2>> $OUT/bench.err
python bench.py --workload door_pf --batch 32 --steps 64 $LEAN --no-f32-mode > $OUT/bench_door_pf_n32_m4096.json 2>> $OUT/bench.err
python bench.py --workload door_pf --batch 1024 --steps 32 $LEAN --no-f32-mode > $OUT/bench_door_pf_n1024_m4096.json 2>> $OUT/bench.err
python bench.py --workload door_ekf --global-batch 8192 --steps 32 --warmup 4 $LEAN > $OUT/bench_c4_door_ekf_n8192_one_gpu.json 2>> $OUT/bench.err
python bench.py --workload door_pf --steps 800 $LEAN --no-f32-mode > $OUT/bench_door_pf_800_steps.json 2>> $OUT/bench.err
MMF_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 32 --warmup 8 2>> $OUT/bench.err | grep "^{" > $OUT/bench_gpus2_gloo_one_gpu.json
python scripts/bench_reference_sizes.py > $OUT/bench_reference_sizes.txt 2>> $OUT/bench.err
MMF_LOOP_GRAPH=1 python scripts/bench_reference_sizes.py --only eval --no-cpu > $OUT/bench_reference_sizes_graph.txt 2>> $OUT/bench.err
python scripts/bench_train.py > $OUT/bench_train_push_unimodal_pf.json 2>> $OUT/bench.err
python scripts/bench_train.py --backends hip --cnn-precision bf16 >> $OUT/bench_train_push_unimodal_pf.json 2>> $OUT/bench.err
./scripts/ubench/k1_phases 256 4096 > $OUT/k1_phases.txt 2>&1; ./scripts/ubench/k1_phases 256 1024 >> $OUT/k1_phases.txt 2>&1; ./scripts/ubench/k1_phases 32 300 >> $OUT/k1_phases.txt 2>&1
MMF_PRECISION=f32 python -m pytest tests -m gpu -q 2>&1 | grep -E '^E  |^FAILED|passed|failed' | tail -20 > $OUT/pytest_gpu_f32_mode.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E '^E  |^FAILED|passed|failed' | tail -20 > $OUT/pytest_gpu.txt
du -sh $OUT
