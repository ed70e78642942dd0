# Round-6 profile (run on the GPU box): kernel-trace stats + HBM traffic counters of the default bench, the EKF / f32 /
# philox variants, the reference-sized regimes (persistent loop on and off), K2's / K4's / the fused training kernels' SQ
# counters, the training step with its HBM traffic per kernel (fused against the three-pass backward), the phase clocks of
# the fused training kernel, the K1 batch sweep, and the plain bench lines (incl. --workload push_train).
#   bash scripts/profile_round_r06.sh [out-dir under gpurun_out/]
#   then: python scripts/collect_profiles.py gpurun_out/<dir> profiles/r06 ; python scripts/profiles_summary.py profiles/r06
# rocprofv3 writes under /tmp (a kernel trace of a long run is tens of MB; gpurun_out/ returns <= 64 MiB): only the
# summaries (stats CSVs, this repo's kernels' trace / counter rows) are copied into $OUT.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof_r06}
P=/tmp/mmf_prof
rm -rf $P; mkdir -p $OUT $P
HIPCC=/opt/rocm/bin/hipcc
# micro-benchmarks the round quotes (binaries are git-ignored: build them here)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/multimodalfilter_amd/csrc -o $R/scripts/ubench/k1_phases $R/scripts/ubench/k1_phases.hip > $OUT/ubench_build.log 2>&1
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-value -I$R/include -I$R/multimodalfilter_amd/csrc -o $R/scripts/ubench/fused_phases $R/scripts/ubench/fused_phases.hip >> $OUT/ubench_build.log 2>&1
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-value -DNO_PHASE_CLOCKS -I$R/include -I$R/multimodalfilter_amd/csrc -o $R/scripts/ubench/fused_phases_noclk $R/scripts/ubench/fused_phases.hip >> $OUT/ubench_build.log 2>&1
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value -I$R/include -I$R/multimodalfilter_amd/csrc -o $R/scripts/ubench/k4_wg_spread $R/scripts/ubench/k4_wg_spread.hip >> $OUT/ubench_build.log 2>&1
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -o $R/scripts/ubench/mfma_shape $R/scripts/ubench/mfma_shape.hip >> $OUT/ubench_build.log 2>&1
$HIPCC --offload-arch=gfx950 -O3 -Wno-unused-value -o $R/scripts/ubench/handoff_latency $R/scripts/ubench/handoff_latency.hip >> $OUT/ubench_build.log 2>&1
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 $R/bench.py --steps 32 --warmup 4 $LEAN --no-f32-mode > $OUT/bench_under_rocprof.json 2> $P/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_f32 -- python3 $R/bench.py --steps 32 --warmup 4 $LEAN --precision f32 > $OUT/bench_under_rocprof_f32.json 2> $P/stats_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_ekf -- python3 $R/bench.py --workload door_ekf --steps 32 --warmup 4 $LEAN > $OUT/bench_under_rocprof_ekf.json 2> $P/stats_ekf.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_philox -- python3 $R/bench.py --steps 32 --warmup 4 $LEAN --no-f32-mode --noise philox > $OUT/bench_under_rocprof_philox.json 2> $P/stats_philox.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_$C -- python3 $R/bench.py --steps 4 --warmup 1 $LEAN --no-kernel-timers --no-f32-mode --preroll-seconds 0 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_f32_$C -- python3 $R/bench.py --steps 4 --warmup 1 $LEAN --no-kernel-timers --precision f32 --preroll-seconds 0 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_philox_$C -- python3 $R/bench.py --steps 4 --warmup 1 $LEAN --no-kernel-timers --no-f32-mode --noise philox --no-calibration --preroll-seconds 0 > /dev/null 2>&1  # calibration segments draw from a noise tensor
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_ekf_$C -- python3 $R/bench.py --workload door_ekf --steps 8 --warmup 0 $LEAN --no-kernel-timers --preroll-seconds 0 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_k4_$C -- python3 $R/scripts/bench_k4.py > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_train_$C -- python3 $R/scripts/bench_train.py --steps 2 --backends hip > /dev/null 2>&1
done
# the three-pass backward (round 4) under the same counters: a child python (never an env / shell hop under rocprofv3) selects it
cat > /tmp/mmf_train_threepass.py <<'PY'
import os, runpy, sys
os.environ["MMF_TRAIN_FUSED"] = "0"
sys.argv = [sys.argv[1]] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
PY
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_train3_$C -- python3 /tmp/mmf_train_threepass.py $R/scripts/bench_train.py --steps 2 --backends hip > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_train3 -- python3 /tmp/mmf_train_threepass.py $R/scripts/bench_train.py --steps 3 --backends hip > /dev/null 2> $P/train3.err
# reference-sized regimes: kernel stats + GPU-busy fraction, the persistent loop (default) and the loop of launches
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_ref -- python3 $R/scripts/bench_reference_sizes.py --only eval --no-cpu --eval-repeats 1 > /dev/null 2> $P/ref.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_train -- python3 $R/scripts/bench_train.py --steps 3 --backends hip > /dev/null 2> $P/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_trainref -- python3 $R/scripts/bench_reference_sizes.py --only train --backends hip --no-cpu --train-iters 10 > /dev/null 2> $P/trainref.err
cd $R
cp $(find $P/stats_ref -name "*kernel_stats.csv" | head -1) $OUT/reference_sizes_kernel_stats.csv
cp $(find $P/stats_train -name "*kernel_stats.csv" | head -1) $OUT/train_kernel_stats.csv
cp $(find $P/stats_train3 -name "*kernel_stats.csv" | head -1) $OUT/train_threepass_kernel_stats.csv
python scripts/pmc_traffic_by_kernel.py $P/pmc_train_FETCH_SIZE $P/pmc_train_WRITE_SIZE "scripts/bench_train.py --steps 2 --backends hip (fused backward, default)" > $OUT/pmc_hbm_traffic_train.json 2>> $OUT/collect.log
python scripts/pmc_traffic_by_kernel.py $P/pmc_train3_FETCH_SIZE $P/pmc_train3_WRITE_SIZE "MMF_TRAIN_FUSED=0 scripts/bench_train.py --steps 2 --backends hip (three-pass backward, round 4)" > $OUT/pmc_hbm_traffic_train_threepass.json 2>> $OUT/collect.log
cp $(find $P/stats_trainref -name "*kernel_stats.csv" | head -1) $OUT/train_refsize_kernel_stats.csv
cp $(find $P/stats_philox -name "*kernel_stats.csv" | head -1) $OUT/door_pf_philox_kernel_stats.csv
python scripts/gpu_busy.py $(find $P/stats_ref -name "*kernel_trace.csv" | head -1) --top 8 --kernels 4 > $OUT/reference_sizes_gpu_busy.txt 2>&1
python scripts/gpu_busy.py $(find $P/stats_ekf -name "*kernel_trace.csv" | head -1) --top 3 --kernels 12 > $OUT/door_ekf_gpu_busy.txt 2>&1
python scripts/gpu_busy.py $(find $P/stats -name "*kernel_trace.csv" | head -1) --top 3 --kernels 8 > $OUT/door_pf_gpu_busy.txt 2>&1
python scripts/collect_profiles.py $P $OUT/collected > $OUT/collect.log 2>&1
# the bench lines below quote `roofline.traffic` from the newest committed profile: make that THIS run's counter passes (the
# box's copy of the repo; the same files go into profiles/r06 afterwards), so that a line and the file it names agree to the bit
mkdir -p $R/profiles/r06 && cp $OUT/collected/pmc_hbm_traffic*.json $R/profiles/r06/
# K2: SQ counters + effective clock of the shipped (column-half) pipeline
bash scripts/pmc_k2_r04.sh shipped > /dev/null 2>&1
cp gpurun_out/pmc_k2_r04/shipped.json $OUT/pmc_k2_sq_counters.json
# the fused training kernels: SQ counters, phase clocks, correctness of one network call
bash scripts/pmc_fused.sh final > /dev/null 2>&1
cp gpurun_out/pmc_fused/final.json $OUT/pmc_fused_sq_counters.json
{ ./scripts/ubench/fused_phases 32 8192 1; ./scripts/ubench/fused_phases 32 8192 0; ./scripts/ubench/fused_phases_noclk 32 8192 1; ./scripts/ubench/fused_phases_noclk 32 8192 0; } > $OUT/fused_phases.txt 2>&1
python scripts/debug/fused_check.py --sizes 3x40,5x7,32x30,7x300,32x8192 --quiet --timing 2>&1 | grep -v amdgpu.ids > $OUT/check_train_fused_kernel.txt
# K4: the same counters for the image-encoder kernels of the EKF bench
bash scripts/pmc_k4_r04.sh final > /dev/null 2>&1
cp gpurun_out/pmc_k4_r04/final.json $OUT/pmc_k4_sq_counters.json
bash scripts/debug/k4_ablate.sh > $OUT/k4_ablate_final.txt 2>&1
./scripts/ubench/mfma_shape > $OUT/ubench_mfma_shape.txt 2>&1
python scripts/debug/find_copies.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-reference-sizes > /dev/null 2> $OUT/find_copies.txt
bash scripts/debug/k1_cluster_ab.sh > $OUT/k1_cluster_ab_final.txt 2>&1
{ ./scripts/ubench/k4_wg_spread 4096 2 0; ./scripts/ubench/k4_wg_spread 4096 3 0; ./scripts/ubench/k4_wg_spread 1024 2 0 b; } > $OUT/ubench_k4_wg_spread.txt 2>&1
./scripts/ubench/handoff_latency > $OUT/ubench_handoff_latency.txt 2>&1
{ echo "# default (exact-fp32 image-encoder training forward)"; python scripts/bench_reference_sizes.py --only train --backends hip --no-cpu 2>&1 | grep "^{";
  echo "# MMF_K4_PRECISION=f16x3 (image-encoder training forward = the resident K4 kernel)"; MMF_K4_PRECISION=f16x3 python scripts/bench_reference_sizes.py --only train --backends hip --no-cpu 2>&1 | grep "^{";
  echo "# default again"; python scripts/bench_reference_sizes.py --only train --backends hip --no-cpu 2>&1 | grep "^{"; } > $OUT/bench_train_refsize_ab.txt
# plain bench lines (un-profiled)
python bench.py > $OUT/bench_door_pf_n1.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags_door_pf.json 2>> $OUT/bench.err
python bench.py --noise philox $LEAN > $OUT/bench_door_pf_philox.json 2>> $OUT/bench.err
python bench.py --workload push_pf --no-reference-sizes --no-configs > $OUT/bench_push_pf_n1.json 2>> $OUT/bench.err
python bench.py --workload door_ekf --no-configs > $OUT/bench_door_ekf_n1.json 2>> $OUT/bench.err
python bench.py --workload door_ekf --steps 20 --warmup 5 $LEAN > $OUT/bench_driver_flags_door_ekf.json 2>> $OUT/bench.err
python bench.py --workload door_pf_blackout $LEAN --no-f32-mode > $OUT/bench_door_pf_blackout.json 2>> $OUT/bench.err
python bench.py --workload door_ekf_blackout $LEAN > $OUT/bench_door_ekf_blackout.json 2>> $OUT/bench.err
python bench.py --workload door_pf --batch 32 --steps 64 $LEAN --no-f32-mode > $OUT/bench_door_pf_n32_m4096.json 2>> $OUT/bench.err
python bench.py --workload door_pf --batch 1024 --steps 32 $LEAN --no-f32-mode > $OUT/bench_door_pf_n1024_m4096.json 2>> $OUT/bench.err
python bench.py --workload door_ekf --global-batch 8192 --steps 32 --warmup 4 $LEAN > $OUT/bench_c4_door_ekf_n8192_one_gpu.json 2>> $OUT/bench.err
python bench.py --workload door_pf --steps 800 $LEAN --no-f32-mode > $OUT/bench_door_pf_800_steps.json 2>> $OUT/bench.err
MMF_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 32 --warmup 8 2>> $OUT/bench.err | grep "^{" > $OUT/bench_gpus2_gloo_one_gpu.json
python bench.py --workload push_train --steps 20 --warmup 5 2>> $OUT/bench.err | grep "^{" > $OUT/bench_push_train_n1.json
MMF_DIST_BACKEND=gloo python bench.py --workload push_train --gpus 2 --steps 4 --warmup 1 --batch 16 2>> $OUT/bench.err | grep "^{" > $OUT/bench_push_train_gpus2_gloo_one_gpu.json
python scripts/debug/rccl_probe.py 2>&1 | grep -E "^rank|^world|Duplicate GPU" | sort -u > $OUT/bench_rccl_probe.txt
python scripts/bench_k4.py > $OUT/bench_k4.txt 2>> $OUT/bench.err
python scripts/bench_k1.py > $OUT/bench_k1.txt 2>> $OUT/bench.err
python scripts/bench_k1.py --batch-sweep > $OUT/bench_k1_dephase_ab.txt 2>> $OUT/bench.err
# kernel-level checks of the compact training path against fp64 / the exact-fp32 kernel
# the sizes the reference itself runs: persistent loop (default) vs the loop of launches, with the stamps of one step
python scripts/bench_reference_sizes.py > $OUT/bench_reference_sizes.txt 2>> $OUT/bench.err
{ echo "# MMF_PF_PERSISTENT=1 (default)"; MMF_PERSIST_STAMPS=40 python scripts/bench_reference_sizes.py --only eval --no-cpu --eval-repeats 1 2>&1 | grep -v amdgpu.ids;
  echo "# MMF_PF_PERSISTENT=0 (one launch per kernel and step)"; MMF_PF_PERSISTENT=0 python scripts/bench_reference_sizes.py --only eval --no-cpu 2>&1 | grep -v amdgpu.ids; } > $OUT/bench_persistent_loop_ab.txt
python scripts/bench_train.py > $OUT/bench_train_push_unimodal_pf.json 2>> $OUT/bench.err
python scripts/bench_train.py --backends hip --cnn-precision bf16 >> $OUT/bench_train_push_unimodal_pf.json 2>> $OUT/bench.err
{ echo "# default: recompute + backward + weight gradients of a network call as ONE kernel (MMF_TRAIN_FUSED=1)"; python scripts/bench_train.py --backends hip 2>&1 | grep "^{";
  echo "# MMF_TRAIN_FUSED=0: the cross-check form, three passes of exact fp32 products over f16 recompute buffers"; MMF_TRAIN_FUSED=0 python scripts/bench_train.py --backends hip 2>&1 | grep "^{";
  echo "# default again"; python scripts/bench_train.py --backends hip 2>&1 | grep "^{"; } > $OUT/bench_train_fused_ab.txt
./scripts/ubench/k1_phases 256 4096 > $OUT/k1_phases.txt 2>&1; ./scripts/ubench/k1_phases 256 1024 >> $OUT/k1_phases.txt 2>&1; ./scripts/ubench/k1_phases 32 300 >> $OUT/k1_phases.txt 2>&1
MMF_PRECISION=f32 python -m pytest tests -m gpu -q 2>&1 | grep -E '^E  |^FAILED|passed|failed' | tail -20 > $OUT/pytest_gpu_f32_mode.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E '^E  |^FAILED|passed|failed' | tail -20 > $OUT/pytest_gpu.txt
du -sh $OUT
