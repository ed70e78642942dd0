# Round profile: kernel-trace stats + HBM traffic counters of the default bench (run on the GPU box)
#   bash scripts/profile_round.sh [out-dir under gpurun_out/]      then: python scripts/collect_profiles.py gpurun_out/<dir> profiles/rNN
R=/root/repo
OUT=$R/gpurun_out/${1:-prof_r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-f32-mode --no-precision-study > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -- python3 $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --precision f32 --no-precision-study > $OUT/bench_under_rocprof_f32.json 2> $OUT/stats_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_ekf -- python3 $R/bench.py --workload door_ekf --steps 32 --warmup 4 --no-cpu-baseline --no-precision-study > $OUT/bench_under_rocprof_ekf.json 2> $OUT/stats_ekf.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-f32-mode --no-precision-study > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-f32-mode --no-precision-study > /dev/null 2>&1
# image encoder alone, 2048 images x 2 encoders per launch sequence: traffic per image-encoder
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_k4_FETCH_SIZE -- python3 $R/scripts/bench_k4.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_k4_WRITE_SIZE -- python3 $R/scripts/bench_k4.py > /dev/null 2>&1
cd $R
python bench.py > $OUT/bench_door_pf_n1.json 2> $OUT/bench.err
python bench.py --workload push_pf > $OUT/bench_push_pf_n1.json 2>> $OUT/bench.err
python bench.py --workload door_ekf > $OUT/bench_door_ekf_n1.json 2>> $OUT/bench.err
python scripts/bench_k4.py > $OUT/bench_k4.txt 2>> $OUT/bench.err
python scripts/bench_k1.py > $OUT/bench_k1.txt 2>> $OUT/bench.err
# other SURVEY 8d configurations, for the record (C2: door PF N=256 M=1024; C3: push PF N=1024 M=4096; reference-sized eval)
python bench.py --workload door_pf --particles 1024 --no-f32-mode --no-cpu-baseline --no-precision-study > $OUT/bench_c2_door_pf_n256_m1024.json 2>> $OUT/bench.err
python bench.py --workload push_pf --batch 1024 --steps 64 --no-f32-mode --no-cpu-baseline --no-precision-study > $OUT/bench_c3_push_pf_n1024_m4096.json 2>> $OUT/bench.err
# SURVEY 8d's headline trio is door PF at M=4096 with N in {32, 256, 1024}: the default bench is N=256
python bench.py --workload door_pf --batch 32 --steps 64 --no-f32-mode --no-cpu-baseline --no-precision-study > $OUT/bench_door_pf_n32_m4096.json 2>> $OUT/bench.err
python bench.py --workload door_pf --batch 1024 --steps 32 --no-f32-mode --no-cpu-baseline --no-precision-study > $OUT/bench_door_pf_n1024_m4096.json 2>> $OUT/bench.err
# C4's global batch (8192 EKF trajectories) on the one GPU: the strong-scaling switch at N = 1
python bench.py --workload door_ekf --global-batch 8192 --steps 32 --warmup 4 --no-cpu-baseline --no-precision-study > $OUT/bench_c4_door_ekf_n8192_one_gpu.json 2>> $OUT/bench.err
python bench.py --workload door_pf --batch 32 --particles 300 --steps 200 --no-f32-mode --no-precision-study > $OUT/bench_door_pf_n32_m300.json 2>> $OUT/bench.err
# the N > 1 launcher on one GPU (two gloo ranks sharing it): plumbing evidence, not a scaling number
MMF_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 32 --warmup 8 --no-f32-mode --no-precision-study 2>> $OUT/bench.err | grep "^{" > $OUT/bench_gpus2_gloo_one_gpu.json
MMF_PRECISION=f32 python -m pytest tests -m gpu -q 2>&1 | grep -E '^E  |^FAILED|passed|failed' | tail -20 > $OUT/pytest_gpu_f32_mode.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E '^E  |^FAILED|passed|failed' | tail -20 > $OUT/pytest_gpu.txt
# training step (K6 vs torch autograd), SURVEY 8d config C5 shape scaled to N*M = 2^18 per step
python scripts/bench_train.py > $OUT/bench_train_push_unimodal_pf.json 2>> $OUT/bench.err
# ... and config C5 as specified: bf16 measurement CNN in the training forward
python scripts/bench_train.py --backends hip --cnn-precision bf16 >> $OUT/bench_train_push_unimodal_pf.json 2>> $OUT/bench.err
find $OUT -name "*.csv" | head -30
