# Round profile: kernel-trace stats + HBM traffic counters of the default bench (run on the GPU box)
R=/root/repo
OUT=$R/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-f32-mode > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-f32-mode > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-f32-mode > /dev/null 2>&1
cd $R
python bench.py > $OUT/bench_door_pf_n1.json 2> $OUT/bench.err
python bench.py --workload push_pf > $OUT/bench_push_pf_n1.json 2>> $OUT/bench.err
python bench.py --workload door_ekf > $OUT/bench_door_ekf_n1.json 2>> $OUT/bench.err
find $OUT -name "*.csv" | head -20
# other SURVEY 8d configurations, for the record (C2: door PF N=256 M=1024; C3: push PF N=1024 M=4096; reference-sized eval)
python bench.py --workload door_pf --particles 1024 --no-f32-mode --no-cpu-baseline > $OUT/bench_c2_door_pf_n256_m1024.json 2>> $OUT/bench.err
python bench.py --workload push_pf --batch 1024 --steps 64 --no-f32-mode --no-cpu-baseline > $OUT/bench_c3_push_pf_n1024_m4096.json 2>> $OUT/bench.err
python bench.py --workload door_pf --batch 32 --particles 300 --steps 200 --no-f32-mode > $OUT/bench_door_pf_n32_m300.json 2>> $OUT/bench.err
MMF_PRECISION=f32 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > $OUT/pytest_gpu_f32_mode.txt
# training step (K6 vs torch autograd), SURVEY 8d config C5 shape scaled to N*M = 2^18 per step
python scripts/bench_train.py > $OUT/bench_train_push_unimodal_pf.json 2>> $OUT/bench.err
