# sample GPU clock / power while the bench loops (K2-dominated workload)
python bench.py --no-cpu-baseline --no-f32-mode --no-kernel-timers --steps 3000 --warmup 16 > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
sleep 45
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo
  sleep 0.4
done
wait $BP
tail -c 400 gpurun_out/clock_bench.json
