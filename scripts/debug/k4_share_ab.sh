{ for sh in 16 17 18 19 20 16 19; do ./scripts/ubench/k4_wg_spread 4096 2 $sh b | grep "rep 3 conv2b"; done
  for sh in 16 18 19 20 16 19; do ./scripts/ubench/k4_wg_spread 4096 3 $sh b | grep "rep 3 conv2b"; done
  for sh in 16 19; do ./scripts/ubench/k4_wg_spread 1024 2 $sh b | grep "rep 3 conv2b"; done
  ./scripts/ubench/k4_wg_spread 4096 2 16 | grep -A12 "rep 3 conv2b"
  ./scripts/ubench/k4_wg_spread 4096 2 19 | grep -A12 "rep 3 conv2b"; } > gpurun_out/ubench_k4_wg_spread.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_strict.py -x -q 2>&1 | tail -3
python scripts/bench_k4.py 2>/dev/null | grep fused
