L="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-f32-mode --no-configs --steps 64 --warmup 8"
for rep in 1 2; do for v in 0 4; do MMF_K2_VARIANT=$v python bench.py $L 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels']
print('variant $v', 'ms/step %.4f'%d['ms_per_step'], 'meas us %.1f'%(1e3*k['particle_net_measure']['avg_ms']), 'dyn us %.1f'%(1e3*k['particle_net_dynamics']['avg_ms']), 'rmse', [round(x,6) for x in d['posterior_rmse_vs_truth']])
"; done; done
MMF_K2_VARIANT=4 python -m pytest tests/test_gpu_kernels.py -x -q -k "k2" 2>&1 | tail -2
