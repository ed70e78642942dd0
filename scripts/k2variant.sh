for v in 0 1 2 3; do
MMF_K2_VARIANT=$v python bench.py --no-cpu-baseline --no-f32-mode --steps 64 2>&1 | tail -1 > gpurun_out/b.json
python - <<PY
import json
j=json.load(open("gpurun_out/b.json"))
k=j["kernels"]
print("variant", $v, "ms/step", round(j["ms_per_step"],4), "dyn", k["particle_net_dynamics"]["avg_ms"], "meas", k["particle_net_measure"]["avg_ms"])
PY
done
