for c in 512 1024 2048; do
MMF_IMAGE_CHUNK=$c python bench.py --no-cpu-baseline --no-f32-mode 2>&1 | tail -1 > gpurun_out/b.json
python - <<PY
import json
j=json.load(open("gpurun_out/b.json"))
k=j["kernels"]["image_encoder"]
print("PF  chunk", $c, "ms/step", round(j["ms_per_step"],4), "K4 total ms", round(k["total_ms"],3), "launches", k["launches"])
PY
MMF_IMAGE_CHUNK=$c python bench.py --workload door_ekf --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b.json
python - <<PY
import json
j=json.load(open("gpurun_out/b.json"))
k=j["kernels"]["image_encoder"]
print("EKF chunk", $c, "ms/step", round(j["ms_per_step"],4), "K4 total ms", round(k["total_ms"],3), "launches", k["launches"])
PY
done
