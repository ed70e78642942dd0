for c in 256 512 1024 2048 4096; do
MMF_IMAGE_CHUNK=$c python bench.py --no-cpu-baseline --no-f32-mode --steps 64 2>&1 | tail -1 > gpurun_out/b.json
python - <<PY
import json
j=json.load(open("gpurun_out/b.json"))
k=j["kernels"]["image_encoder"]
print("chunk", $c, "ms/step", round(j["ms_per_step"],4), "K4 total ms", k["total_ms"], "launches", k["launches"])
PY
done
