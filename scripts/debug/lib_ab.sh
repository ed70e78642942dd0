# A/B of two builds of the library on one box: multimodalfilter_amd/libmmf_hip_prev.so (a build of the previous commit) against
# the in-tree one, alternating.  bash scripts/debug/lib_ab.sh "<bench.py arguments>" [repetitions]
ARGS=${1:---steps 128}
for rep in $(seq 1 ${2:-3}); do
  for lib in prev new; do
    if [ $lib = prev ]; then export MMF_LIB_PATH=$GRAFT_REPO_ROOT/multimodalfilter_amd/libmmf_hip_prev.so; else unset MMF_LIB_PATH; fi
    python bench.py $ARGS --no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs --no-f32-mode 2>/dev/null | grep "^{" | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$lib', '$ARGS', j['value'], j['ms_per_step'], j.get('kernels_ms'))"
  done
done
