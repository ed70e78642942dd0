# Whole-kernel ablations of the resident K4 kernel (timing only: the ablated results are garbage).
#   bash scripts/debug/k4_ablate.sh   (GPU box; builds a variant library with -DMMF_K4_ABLATE next to the product's)
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/multimodalfilter_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMMF_K4_ABLATE -c -o /tmp/ie_ablate.o $C/image_encoder.hip || exit 1
OBJS=$(ls $C/_obj/*.o | grep -v image_encoder.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmmf_ablate.so $OBJS /tmp/ie_ablate.o || exit 1
export MMF_LIB_PATH=/tmp/libmmf_ablate.so
for bits in ${K4_ABLATE_BITS:-0 1 2 3 4 8 16 12 28 31 32 35 60 63}; do
  echo "# MMF_K4_ABLATE=$bits (1: no conv2a, 2: no conv2b, 4: no conv3, 8: no conv4, 16: no stem, 32: no barriers; 256 x p: X waves at s_setprio p in their MFMA loop; 1024 x p: Y / Z waves at priority p)"
  MMF_K4_ABLATE=$bits K4_MODES=fused K4_SHAPES=4096x2 REPS=20 python $R/scripts/bench_k4.py 2>&1 | grep "^{"
done
