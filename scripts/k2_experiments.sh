# What bounds K2 (particle_net_kernel, f16x3, pipelined)?  Builds variant libraries with one cost
# removed each -- the third MFMA of every product, the range tracking, the residual half of the
# operand split -- and times the measurement kernel in the default bench (results of the variants
# are WRONG by construction; only their time and SQ counters are of interest).
#   bash scripts/k2_experiments.sh build     (build container)      -> gpurun_out/k2exp/libmmf_*.so
#   bash scripts/k2_experiments.sh run       (GPU box)              -> gpurun_out/k2exp/*.json
R=$(cd "$(dirname "$0")/.." && pwd)
LIBS=$R/scripts/ubench/k2exp   # variant libraries travel with the snapshot (git-ignored)
OUT=$R/gpurun_out/k2exp
mkdir -p $OUT $LIBS
VARIANTS="BASE MMF_EXP_TWO_PRODUCTS MMF_EXP_NO_RANGE MMF_EXP_NO_SPLIT_LO"
if [ "$1" = "build" ]; then
  for v in $VARIANTS; do
    objs=""
    for src in abi pf_resample pf_init ekf ukf particle_net image_encoder traj_program pf_loop ekf_loop; do
      if [ $src = particle_net ]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -D$v -c -o $LIBS/pn_$v.o $R/multimodalfilter_amd/csrc/particle_net.hip || exit 1
        objs="$objs $LIBS/pn_$v.o"
      else
        objs="$objs $R/multimodalfilter_amd/csrc/_obj/$src.o"
      fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIBS/libmmf_$v.so $objs || exit 1
  done
  ls -la $LIBS/*.so
else
  cd /tmp && export TMPDIR=/tmp
  for v in $VARIANTS; do
    MMF_LIB_PATH=$LIBS/libmmf_$v.so python3 $R/bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-f32-mode --no-precision-study 2>/dev/null | grep "^{" > $OUT/bench_$v.json
    MMF_LIB_PATH=$LIBS/libmmf_$v.so rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$v -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-f32-mode --no-precision-study > /dev/null 2>&1
  done
  python3 - <<PY
import json, csv, glob, collections
for v in "$VARIANTS".split():
    d = json.loads(open("$OUT/bench_%s.json" % v).read())
    k = d["kernels"]
    c = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "particle_net_kernel<3, 2, 1, 2, 1, 2, true>" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    cs = {n: round(sum(x) / len(x) / 1e6, 2) for n, x in sorted(c.items())}
    print(json.dumps({"variant": v, "ms_per_step": round(d["ms_per_step"], 4), "measure_us": round(1e3 * k["particle_net_measure"]["avg_ms"], 1),
                      "dynamics_us": round(1e3 * k["particle_net_dynamics"]["avg_ms"], 1), "sq_counters_millions": cs}))
PY
fi
