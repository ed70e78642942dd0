# SQ counters for the K2 measurement kernel, pipelined (variant 0) vs unpipelined (variant 3)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for v in 0 3; do
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | tr ' ' '+')
  MMF_K2_VARIANT=$v rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_k2/v${v}_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-f32-mode > /dev/null 2>&1
done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for v in (0, 3):
    tot = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_k2/v{v}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "particle_net_kernel" in row["Kernel_Name"] and "Li1ELi2ELi1ELi2" in row["Kernel_Name"].replace(" ", "").replace("<","I").replace(">","E").replace(",","EL") or ("particle_net_kernel<3, 2, 1, 2, 1, 2" in row["Kernel_Name"]):
                tot[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("variant", v, {k: round(sum(x) / len(x)) for k, x in sorted(tot.items())}, "launches", {k: len(x) for k, x in tot.items()})
PY
