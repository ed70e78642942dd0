# K2 (pipelined 64-particle tiles): weights by LDS-DMA under the first tile (default) against the synchronous copy through
# registers (a variant library built with -DMMF_K2_SYNC_STAGING).  rocprofv3 kernel durations + bench values, alternating.
#   bash scripts/debug/k2_async_staging_ab.sh   (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$R/multimodalfilter_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DMMF_K2_SYNC_STAGING -I$R/include -c -o /tmp/pn_sync.o $C/particle_net.hip || exit 1
OBJS=$(ls $C/_obj/*.o | grep -v "/particle_net.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmmf_sync.so $OBJS /tmp/pn_sync.o || exit 1
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-configs --no-f32-mode"
cat > /tmp/with_lib.py <<'PY'
import os, runpy, sys
os.environ["MMF_LIB_PATH"] = sys.argv[1]
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
PY
line() { python3 -c "
import json,sys
l=[json.loads(x) for x in sys.stdin if x.startswith('{')][-1]
print('  $1: %.4e particle-steps/s  %.2f us/step' % (l['value'], 1e3*l['ms_per_step']))"; }
for shape in 256x4096 32x4096 64x4096 256x1024; do
  N=${shape%x*}; M=${shape#*x}
  echo "# door_pf $N x $M, 64 steps"
  for rep in 1 2; do
    python3 $R/bench.py --workload door_pf --batch $N --particles $M --steps 64 --warmup 8 $LEAN 2>/dev/null | line "async (default)"
    MMF_LIB_PATH=/tmp/libmmf_sync.so python3 $R/bench.py --workload door_pf --batch $N --particles $M --steps 64 --warmup 8 $LEAN 2>/dev/null | line "sync copy     "
  done
  for v in ; do  # (kernel tables: see door_pf_kernel_stats.csv of the profile set)
    rm -rf /tmp/k2st_$v
    if [ $v = async ]; then rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2st_$v -- python3 $R/bench.py --workload door_pf --batch $N --particles $M --steps 32 --warmup 4 $LEAN --no-kernel-timers > /dev/null 2>&1
    else rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2st_$v -- python3 /tmp/with_lib.py /tmp/libmmf_sync.so $R/bench.py --workload door_pf --batch $N --particles $M --steps 32 --warmup 4 $LEAN --no-kernel-timers > /dev/null 2>&1; fi
    f=$(find /tmp/k2st_$v -name "*kernel_stats.csv" | head -1)
    echo "  kernel averages ($v):"; grep "particle_net_kernel<3" $f | awk -F'","' '{printf "    %s calls %s avg %.1f us\n", substr($1,1,90), $2, $4/1000}'
  done
done
