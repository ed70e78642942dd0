cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 $GRAFT_REPO_ROOT/scripts/bench_reference_sizes.py --only train --backends hip --no-cpu --train-iters 10 > /dev/null 2> /tmp/pt.err
cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/trainref_stats.csv
