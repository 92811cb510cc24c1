cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for z in 0 1; do
( cd /tmp && DGM_BENCH_ZERO=$z timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/zero$z -o b -- python $GRAFT_REPO_ROOT/tools/mlp_bench.py 100000 20 > $GRAFT_REPO_ROOT/gpurun_out/zero$z.log 2>&1 )
echo "== zero=$z: $(grep impl= gpurun_out/zero$z.log)"
f=$(find gpurun_out/zero$z -name "*kernel_stats.csv" | head -1); head -8 $f | cut -d, -f1,2,4 | cut -c1-120
find gpurun_out/zero$z -name "*kernel_trace.csv" -delete
done
