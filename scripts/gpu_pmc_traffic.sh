#!/bin/bash
# FETCH_SIZE and WRITE_SIZE in separate passes over the bench -> gpurun_out/pmc_traffic/*.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c" -o pmc -- env DGM_BENCH_STEADY_STEPS=0 python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log" 2>&1; echo "$c: rocprof exit $?" )
done
F=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" gpurun_out/pmc_traffic cfg2
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -delete
