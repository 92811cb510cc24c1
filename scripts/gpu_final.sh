#!/bin/bash
# PMC traffic passes + the bench line (with CPU baseline) -> gpurun_out/
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_pmc_traffic.sh
cp gpurun_out/pmc_traffic/*.json profiles/ 2>/dev/null   # bench.py reads the committed location
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench exit $?"; tail -2 gpurun_out/bench_final.err; cat gpurun_out/bench_final.json
