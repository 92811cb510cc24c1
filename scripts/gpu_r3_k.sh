#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in f16x3p f16x3; do
DGM_MLP_GEMM=$m timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r3k_bench_$m.json 2> gpurun_out/r3k_bench_$m.err; echo "bench $m exit $?"; python -c "
import json; d=json.load(open('gpurun_out/r3k_bench_$m.json')); print('$m', round(d['value'],1), round(d['ms_per_step'],3), d['stages_ms'], d['host_ms_per_step'])"
done
timeout 900 python -m pytest tests/test_mlp.py -m gpu -q 2>&1 | tail -5
