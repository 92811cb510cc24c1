#!/bin/bash
# f16x3 MLP: parity tests + micro benchmark in the three arithmetics + stage timings from the bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mlp.py -m gpu -q -x > gpurun_out/pytest_mlp.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_mlp.log
for m in f16x3 bf16x6; do DGM_MLP_GEMM=$m timeout 300 python tools/mlp_bench.py 100000 20; done
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; echo "bench exit $?"; tail -3 gpurun_out/bench_b.err; python -c "
import json; d=json.load(open('gpurun_out/bench_b.json')); print(d['value'], d['stages_ms'], d['host_ms_per_step'])"
