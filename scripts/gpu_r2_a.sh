#!/bin/bash
# round 2, first GPU call: new parity tests + bench + SQ counter passes on the round-1 kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench exit $?"; tail -3 gpurun_out/bench_a.err; cat gpurun_out/bench_a.json
bash scripts/gpu_pmc_sq.sh r02a python bench.py --steps 4 --warmup 1 --no-cpu-baseline
