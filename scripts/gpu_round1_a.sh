#!/bin/bash
# first GPU contact: smoke, parity tests, stage timings, rocprof kernel stats
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Name:|Compute Unit" | head -8
nproc
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -5 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -40 gpurun_out/pytest_gpu.log
timeout 300 python tools/raster_bench.py cfg2 --kind init --iters 20 > gpurun_out/rb_cfg2_init.log 2>&1; tail -3 gpurun_out/rb_cfg2_init.log
timeout 300 python tools/raster_bench.py cfg2 --kind trained --iters 20 > gpurun_out/rb_cfg2_trained.log 2>&1; tail -3 gpurun_out/rb_cfg2_trained.log
