#!/bin/bash
# rasterizer parity tests + stage timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -q -x > gpurun_out/pytest_raster.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_raster.log
timeout 300 python tools/raster_bench.py cfg2 --kind init --iters 20 > gpurun_out/rb_cfg2_init.log 2>&1; tail -12 gpurun_out/rb_cfg2_init.log
timeout 300 python tools/raster_bench.py cfg2 --kind trained --iters 20 > gpurun_out/rb_cfg2_trained.log 2>&1; tail -12 gpurun_out/rb_cfg2_trained.log
