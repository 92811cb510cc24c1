#!/bin/bash
# rasterizer parity + per-stage timings (init-like and trained-like cfg2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -q -x > gpurun_out/pytest_raster.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_raster.log
timeout 300 python tools/raster_bench.py cfg2 --kind init 2>&1 | tail -4
timeout 300 python tools/raster_bench.py cfg2 --kind trained 2>&1 | tail -4
