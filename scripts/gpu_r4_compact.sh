#!/bin/bash
# compact backward grid: raster parity suites, both scenes timed, per-wave trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py tests/test_reference_render.py -m gpu -x -q 2>&1 | tail -4
for k in init trained; do
  echo "== default $k"; timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1
  echo "== trace $k"; DGM_LIB_PATH=dg-mesh_amd/lib/variants/r4_trace.so timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 3 --trace 2>&1 | tail -5
done
