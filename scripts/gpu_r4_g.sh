#!/bin/bash
# A/B of render_bwd4 variants on both scenes (+ parity of the default build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -3
for k in init trained; do
  echo "== default $k"; timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1
  for v in dg-mesh_amd/lib/variants/r4_*.so; do
    [ -f "$v" ] || continue
    echo "== $(basename $v .so) $k"; DGM_LIB_PATH=$v timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1
  done
done
