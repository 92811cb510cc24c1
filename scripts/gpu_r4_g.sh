#!/bin/bash
# round 4, call H: gradient-row design matrix (stride 9 | 12, dead rows written or not, gather through LDS | per thread)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -5
for k in init trained; do
  for p in 0; do
    echo "== perm$p $k"; DGM_TILE_PERM=$p timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | tee gpurun_out/r4_g_perm${p}_$k.json
  done
  for v in dg-mesh_amd/lib/variants/r4_*.so; do
    [ -f "$v" ] || continue
    echo "== $(basename $v .so) $k"; DGM_LIB_PATH=$v timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | tee gpurun_out/r4_h_$(basename $v .so)_$k.json
  done
done
