#!/bin/bash
# round 4, call B: render_bwd4 (MFMA reduction, 64-entry units, exact culling, 36-byte rows + live flags): parity, then A/B timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -15
for k in init trained; do
  echo "== new $k"; timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | tee gpurun_out/r4_b_new_$k.json
  echo "== base $k"; DGM_ABI_ANY=1 DGM_LIB_PATH=dg-mesh_amd/lib/variants/r3_base.so timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | tee gpurun_out/r4_b_base_$k.json
  for v in dg-mesh_amd/lib/variants/r4_*.so; do
    [ -f "$v" ] || continue
    echo "== $v $k"; DGM_LIB_PATH=$v timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | tee gpurun_out/r4_b_$(basename $v .so)_$k.json
  done
done
