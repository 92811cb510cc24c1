#!/bin/bash
# A/B of library variants (dg-mesh_amd/lib/variants/*.so) on both rasterizer scenes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in init trained; do
  echo "== default $k"; timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1
  for v in dg-mesh_amd/lib/variants/r4_*.so; do
    [ -f "$v" ] || continue
    echo "== $(basename $v .so) $k"; DGM_LIB_PATH=$v timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1
  done
done
if [ -n "$SQ_TRAINED" ]; then
  bash scripts/gpu_pmc_sq.sh r4_trained python /root/repo/tools/raster_bench.py cfg2 --kind trained --iters 15 --profile 0 2>&1 | grep -i "render_bwd4\|render_fwd\|pass"
fi
