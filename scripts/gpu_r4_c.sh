#!/bin/bash
# round 4, call D: render_bwd4 with 16-byte row accesses: parity, timing, SQ counters of the blend kernels on both scenes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -5
for k in init trained; do
  echo "== new $k"; timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | tee gpurun_out/r4_c_new_$k.json
  for v in dg-mesh_amd/lib/variants/r4_*.so; do
    [ -f "$v" ] || continue
    echo "== $v $k"; DGM_LIB_PATH=$v timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | tee gpurun_out/r4_c_$(basename $v .so)_$k.json
  done
done
for k in init trained; do
  bash scripts/gpu_pmc_sq.sh r4_$k python tools/raster_bench.py cfg2 --kind $k --iters 12 --profile 0 2>&1 | grep -i "render\|preprocess_bwd\|pass"
done
