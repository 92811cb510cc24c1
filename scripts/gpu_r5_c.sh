#!/bin/bash
# round 5, call C: rasterizer backward variants -- parity suites, both scenes timed, per-wave trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -q -x 2>&1 | tail -3
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k: round(d[k],4) for k in ('render_fwd','render_bwd','preprocess_bwd','tile_sort','wall_ms_fwd_bwd')})"; }
for v in $VARIANTS; do
  echo "== parity $v"; DGM_LIB_PATH=dg-mesh_amd/lib/variants/$v.so timeout 600 python -m pytest tests/test_gpu_raster.py -m gpu -q -x -k "parity or cfg2 or long_lists or replay" 2>&1 | tail -2
done
for k in init trained; do
  echo "== default $k"; timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | pick
  for v in $VARIANTS; do
    echo "== $v $k"; DGM_LIB_PATH=dg-mesh_amd/lib/variants/$v.so timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | pick
  done
done
if [ -n "$TRACE" ]; then
echo "== trace trained"; DGM_LIB_PATH=dg-mesh_amd/lib/variants/r5_trace.so timeout 300 python tools/raster_bench.py cfg2 --kind trained --iters 10 --trace 2>&1 | tail -5 | cut -c1-600
echo "== trace init"; DGM_LIB_PATH=dg-mesh_amd/lib/variants/r5_trace.so timeout 300 python tools/raster_bench.py cfg2 --kind init --iters 10 --trace 2>&1 | tail -5 | cut -c1-600
fi
