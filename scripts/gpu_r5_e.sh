#!/bin/bash
# quick A/B of library variants on both rasterizer scenes (no parity run)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k: round(d[k],4) for k in ('render_fwd','render_bwd','preprocess_bwd','tile_sort','bin_scatter','bin_count','bin_scan','wall_ms_fwd_bwd')})"; }
for k in init trained; do
  echo "== default $k"; timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | pick
  for v in $VARIANTS; do
    echo "== $v $k"; DGM_LIB_PATH=dg-mesh_amd/lib/variants/$v.so timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | pick
  done
done
