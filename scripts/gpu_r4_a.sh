#!/bin/bash
# round 4, call A: VALU instruction costs + the rasterizer baseline on this box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 tools/bin/valu_micro > gpurun_out/r4_valu_micro.txt 2>&1
for k in init trained; do
  timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 > gpurun_out/r4_a_raster_$k.json
done
cat gpurun_out/r4_valu_micro.txt
cat gpurun_out/r4_a_raster_*.json
