#!/bin/bash
# round 5 evidence: HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and SQ counters of the bench's kernels; SQ counters of the
# rasterizer on the trained-like scene
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -20
bash scripts/gpu_pmc_sq.sh r05 env DGM_BENCH_STEADY_STEPS=0 python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -i "render\|preprocess_bwd\|bwd_pair\|pass\|tile_s\|scatter\|count" | cut -c1-300
bash scripts/gpu_pmc_sq.sh r05_trained python $GRAFT_REPO_ROOT/tools/raster_bench.py cfg2 --kind trained --iters 15 --profile 0 2>&1 | grep -i "render_bwd4\|render_fwd\|pass" | cut -c1-300
