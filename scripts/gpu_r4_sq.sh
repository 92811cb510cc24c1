#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_pmc_sq.sh r4_trained python /root/repo/tools/raster_bench.py cfg2 --kind trained --iters 15 --profile 0 2>&1 | grep -i "render_bwd4\|render_fwd\|pass"
