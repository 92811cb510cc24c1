#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
p() { python -c 'import json,sys; r=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(r["value"],1), round(r["ms_per_step"],3), r["host_ms_per_step"], "one_stream:", {k:v for k,v in r.get("one_stream",{}).items() if k!="avg_ms"}, "f32:", r.get("mlp_f32_mode",{}).get("value"))'; }
echo "side=0 200 steps: $(DGM_SIDE_STREAM=0 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | p)"
echo "side=1 200 steps: $(DGM_SIDE_STREAM=1 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | p)"
echo "side=1 full: $(python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | p)"
echo "side=0 full: $(DGM_SIDE_STREAM=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | p)"
