#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/g7
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x -k "stage_by_stage or (f16x3p and big_batch)" 2>&1 | tail -2
for m in 1 2 3; do
  ( cd /tmp && DGM_P4_G7_MULT=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/g7/m$m" -o b -- python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" 100000 20 > "$GRAFT_REPO_ROOT/gpurun_out/g7/m$m.log" 2>&1 )
  echo "== mult $m: $(grep impl= gpurun_out/g7/m$m.log)"
  f=$(find gpurun_out/g7/m$m -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 20 12 | grep -E "Li1ELi128|dw4_kernel<8, 1|Li8ELi1E"
  find gpurun_out/g7/m$m -name "*kernel_trace.csv" -delete
done
echo "mesh side=0: $(DGM_SIDE_STREAM=0 python bench.py --phase mesh --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep '^{' | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])')"
echo "mesh side=1: $(DGM_SIDE_STREAM=1 python bench.py --phase mesh --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep '^{' | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])')"
