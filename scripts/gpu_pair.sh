#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x -k "stage_by_stage or (f16x3p and (big_batch or torch_trunk))" 2>&1 | tail -3
for n in 0 112 120 128 136 144; do
  echo "pair=$n: $(DGM_MLP_PAIR=$n python tools/mlp_bench.py 100000 20 2>&1 | grep impl=)"
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pair112" -o b -- python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" 100000 20 > "$GRAFT_REPO_ROOT/gpurun_out/pair112.log" 2>&1 )
f=$(find gpurun_out/pair112 -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 23 12; find gpurun_out/pair112 -name "*kernel_trace.csv" -delete
