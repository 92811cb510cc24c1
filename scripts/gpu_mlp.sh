#!/bin/bash
# MLP parity tests + micro benchmark (both GEMM arithmetics) + kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mlp.py -m gpu -q -x -s > gpurun_out/pytest_mlp.log 2>&1; echo "pytest exit $?"; grep -E "max rel|passed|failed|Error|error" gpurun_out/pytest_mlp.log | tail -15
timeout 300 python tools/mlp_bench.py 100000 20 hip 2>&1 | tail -1
DGM_MLP_GEMM=f32 timeout 300 python tools/mlp_bench.py 100000 20 hip 2>&1 | tail -1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_mlp" -o mlp -- python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" 100000 10 hip > "$GRAFT_REPO_ROOT/gpurun_out/prof_mlp.log" 2>&1
cd "$GRAFT_REPO_ROOT"; python tools/prof_summary.py gpurun_out/prof_mlp/mlp_kernel_stats.csv 2>&1 | head -24
find gpurun_out/prof_mlp -name "*kernel_trace.csv" -size +20M -delete
