#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x -s -k "stage_by_stage" > gpurun_out/r3b_stage.log 2>&1; echo "stage exit $?"; grep -E "f16x3p stages|stages off|Error|error|passed|failed" gpurun_out/r3b_stage.log | cut -c1-1200 | head -20
timeout 900 python -m pytest tests/test_mlp.py -m gpu -q -k "f16x3p or cfg4" > gpurun_out/r3b_mlp.log 2>&1; echo "mlp exit $?"; tail -5 gpurun_out/r3b_mlp.log
bash tools/mlp_kernel_times.sh 100000 r3b_kt
