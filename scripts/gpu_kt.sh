#!/bin/bash
# per-kernel times of the MLP micro benchmark for the product library and all variants (plus the stage test as a guard)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x -k "stage_by_stage" 2>&1 | tail -2
bash tools/mlp_kernel_times.sh ${1:-100000} ${2:-kt} 2>&1 | grep -E "^==|gemm4_kernel|dw4_kernel|reduce_dw|embed4"
