#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x -k "stage_by_stage" 2>&1 | tail -2
bash tools/mlp_kernel_times.sh ${1:-100000} ${2:-kt} 2>&1 | grep -E "^==|gemm4_kernel|dw4_kernel<8, 8"
[ -f dg-mesh_amd/lib/variants/timing.so ] && DGM_LIB_PATH=$GRAFT_REPO_ROOT/dg-mesh_amd/lib/variants/timing.so python tools/mlp_bench.py 100000 10 | tail -10
