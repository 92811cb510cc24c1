#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mlp.py tests/test_trainer_dp_gpu.py -m gpu -q -k "stage_by_stage or goldens or mesh_phase" 2>&1 | tail -3
bash scripts/gpu_kt.sh 100000 r3o_kt 2>&1 | grep -E "libdgmesh|embed4|prep4|gemm4_kernel<16, 1024, 512, [01]"
bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_vs_reference.py -m gpu -q -k "knn or timed" -s 2>&1 | tail -5
ls gpurun_out/*.json | head -30
