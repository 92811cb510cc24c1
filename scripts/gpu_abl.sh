#!/bin/bash
# ablation timings of the forward GEMM (profiling aid; results of ablated runs are wrong by construction)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for a in 0 1 2 3 4; do
  (cd /tmp && DGM_MLP_ABL=$a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/abl$a" -o m -- python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" 100000 5 hip > /dev/null 2>&1)
  echo "ABL=$a"; python tools/prof_summary.py gpurun_out/abl$a/m_kernel_stats.csv 2>&1 | grep -E "gemm6_kernel<0|dw6" | cut -c1-60,75-120
  rm -f gpurun_out/abl$a/m_kernel_trace.csv
done
