#!/bin/bash
# round 4, call K: dX of the fused trunk; mesh-phase trainer on it; mesh bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mlp.py -m gpu -x -q -k "input_gradient or stage_by_stage or refuses" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_trainer_dp_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --workload cfg5 --phase mesh --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 mesh', round(d['value'],1), 'it/s', round(d['ms_per_step'],2), 'ms')"
( cd /tmp && DGM_BENCH_STEADY_STEPS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4_mesh" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --workload cfg5 --phase mesh --steps 10 --warmup 3 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r4_mesh.log" 2>&1 )
f=$(find gpurun_out/r4_mesh -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04_cfg5_mesh_kernel_stats.csv; python tools/prof_summary.py $f 23 25 | tee gpurun_out/r04_cfg5_mesh_kernel_stats.txt; find gpurun_out/r4_mesh -name "*kernel_trace.csv" -delete
