#!/bin/bash
# round 3, call A: plane-format MLP (f16x3p) -- stage-by-stage diagnostic, parity tests of the mode, micro benchmark, bench + kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x -s -k "stage_by_stage" > gpurun_out/r3a_stage.log 2>&1; echo "stage exit $?"; grep -E "f16x3p stages|stages off|Error|error|passed|failed" gpurun_out/r3a_stage.log | cut -c1-1500 | head -20
timeout 900 python -m pytest tests/test_mlp.py -m gpu -q -k "f16x3p or cfg4" > gpurun_out/r3a_mlp.log 2>&1; echo "mlp exit $?"; tail -12 gpurun_out/r3a_mlp.log
for m in f16x3p f16x3; do DGM_MLP_GEMM=$m timeout 300 python tools/mlp_bench.py 100000 20; done
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench exit $?"; tail -3 gpurun_out/r3a_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3a_bench.json')); print(d['value'], d['ms_per_step'], d['stages_ms'], d['host_ms_per_step'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r3a_prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r3a_prof_run.log" 2>&1; echo "rocprof exit $?"
cd "$GRAFT_REPO_ROOT"; f=$(find gpurun_out/r3a_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" | cut -c1-200
find gpurun_out/r3a_prof -name "*kernel_trace.csv" -size +20M -delete
