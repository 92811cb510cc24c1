#!/bin/bash
# bench line + rocprofv3 kernel stats of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 100 --warmup 10 > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench.json'))
print(d['value'], d['ms_per_step'], d['host_ms_per_step'])
print(d['stages_ms'])
print({k:d[k] for k in ('roofline_render_bwd','roofline_render_bwd_trained') if k in d})
PY
( cd /tmp && DGM_BENCH_STEADY_STEPS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4_prof" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 40 --warmup 5 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r4_prof.log" 2>&1 )
f=$(find gpurun_out/r4_prof -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 55 45 | tee gpurun_out/r4_bench_kernel_stats.txt; find gpurun_out/r4_prof -name "*kernel_trace.csv" -delete
