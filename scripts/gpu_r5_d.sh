#!/bin/bash
# round 5, call D: the whole gpu suite, the bench line with extras, rocprofv3 kernel stats of the same bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-r05_b}
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench.json'))
print(d['value'], d['ms_per_step'], d['host_ms_per_step'])
print(d['stages_ms'])
print('trained', {k: d.get('roofline_render_bwd_trained',{}).get(k) for k in ('avg_ms','frac','render_fwd_ms')}, d.get('roofline_render_bwd_trained',{}).get('group_with_preprocess_bwd'))
print('rb', d['roofline_render_bwd']['avg_ms'], d['roofline_render_bwd']['frac'], 'densify', (d.get('with_densify') or {}).get('value'), 'f32', (d.get('mlp_f32_mode') or {}).get('value'))
PY
( cd /tmp && DGM_BENCH_STEADY_STEPS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log" 2>&1 )
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 1 50 > gpurun_out/${TAG}_bench_kernel_stats.txt; head -24 gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-120; find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete
