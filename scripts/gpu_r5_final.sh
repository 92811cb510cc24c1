#!/bin/bash
# round 5 final evidence: smoke, the whole gpu suite, the bench line with extras, rocprofv3 kernel stats of the same bench, cfg1 / cfg5 lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-r05_c}
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench.json'))
print(d['value'], d['ms_per_step'], d['host_ms_per_step'])
print(d['stages_ms'])
print('roofline', {k: d['roofline'].get(k) for k in ('kernel','frac','avg_ms','frac_mfma_pipe')})
print('trained', {k: d.get('roofline_render_bwd_trained',{}).get(k) for k in ('avg_ms','frac','render_fwd_ms')}, d.get('roofline_render_bwd_trained',{}).get('group_with_preprocess_bwd'))
print('rb', d['roofline_render_bwd']['avg_ms'], d['roofline_render_bwd']['frac'], 'densify', (d.get('with_densify') or {}).get('value'), 'f32', (d.get('mlp_f32_mode') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
print('pre_bwd', d['kernels'].get('preprocess_bwd'))
PY
( cd /tmp && DGM_BENCH_STEADY_STEPS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log" 2>&1 )
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 1 50 > gpurun_out/${TAG}_bench_kernel_stats.txt; head -30 gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-120; find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete
for w in cfg1 cfg4 cfg5; do
  timeout 600 python bench.py --workload $w --steps 40 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_$w.json')); print('$w', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', d['host_ms_per_step'])"
done
