#!/bin/bash
# round 6 final evidence: smoke, the whole gpu suite, the driver's bench command, the 200-step bench with extras and the CPU leg,
# rocprofv3 kernel stats of the same bench, cfg1 (with its CPU leg) / cfg4 / cfg5 lines, the raster micro-bench on both scenes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-r06_g}
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench20.json 2> gpurun_out/${TAG}_bench20.err; echo "bench20 exit $?"
timeout 900 python bench.py --steps 200 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
python - <<PY
import json
for f in ("${TAG}_bench20", "${TAG}_bench"):
    d=json.load(open(f'gpurun_out/{f}.json'))
    print(f, round(d['value'],2), round(d['ms_per_step'],3), 'R', d['config']['num_rendered'], d['host_ms_per_step'], 'steady', d.get('steady_state'))
    print('   roofline', {k: d['roofline'].get(k) for k in ('kernel','frac','avg_ms','frac_mfma_pipe','traffic','traffic_source')})
    print('   rb', d['roofline_render_bwd']['avg_ms'], d['roofline_render_bwd']['frac'], 'densify', (d.get('with_densify') or {}).get('value'), 'f32', (d.get('mlp_f32_mode') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('num_rendered'))
    t = d.get('roofline_render_bwd_trained') or {}
    print('   trained', {k: t.get(k) for k in ('avg_ms','frac','render_fwd_ms')}, t.get('group_with_preprocess_bwd'), 'frac_valu', {k: v.get('frac_valu') for k, v in (d.get('frac_valu') or {}).items()})
    print('   stages', d['stages_ms'])
PY
( cd /tmp && DGM_BENCH_STEADY_STEPS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log" 2>&1 )
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 1 50 200 > gpurun_out/${TAG}_bench_kernel_stats.txt; head -34 gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-120; find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" -delete
timeout 600 python bench.py --workload cfg1 --steps 200 --no-extras > gpurun_out/${TAG}_bench_cfg1.json 2>/dev/null
for w in cfg4 cfg5; do
  timeout 600 python bench.py --workload $w --steps 40 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/${TAG}_bench_$w.json 2>/dev/null
done
for w in cfg1 cfg4 cfg5; do
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_$w.json')); print('$w', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms R', d['config']['num_rendered'], d['host_ms_per_step'], (d.get('cpu_baseline') or {}).get('value'))"
done
for k in trained init; do timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 > gpurun_out/${TAG}_raster_bench_$k.json; cat gpurun_out/${TAG}_raster_bench_$k.json | cut -c1-600; done
