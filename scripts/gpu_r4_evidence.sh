#!/bin/bash
# round 4 evidence: bench line, rocprofv3 kernel stats of the bench, HBM traffic (FETCH_SIZE / WRITE_SIZE passes), SQ counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench.json'))
print(d['value'], d['ms_per_step'], d['host_ms_per_step'], d['dtype'])
print(d['stages_ms'])
print('trained', d.get('roofline_render_bwd_trained',{}).get('avg_ms'), d.get('roofline_render_bwd_trained',{}).get('group_with_preprocess_bwd'))
print('densify', d.get('with_densify'))
print('cpu', d.get('cpu_baseline'))
print('f32', d.get('mlp_f32_mode'))
PY
( cd /tmp && DGM_BENCH_STEADY_STEPS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r04_prof" -o b -- python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r04_prof.log" 2>&1 )
f=$(find gpurun_out/r04_prof -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 1 45 > gpurun_out/r04_bench_kernel_stats.txt; head -12 gpurun_out/r04_bench_kernel_stats.txt; find gpurun_out/r04_prof -name "*kernel_trace.csv" -delete
bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -25
bash scripts/gpu_pmc_sq.sh r04 env DGM_BENCH_STEADY_STEPS=0 python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -i "render\|preprocess_bwd\|bwd_pair\|pass"
