#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r04_bench2.json 2> gpurun_out/r04_bench2.err; echo "bench exit $?"; tail -3 gpurun_out/r04_bench2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench2.json'))
print(d['value'], d['ms_per_step'], d['host_ms_per_step'])
print('densify', d.get('with_densify'))
print('trained', d.get('roofline_render_bwd_trained'))
print('rb', d.get('roofline_render_bwd'))
print('fv', d.get('frac_valu'))
PY
