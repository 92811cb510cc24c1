#!/bin/bash
# round 5, call A: the gpu suite, a baseline bench line on this box, and the clock / power telemetry of the MLP kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/r05_a_bench.json 2> gpurun_out/r05_a_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_a_bench.json'))
print(d['value'], d['ms_per_step'], d['host_ms_per_step'])
print(d['stages_ms'])
print('pre_bwd', d['kernels'].get('preprocess_bwd'), 'live', d['config'].get('live_rows'))
print('cpu', d.get('cpu_baseline'))
PY
timeout 600 python tools/power_probe.py 5 100000 2>&1 | tail -12
