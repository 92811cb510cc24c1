#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
b() { python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>&1 | grep '^{' | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'; }
for n in 112 128 144; do echo "pair=$n: $(DGM_MLP_PAIR=$n b)"; done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_e_bench.json 2> gpurun_out/r03_e_bench.err
python - <<'PY'
import json
r=json.loads([l for l in open("gpurun_out/r03_e_bench.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["roofline"]["kernel"][:30], r["roofline"]["frac"], r["roofline"].get("one_stream"), r.get("one_stream",{}).get("value"), r.get("mlp_f32_mode",{}).get("value"))
for k,v in r["kernels"].items(): print(k, v["avg_ms"], v["launches_per_step"], r.get("one_stream",{}).get("avg_ms",{}).get(k))
PY
