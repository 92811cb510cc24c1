#!/bin/bash
# the round-end sequence: every GPU test, smoke, the bench line, rocprofv3 kernel stats of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo "bench exit $?"; tail -2 gpurun_out/r3_bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "steps")}, d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
print("f32", d.get("mlp_f32_mode", {}).get("value"), "cpu", d.get("cpu_baseline", {}).get("value"))
print({k: (v["avg_ms"], v["frac_hbm"]) for k, v in d["kernels"].items()})
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r3_prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/r3_prof_run.log" 2>&1; echo "rocprof exit $?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/r3_prof -name "*kernel_trace.csv" -delete; python tools/prof_summary.py gpurun_out/r3_prof 2>/dev/null | head -50
