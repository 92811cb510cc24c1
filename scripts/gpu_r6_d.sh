#!/bin/bash
# round 6, call D: the stationary bench (teacher-rendered targets): R at 20 vs 200 steps, cfg4, cfg1 with its CPU leg; then the gpu suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], "it/s", round(d["value"], 2), "ms", round(d["ms_per_step"], 3), "R", d["config"]["num_rendered"], "live", d["config"].get("live_rows"),
      "allocs", d["host_ms_per_step"]["device_allocations_in_timed_region"], "steady", d.get("steady_state"))
print("   roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "traffic", "traffic_source")})
print("   frac_valu", d.get("frac_valu"))
print("   cpu", d.get("cpu_baseline"))
t = d.get("roofline_render_bwd_trained")
if t: print("   trained", {k: t.get(k) for k in ("avg_ms", "render_fwd_ms", "frac", "frac_valu")}, t.get("group_with_preprocess_bwd"))
PY
}
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06_d_bench20.json 2> gpurun_out/r06_d_bench20.err; show gpurun_out/r06_d_bench20.json
timeout 900 python bench.py --steps 200 > gpurun_out/r06_d_bench.json 2> gpurun_out/r06_d_bench.err; show gpurun_out/r06_d_bench.json
timeout 600 python bench.py --workload cfg4 --steps 40 --no-cpu-baseline --no-extras > gpurun_out/r06_d_bench_cfg4.json 2> gpurun_out/r06_d_bench_cfg4.err; show gpurun_out/r06_d_bench_cfg4.json
timeout 600 python bench.py --workload cfg1 --steps 200 --no-extras > gpurun_out/r06_d_bench_cfg1.json 2> gpurun_out/r06_d_bench_cfg1.err; show gpurun_out/r06_d_bench_cfg1.json
tail -3 gpurun_out/r06_d_bench*.err
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
