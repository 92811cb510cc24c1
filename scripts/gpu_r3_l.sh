#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_reference_render.py tests/test_densify.py tests/test_opacity_field.py tests/test_rccl.py -m gpu -q -x 2>&1 | tail -15
timeout 900 python bench.py --steps 100 --warmup 10 > gpurun_out/r3l_bench.json 2> gpurun_out/r3l_bench.err; echo "bench exit $?"; tail -3 gpurun_out/r3l_bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r3l_bench.json"))
for k in ("value", "ms_per_step", "n_gpus", "rccl_world_size", "mlp_f32_mode", "roofline_render_bwd_trained", "frac_valu", "roofline", "cpu_baseline"):
    print(k, json.dumps(d.get(k))[:600])
PY
