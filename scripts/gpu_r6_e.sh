#!/bin/bash
# round 6, call E: the new tests (sync-free forward, loss vs the reference functions, adversarial plane statistics, full gradient
# goldens, first steps vs the reference-shaped step), the raster suites under DGM_SYNC_FREE=1, the bench at the driver's 20 steps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_trainer_dp_gpu.py tests/test_loss.py -m gpu -q -x -k "capacity or sync_free or reference_functions" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_mlp.py -m gpu -q -k "adversarial or every_gradient" -s 2>&1 | grep -E "adversarial|passed|failed|Error|error" | head -30
timeout 900 python -m pytest tests/test_gpu_vs_reference.py -m gpu -q -x -k "first_steps" -s 2>&1 | tail -12
echo "== raster suites under DGM_SYNC_FREE=1"
DGM_SYNC_FREE=1 timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py tests/test_reference_render.py -m gpu -q -x 2>&1 | tail -6
echo "== bench 20 steps (driver's command), sync and sync-free"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06_e_bench20.json 2> gpurun_out/r06_e_bench20.err
DGM_SYNC_FREE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06_e_bench20_sf.json 2> gpurun_out/r06_e_bench20_sf.err
python - <<'PY'
import json
for f in ("r06_e_bench20", "r06_e_bench20_sf"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, round(d["value"], 2), round(d["ms_per_step"], 3), "R", d["config"]["num_rendered"], d["host_ms_per_step"], "steady", d.get("steady_state"))
    except Exception as e:
        print(f, "failed", e, open(f"gpurun_out/{f}.err").read()[-1500:])
PY
