#!/bin/bash
# round 6, call F: alloc probe, the asynchronous-quadrant sparse forward (tests + raster_bench A/B on the trained-like scene), the
# adversarial / envelope tests, first-steps value test, the long multi-rank runs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/alloc_probe.py 40 2>&1 | tail -8
echo "== raster parity (sparse frames take the async kernel)"
timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -q -x 2>&1 | tail -6
echo "== trained-like scene: async vs sync quadrants"
for m in async sync; do
  DGM_RF_SPARSE=$m timeout 300 python tools/raster_bench.py cfg2 --kind trained --iters 30 2>&1 | tail -14 > gpurun_out/r06_f_raster_trained_$m.txt
  echo "-- $m"; grep -E "render_fwd|render_bwd|preprocess|sort|scatter|total|sum|fwd\+bwd" gpurun_out/r06_f_raster_trained_$m.txt | head -12
done
echo "-- init scene (dense kernel, unchanged)"; timeout 300 python tools/raster_bench.py cfg2 --kind init --iters 20 2>&1 | grep -E "render_fwd|render_bwd" | head -4
echo "== MLP adversarial / envelope"
timeout 900 python -m pytest tests/test_mlp.py -m gpu -q -k "adversarial or envelope" -s 2>&1 | grep -E "hidden units|passed|failed" | head -20
echo "== long multi-rank runs"
timeout 1500 python -m pytest tests/test_trainer_dp_gpu.py -m gpu -q -x -k "many_ranks" -s 2>&1 | tail -12
