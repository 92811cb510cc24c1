#!/bin/bash
# sparse-frame forward variants on the trained-like scene: asynchronous quadrants with / without the length-ordered tile hand-out,
# the ring (shared staging without lock step) and round 5's barrier form; then the raster parity suites
for m in "async 1" "async 0" "ring 1" "sync 1" "async 1"; do set -- $m; DGM_RF_SPARSE=$1 DGM_RF_ORDER=$2 timeout 120 python tools/raster_bench.py cfg2 --kind trained --iters 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 order=$2', round(d['render_fwd'],4), round(d['render_bwd'],4), 'scan', round(d['bin_scan'],4))"; done
timeout 120 python tools/raster_bench.py cfg2 --kind init --iters 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('init', round(d['render_fwd'],4), 'scan', round(d['bin_scan'],4))"
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -q -x 2>&1 | tail -3
