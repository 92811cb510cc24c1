#!/bin/bash
# ordered tile hand-out on dense and sparse frames (DGM_RF_ORDER=0: raster order), the per-wave trace of the asynchronous sparse
# forward with and without it (library built with -DRF_TRACE=1: tools/build_variant.sh rf_trace -DRF_TRACE=1 render), then the raster parity suites
for k in trained init; do for o in 1 0 1 0; do DGM_RF_ORDER=$o timeout 120 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$k order=$o', 'render_fwd', round(d['render_fwd'],4), 'render_bwd', round(d['render_bwd'],4), 'tile_scan', round(d['bin_scan'],4))"; done; done | tee gpurun_out/r06_order.txt
for o in 1 0; do echo "== DGM_RF_ORDER=$o"; DGM_RF_ORDER=$o DGM_LIB_PATH=dg-mesh_amd/lib/variants/rf_trace.so timeout 120 python tools/raster_bench.py cfg2 --kind trained --iters 5 --trace-fwd 2>&1 | tail -7 | cut -c1-1500; done > gpurun_out/r06_render_fwd_async_trace.txt
tail -5 gpurun_out/r06_render_fwd_async_trace.txt | cut -c1-600
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py tests/test_trainer_dp_gpu.py -m gpu -q -x 2>&1 | tail -3
