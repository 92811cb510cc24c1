#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_host.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -3
for m in mail copy mail copy; do
DGM_R_READBACK=$m timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m cfg2', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', d['host_ms_per_step'])"
done
for m in mail copy; do
DGM_R_READBACK=$m timeout 600 python bench.py --workload cfg1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m cfg1', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', d['host_ms_per_step'])"
done
