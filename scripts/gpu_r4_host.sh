#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/host_profile.py 200 > gpurun_out/r4_host_cfg2.txt 2>&1
DGM_BENCH_WORKLOAD=cfg1 timeout 600 python tools/host_profile.py 200 > gpurun_out/r4_host_cfg1.txt 2>&1
timeout 600 python bench.py --workload cfg1 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg1', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', d['host_ms_per_step'])"
head -60 gpurun_out/r4_host_cfg1.txt
