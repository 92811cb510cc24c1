#!/bin/bash
# full GPU suite + host profile + bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
DGM_BENCH_WORKLOAD=cfg1 timeout 600 python tools/host_profile.py 200 > gpurun_out/r4_host_cfg1_b.txt 2>&1
timeout 600 python bench.py --workload cfg1 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg1', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', d['host_ms_per_step'])"
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', d['host_ms_per_step'])"
head -40 gpurun_out/r4_host_cfg1_b.txt
