#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_optim.py tests/test_trainer_dp_gpu.py tests/test_densify.py -m gpu -q -x 2>&1 | tail -3
DGM_BENCH_WORKLOAD=cfg1 timeout 600 python tools/host_profile.py 200 > gpurun_out/r4_host_cfg1_c.txt 2>&1
for w in cfg1 cfg2; do
timeout 600 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms', d['host_ms_per_step'])"
done
head -30 gpurun_out/r4_host_cfg1_c.txt
