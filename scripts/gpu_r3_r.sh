#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_mlp.py tests/test_trainer_dp_gpu.py tests/test_host.py -m gpu -q 2>&1 | tail -4
for m in 128 0; do
DGM_MLP_PAIR=$m timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r3r_bench_$m.json 2> gpurun_out/r3r_bench_$m.err; echo "bench pair=$m exit $?"; python -c "
import json; d=json.load(open('gpurun_out/r3r_bench_$m.json')); print('pair=$m', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['kernel'][:40], d['roofline']['frac'], {k: v['avg_ms'] for k, v in d['kernels'].items() if k.startswith('mlp')})"
done
bash scripts/gpu_pmc_traffic.sh 2>&1 | grep -E "bwd_pair|gemm4_fwd|reduce"
