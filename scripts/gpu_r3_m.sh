#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_trainer_dp_gpu.py tests/test_mlp.py tests/test_gpu_vs_reference.py tests/test_dpsr.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|Error|error|fragile|unmasked|max rel-to-max|assert" | cut -c1-400 | tail -30
for ph in "cfg2 mesh" "cfg5 mesh" "cfg5 gs"; do set -- $ph
timeout 900 python bench.py --workload $1 --phase $2 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3m_bench_$1_$2.json 2> gpurun_out/r3m_bench_$1_$2.err; echo "bench $1 $2 exit $?"; tail -2 gpurun_out/r3m_bench_$1_$2.err; python -c "
import json; d=json.load(open('gpurun_out/r3m_bench_$1_$2.json')); print('$1 $2', round(d['value'],1), round(d['ms_per_step'],3), d['config']['parallelism'], d['config']['num_rendered'])"
done
