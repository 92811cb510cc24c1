#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mlp.py tests/test_trainer_dp_gpu.py -m gpu -q 2>&1 | tail -3
for sd in 1 0; do
  echo "side=$sd: $(DGM_MLP_SIDE=$sd python tools/mlp_bench.py 100000 30 2>&1 | grep impl=)"
  DGM_MLP_SIDE=$sd timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/side$sd.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/side$sd.json')); print('side=$sd bench', round(d['value'],1), round(d['ms_per_step'],3), d['host_ms_per_step'])"
done
