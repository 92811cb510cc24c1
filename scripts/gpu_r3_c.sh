#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/mlp_kernel_times.sh 100000 r3c_kt 2>&1 | grep -E "^==|gemm4_kernel<16, 1024, 512, [01]|dw4_kernel<8, 8"
bash scripts/gpu_pmc_sq.sh r3c python tools/mlp_bench.py 100000 6 2>&1 | tail -5
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_sq_r3c.json"))
for k, v in d.items():
    if "gemm4_kernel<16, 1024, 512, 0" in k or "dw4_kernel<8, 8" in k or "gemm4_kernel<16, 1024, 512, 1" in k:
        print(k[:60], json.dumps(v)[:1500])
PY
