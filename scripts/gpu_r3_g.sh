#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash scripts/gpu_pmc_sq.sh r3g python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" 100000 6 2>&1 | grep -E "gemm4_kernel<16, 1024, 512, [01]|dw4_kernel<8, 8|pass"
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_sq_r3g.json"))
for k, v in d.items():
    if "gemm4_kernel<16, 1024, 512, 0" in k or "dw4_kernel<8, 8" in k:
        print(k[:60])
        for n in sorted(v): print(f"    {n}: {v[n]:.4g}")
PY
