#!/bin/bash
# round 6, call A: the MLP tests in the three modes, then per-kernel times of the MLP micro benchmark (rocprofv3) for round 6's kernel
# forms (DGM_MLP_GEMM=f16x3p) against rounds 3-5's (f16x3p8)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_mlp.py -m gpu -q -x 2>&1 | tail -15
for mode in f16x3p f16x3p8; do
  ( cd /tmp && DGM_MLP_GEMM=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r06_a_$mode" -o b -- python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" 100000 20 > "$GRAFT_REPO_ROOT/gpurun_out/r06_a_$mode.log" 2>&1 )
  echo "== $mode: $(grep impl= gpurun_out/r06_a_$mode.log)"
  f=$(find gpurun_out/r06_a_$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    n = r["Name"].split("(")[0].replace("void dgm::", "").replace("dgm::", "")
    print(f"   {n[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
  find gpurun_out/r06_a_$mode -name "*kernel_trace.csv" -delete
done
