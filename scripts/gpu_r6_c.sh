#!/bin/bash
# round 6, call C: MLP tests (all modes), kernel times of a network pass per mode, and the bench A/B: mode 3 (time row folded, K = 320
# skip layer) against mode 4 (rounds 3-5's forms)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_mlp.py -m gpu -q -x 2>&1 | tail -5
for mode in f16x3p f16x3p8; do
  ( cd /tmp && DGM_MLP_GEMM=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r06_c_$mode" -o b -- python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" 100000 20 > "$GRAFT_REPO_ROOT/gpurun_out/r06_c_$mode.log" 2>&1 )
  echo "== $mode: $(grep impl= gpurun_out/r06_c_$mode.log)"
  f=$(find gpurun_out/r06_c_$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
mlp = sum(float(r["TotalDurationNs"]) for r in rows if "dgm::" in r["Name"])
print(f"   dgm kernels per pass: {mlp/20/1e3/1.15:.1f} us (23 passes incl. warm-up)")
for r in rows[:24]:
    n = r["Name"].split("(")[0].replace("void dgm::", "").replace("dgm::", "")
    if "at::native" in n: continue
    print(f"   {n[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
  find gpurun_out/r06_c_$mode -name "*kernel_trace.csv" -delete
done
for mode in f16x3p f16x3p8 f16x3p f16x3p8; do
  DGM_MLP_GEMM=$mode timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 > gpurun_out/r06_c_bench_$mode.json 2> gpurun_out/r06_c_bench_$mode.err
  python -c "
import json; d=json.load(open('gpurun_out/r06_c_bench_$mode.json')); print('$mode', d['value'], d['ms_per_step'])"
done
