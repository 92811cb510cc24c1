#!/bin/bash
# round 5: forward binning chain -- parity, then wall time of the chain from kernel-trace timestamps (default library and $VARIANTS)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -q -x 2>&1 | tail -3
run() {  # name, lib
  rm -rf /tmp/prof_$1
  ( cd /tmp && DGM_LIB_PATH=$2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$1 -o t -- env DGM_BENCH_STEADY_STEPS=0 python "$GRAFT_REPO_ROOT/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/bin_$1.log" 2>&1 )
  grep '^{"metric"' gpurun_out/bin_$1.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'])"
  python tools/chain_wall.py $(find /tmp/prof_$1 -name "*kernel_trace.csv" | head -1) gpurun_out/r05_chain_$1.json | tr -d '\n' | cut -c1-900; echo
}
run default ""
for v in $VARIANTS; do run $v "$GRAFT_REPO_ROOT/dg-mesh_amd/lib/variants/$v.so"; done
