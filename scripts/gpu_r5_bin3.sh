#!/bin/bash
# round 5: A/B of library variants over the bench's own 230 steps (kernel stats of the binning kernels + it/s); VARIANTS="name ..."
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  rm -rf /tmp/prof_$1
  ( cd /tmp && DGM_LIB_PATH=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o t -- env DGM_BENCH_STEADY_STEPS=0 python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/bin3_$1.log" 2>&1 )
  grep '^{"metric"' gpurun_out/bin3_$1.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'])"
  python tools/prof_summary.py $(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1) 1 60 | grep -i "total\|tile_sort\|scatter_kernel\|count_tiles\|tile_scan" | cut -c1-120
}
for v in $VARIANTS; do
  [ -n "$PARITY" ] && DGM_LIB_PATH="$GRAFT_REPO_ROOT/dg-mesh_amd/lib/variants/$v.so" timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -q -x 2>&1 | tail -2
done
run default ""
for v in $VARIANTS; do run $v "$GRAFT_REPO_ROOT/dg-mesh_amd/lib/variants/$v.so"; done
