#!/bin/bash
# round 5: forward binning chain at the bench's own length (230 steps: the scene inflates under the random targets, mid-class tiles appear)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -q -x 2>&1 | tail -2
rm -rf /tmp/prof_long
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_long -o t -- env DGM_BENCH_STEADY_STEPS=0 python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/bin_long.log" 2>&1 )
grep '^{"metric"' gpurun_out/bin_long.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('long', d['value'], d['ms_per_step'])"
python tools/chain_wall.py $(find /tmp/prof_long -name "*kernel_trace.csv" | head -1) gpurun_out/r05_chain_long.json | tr -d '\n' | cut -c1-900; echo
python tools/prof_summary.py $(find /tmp/prof_long -name "*kernel_stats.csv" | head -1) 1 50 | grep -i "total\|tile_\|scatter\|count_\|preprocess_fwd\|fill\|copy" | cut -c1-120
