#!/bin/bash
# usage: gpu_prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/ + compact summary
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline "$@" > "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log" 2>&1; echo "rocprof exit $?"
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py "$f" 223 45
