#!/bin/bash
# usage: gpu_pmc.sh <tag> "<counters>" <cmd...>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; ctrs=$2; shift; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag" -o pmc -- "$@" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log" 2>&1; echo "rocprof exit $?"
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py "$f" dgm::
find gpurun_out/pmc_$tag -name "*kernel_trace.csv" -delete
