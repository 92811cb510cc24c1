#!/bin/bash
# bench line + rocprofv3 kernel stats of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_run.log" 2>&1; echo "rocprof exit $?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f"
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
