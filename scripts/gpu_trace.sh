#!/bin/bash
# kernel timeline of a few steps (two-stream regime) for tools/timeline.py
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/trace
( cd /tmp && DGM_BENCH_STEADY_STEPS=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/trace" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 12 --warmup 5 --no-extras --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/trace/bench.log" 2>&1 )
f=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-3000:]  # the last steps
out = open("gpurun_out/trace/tail.csv", "w")
out.write("start,end,queue,name\n")
for r in rows:
    out.write(f'{r["Start_Timestamp"]},{r["End_Timestamp"]},{r.get("Queue_Id","")},"{r["Kernel_Name"][:60]}"\n')
PY
find gpurun_out/trace -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/trace/bench.log | cut -c1-200
