#!/bin/bash
# per-kernel average times of the MLP micro benchmark under rocprofv3, for the product library and every variant in lib/variants
# usage (on the GPU box): bash tools/mlp_kernel_times.sh [N] [tag]
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-100000}; TAG=${2:-kt}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
for lib in dg-mesh_amd/lib/libdgmesh_hip.so dg-mesh_amd/lib/variants/*.so; do
  [ -f "$lib" ] || continue
  name=$(basename $lib .so)
  ( cd /tmp && DGM_LIB_PATH="$GRAFT_REPO_ROOT/$lib" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/$TAG/$name" -o b -- python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" $N 20 > "$GRAFT_REPO_ROOT/gpurun_out/$TAG/$name.log" 2>&1 )
  echo "== $name: $(grep impl= gpurun_out/$TAG/$name.log)"
  f=$(find gpurun_out/$TAG/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    n = r["Name"].split("(")[0].replace("void dgm::", "").replace("dgm::", "")
    print(f"   {n[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
  find gpurun_out/$TAG/$name -name "*kernel_trace.csv" -delete
done
