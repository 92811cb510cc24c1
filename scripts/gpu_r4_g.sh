#!/bin/bash
# A/B of render_bwd4 variants on both scenes (+ parity of the default build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -3
for k in init trained; do
  echo "== default $k"; timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1
  for v in dg-mesh_amd/lib/variants/r4_*.so; do
    [ -f "$v" ] || continue
    echo "== $(basename $v .so) $k"; DGM_LIB_PATH=$v timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4_rb_$c" -o pmc -- python "$GRAFT_REPO_ROOT/tools/raster_bench.py" cfg2 --kind init --iters 6 --profile 0 > /dev/null 2>&1 )
  f=$(find gpurun_out/r4_rb_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY'
import csv,sys,collections
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"]==sys.argv[2]:
        for k in ("render_bwd4","render_fwd","preprocess_bwd"):
            if k in r["Kernel_Name"]: tot[k]+=float(r["Counter_Value"]); n[k]+=1
for k in tot: print(sys.argv[2], k, "MB per launch %.1f"%(tot[k]/n[k]/1024), "launches", n[k])
PY
  find gpurun_out/r4_rb_$c -name "*.csv" -delete
done
