#!/bin/bash
# round 4, call E: LDS-staged gather in preprocess_bwd; chunked tile order in the paired MLP backward; SQ counters of the blend kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -5
for k in init trained; do
  echo "== new $k"; timeout 300 python tools/raster_bench.py cfg2 --kind $k --iters 30 2>&1 | tail -1 | tee gpurun_out/r4_e_new_$k.json
done
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x -k "stage_by_stage or (f16x3p and (big_batch or torch_trunk))" 2>&1 | tail -3
for c in 0 1 0 1; do
  echo "pair chunked=$c: $(DGM_MLP_PAIR_CHUNKED=$c python tools/mlp_bench.py 100000 30 2>&1 | grep impl=)"
done
for c in 0 1; do
  ( cd /tmp && DGM_MLP_PAIR_CHUNKED=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4_pair_c$c" -o b -- python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" 100000 20 > "$GRAFT_REPO_ROOT/gpurun_out/r4_pair_c$c.log" 2>&1 )
  f=$(find gpurun_out/r4_pair_c$c -name "*kernel_stats.csv" | head -1); echo "chunked=$c"; python tools/prof_summary.py $f 23 6; find gpurun_out/r4_pair_c$c -name "*kernel_trace.csv" -delete
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && DGM_MLP_PAIR_CHUNKED=$c timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r4_pair_c${c}_$ctr" -o pmc -- python "$GRAFT_REPO_ROOT/tools/mlp_bench.py" 100000 4 > /dev/null 2>&1 )
    f=$(find gpurun_out/r4_pair_c${c}_$ctr -name "*counter_collection.csv" | head -1)
    python - "$f" $ctr <<'PY'
import csv,sys,collections
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"]==sys.argv[2] and "bwd_pair" in r["Kernel_Name"]:
        tot["pair"]+=float(r["Counter_Value"]); n["pair"]+=1
for k in tot: print(sys.argv[2], k, "KB per launch", tot[k]/n[k], "launches", n[k])
PY
    find gpurun_out/r4_pair_c${c}_$ctr -name "*.csv" -delete
  done
done
for k in init trained; do
  bash scripts/gpu_pmc_sq.sh r4_$k python $GRAFT_REPO_ROOT/tools/raster_bench.py cfg2 --kind $k --iters 12 --profile 0 2>&1 | grep -i "render\|preprocess_bwd\|pass"
done
