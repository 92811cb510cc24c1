#!/bin/bash
# round 6 evidence, PMC passes (each with --kernel-trace only): HBM traffic of the bench's kernels (FETCH_SIZE / WRITE_SIZE, separate
# passes; the files now record N, launches per step and the commit they were taken at -- bench.py prints `roofline.traffic` only when
# those match its own run), SQ counters of the bench scene and of the trained-like scene.   usage: DGM_COMMIT=<hash> bash scripts/gpu_r6_pmc.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=4; WARM=1; TOTAL=$((33 + WARM + STEPS + 1))   # bench.py: 33 priming steps, W warm-up, K timed, one live-row counting step
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c" -o pmc -- env DGM_BENCH_STEADY_STEPS=0 python "$GRAFT_REPO_ROOT/bench.py" --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log" 2>&1; echo "$c: rocprof exit $?" )
done
F=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" gpurun_out/pmc_traffic cfg2 100000 $TOTAL "${DGM_COMMIT:-unknown}"
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -delete
bash scripts/gpu_pmc_sq.sh r06_bench env DGM_BENCH_STEADY_STEPS=0 python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | grep -i "render\|preprocess_bwd\|bwd_pair\|pass\|gemm" | cut -c1-300
bash scripts/gpu_pmc_sq.sh r06_trained python $GRAFT_REPO_ROOT/tools/raster_bench.py cfg2 --kind trained --iters 15 --profile 0 2>&1 | grep -i "render_bwd4\|render_fwd\|pass" | cut -c1-300
