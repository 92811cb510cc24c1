#!/bin/bash
# usage: gpu_pmc_sq.sh <tag> <cmd...>
# Three separate rocprofv3 --pmc passes (SQ issue/wait counters, LDS/SALU counters, MFMA-busy + GRBM) over <cmd>,
# each with --kernel-trace only (no other trace domain), summarised per kernel into gpurun_out/pmc_sq_<tag>.json.
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
pass() {  # name, counters
  local name=$1 ctrs=$2
  ( cd /tmp && timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv \
      -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_${tag}_$name" -o pmc -- "${CMD[@]}" \
      > "$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_${tag}_$name.log" 2>&1; echo "pass $name: rocprof exit $?" )
}
CMD=("$@")
pass a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
pass b "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"
pass c "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE GRBM_COUNT"
python tools/pmc_sq_summary.py "$tag"
find gpurun_out -path "*pmc_sq_${tag}_*" -name "*kernel_trace.csv" -delete
find gpurun_out -path "*pmc_sq_${tag}_*" -name "*counter_collection.csv" -size +20M -delete
