#!/bin/bash
# round 5, call B: where the layer GEMM's energy goes -- the power probe over ablation builds of the same kernel (-DP4_ABL bits:
# 1 no global stores, 2 no tile copies, 4 no MFMAs, 8 no epilogue, 16 no fragment reads) and over a plain copy of the same bytes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
export DGM_PROBE_ZERO=0
DGM_PROBE_KINDS=9,0 DGM_PROBE_TAG=r05_abl_full timeout 300 python tools/power_probe.py 4 100000 2>&1 | grep -v amdgpu.ids
for v in 4 16 8 3 1; do
  DGM_PROBE_LIB=probe_abl$v DGM_PROBE_KINDS=0 DGM_PROBE_TAG=r05_abl_$v timeout 300 python tools/power_probe.py 4 100000 2>&1 | grep -v amdgpu.ids
done
