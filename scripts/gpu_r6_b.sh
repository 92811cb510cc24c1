#!/bin/bash
# round 6, call B: time / clock / power of the one-wave-per-SIMD kernels beside the eight-wave ones, and ablations of gemm5
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out tools/bin
export TMPDIR=/tmp
gcc -O2 tools/smi_sampler.c -I/opt/rocm/include -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib -o tools/bin/smi_sampler 2>&1 | tail -2
SEC=${1:-2}
DGM_PROBE_TAG=r06_power DGM_PROBE_KINDS=${2:-0,4,3,5,2,6,1,7} DGM_PROBE_ZERO=0 timeout 600 python tools/power_probe.py $SEC 100000 2>&1 | grep -v "^$" | tail -12
for v in dg-mesh_amd/lib/variants/probe_*.so; do
  [ -f "$v" ] || continue
  n=$(basename $v .so)
  DGM_PROBE_LIB=$n DGM_PROBE_TAG=r06_$n DGM_PROBE_KINDS=${3:-4} DGM_PROBE_ZERO=0 timeout 300 python tools/power_probe.py $SEC 100000 2>&1 | grep "us/launch"
done
