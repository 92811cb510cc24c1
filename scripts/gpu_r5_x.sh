#!/bin/bash
# XCD-local pre-reduction of the weight-gradient partial tiles inside the paired backward launch: what it costs the launch
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp DGM_PROBE_ZERO=0 DGM_PROBE_KINDS=2 DGM_PROBE_LIB=probe_xcd
DGM_P4_XCD_OFF=1 DGM_PROBE_TAG=r05_xcd_off timeout 300 python tools/power_probe.py 4 100000 2>&1 | grep -v amdgpu.ids
DGM_PROBE_TAG=r05_xcd_on timeout 300 python tools/power_probe.py 4 100000 2>&1 | grep -v amdgpu.ids
DGM_P4_XCD_OFF=1 DGM_PROBE_TAG=r05_xcd_off2 timeout 300 python tools/power_probe.py 4 100000 2>&1 | grep -v amdgpu.ids
DGM_PROBE_TAG=r05_xcd_on2 timeout 300 python tools/power_probe.py 4 100000 2>&1 | grep -v amdgpu.ids
