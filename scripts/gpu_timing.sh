#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in dg-mesh_amd/lib/variants/t_*.so; do
  echo "== $(basename $lib .so)"
  DGM_LIB_PATH=$GRAFT_REPO_ROOT/$lib python tools/mlp_bench.py 100000 10 2>&1 | grep -v amdgpu.ids | tail -11
done
