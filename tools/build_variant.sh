#!/bin/bash
# tools/build_variant.sh NAME "-DFOO=1 ..." [SOURCE]: libdgmesh_hip.so with SOURCE.hip (default: mlp) compiled under extra defines
#   -> dg-mesh_amd/lib/variants/NAME.so
# (A/B and trace material for gpurun: DGM_LIB_PATH=dg-mesh_amd/lib/variants/NAME.so python tools/mlp_bench.py ...; e.g.
#  tools/build_variant.sh rf_trace -DRF_TRACE=1 render   -> tools/raster_bench.py --trace-fwd
#  tools/build_variant.sh rb4_trace -DRB4_TRACE=1 render_bwd4 -> tools/raster_bench.py --trace)
set -e
cd "$(dirname "$0")/../dg-mesh_amd/csrc"
src=${3:-mlp}
make -s >/dev/null
mkdir -p build/variants ../lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -ffp-contract=fast $2 -c $src.hip -o build/variants/${src}_$1.o
objs=$(ls build/*.o | grep -v "build/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$1.so $objs build/variants/${src}_$1.o
echo "built variants/$1.so ($2, $src.hip)"
