#!/bin/bash
# tools/build_variant.sh NAME "-DFOO=1 ..." : libdgmesh_hip.so with mlp.hip compiled under extra defines -> dg-mesh_amd/lib/variants/NAME.so
# (A/B material for gpurun: DGM_LIB_PATH=dg-mesh_amd/lib/variants/NAME.so python tools/mlp_bench.py ...)
set -e
cd "$(dirname "$0")/../dg-mesh_amd/csrc"
make -s >/dev/null
mkdir -p build/variants ../lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -ffp-contract=fast $2 -c mlp.hip -o build/variants/mlp_$1.o
objs=$(ls build/*.o | grep -v "build/mlp.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$1.so $objs build/variants/mlp_$1.o
echo "built variants/$1.so ($2)"
