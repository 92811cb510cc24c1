#!/bin/bash
# tools/build_variant_file.sh NAME FILE "-DFOO=1 ..." : libdgmesh_hip.so with csrc/FILE.hip compiled under extra defines
#   -> dg-mesh_amd/lib/variants/NAME.so   (A/B material for gpurun: DGM_LIB_PATH=dg-mesh_amd/lib/variants/NAME.so python tools/raster_bench.py ...)
set -e
cd "$(dirname "$0")/../dg-mesh_amd/csrc"
make -s >/dev/null
mkdir -p build/variants ../lib/variants
flags=$(make -s -p -n 2>/dev/null | grep "^FLAGS_$2 = " | sed 's/.*= //' | sed 's/\$(EXACT)/-ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt/; s/\$(FAST)/-ffp-contract=fast/')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function $flags $3 -c $2.hip -o build/variants/$2_$1.o
objs=$(ls build/*.o | grep -v "build/$2.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$1.so $objs build/variants/$2_$1.o
echo "built variants/$1.so ($2: $flags $3)"
