#!/bin/bash
# tools/build_variant_file.sh NAME FILE "-DFOO=1 ..." [FILE2 "-DBAR=1" ...] : libdgmesh_hip.so with csrc/FILE.hip (and FILE2 ...)
# compiled under extra defines -> dg-mesh_amd/lib/variants/NAME.so
# (A/B material for gpurun: DGM_LIB_PATH=dg-mesh_amd/lib/variants/NAME.so python tools/raster_bench.py ...)
set -e
cd "$(dirname "$0")/../dg-mesh_amd/csrc"
make -s >/dev/null
mkdir -p build/variants ../lib/variants
name=$1; shift
objs=$(ls build/*.o)
extra=""
while [ $# -ge 2 ]; do
  f=$1; defs=$2; shift 2
  flags=$(grep "^FLAGS_$f = " Makefile | sed 's/.*= //' | sed 's/\$(EXACT)/-ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt/; s/\$(FAST)/-ffp-contract=fast/')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function $flags $defs -c $f.hip -o build/variants/${f}_$name.o
  objs=$(echo "$objs" | grep -v "build/$f.o")
  extra="$extra build/variants/${f}_$name.o"
  echo "  $f: $flags $defs"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$name.so $objs $extra
echo "built variants/$name.so"
