#!/bin/bash
# Builds the REFERENCE's own rasterizer (and simple-knn) kernels for gfx950 as a GPU-side checker:
#   /root/reference/dgmesh/submodules/{diff-gaussian-rasterization/cuda_rasterizer,simple-knn}/*  (read where they lie)
#   --hipify-perl + 5 mechanical text fixes, in a temporary directory-->  hipcc --offload-arch=gfx950
#   --> oracle/_ref/libref_raster.so       (-ffp-contract=off: the canonical arithmetic of DESIGN.md section 3)
#       oracle/_ref/libref_raster_fma.so   (hipcc default contraction: what a plain port of the reference would do)
# Only the binaries are kept (oracle/_ref/ is git-ignored but travels to the GPU box); no reference source is
# copied into the repository.  Skips silently when /root/reference is absent (e.g. on the GPU box).
set -e
REF=/root/reference/dgmesh/submodules
HERE="$(cd "$(dirname "$0")" && pwd)"
[ -d "$REF/diff-gaussian-rasterization/cuda_rasterizer" ] || { echo "build_ref: /root/reference not present, skipping"; exit 0; }
TMP="$(mktemp -d /tmp/dgm_refbuild.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
DGR="$REF/diff-gaussian-rasterization"
for f in forward.cu backward.cu rasterizer_impl.cu auxiliary.h forward.h backward.h rasterizer.h rasterizer_impl.h config.h; do
  /opt/rocm/bin/hipify-perl "$DGR/cuda_rasterizer/$f" > "$TMP/$f" 2>/dev/null
done
for f in simple_knn.cu simple_knn.h; do /opt/rocm/bin/hipify-perl "$REF/simple-knn/$f" > "$TMP/$f" 2>/dev/null; done
cd "$TMP"
# mechanical fixes: empty include left by hipify, CUDA-only headers, __trap, spaced launch chevrons
sed -i -e '/#include ""/d' -e '/cooperative_groups\/reduce.h/d' -e '/cub\/device\/device_radix_sort.cuh/d' \
       -e '/device_launch_parameters.h/d' -e 's/__trap()/abort()/' -e 's/<< </<<</g' -e 's/>> >/>>>/g' \
       -e 's/#define __CUDACC__//' *.cu *.h
mkdir -p "$HERE/_ref"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -include cfloat -I$TMP -I$DGR/third_party/glm -DGLM_FORCE_QUIET"
KNN=""
if /opt/rocm/bin/hipcc $COMMON -c simple_knn.cu -o knn_probe.o 2>knn.err; then KNN="-DREF_WITH_KNN simple_knn.cu"; else echo "build_ref: simple-knn does not build here (see below), rasterizer only"; head -5 knn.err; fi
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $KNN forward.cu backward.cu rasterizer_impl.cu "$HERE/ref_shim.cpp" -o "$HERE/_ref/libref_raster.so"
/opt/rocm/bin/hipcc $COMMON forward.cu backward.cu rasterizer_impl.cu "$HERE/ref_shim.cpp" -o "$HERE/_ref/libref_raster_fma.so"
echo "build_ref: wrote $(ls "$HERE/_ref")"
# The reference's Python render() (gaussian_renderer/__init__.py:32-119) and the two pure-torch helper modules it imports,
# byte-compiled (binaries only, same rule as the kernels) so that a GPU test can execute the reference's own render() over
# this repo's drop-in packages (tests/test_reference_render.py).
python3 - "$HERE/_ref/pyref" <<'PY'
import os, py_compile, sys
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
R = "/root/reference/dgmesh"
for src, name in ((R + "/gaussian_renderer/__init__.py", "gaussian_renderer.pyc"), (R + "/utils/sh_utils.py", "sh_utils.pyc"),
                  (R + "/utils/rigid_utils.py", "rigid_utils.pyc"),
                  # the networks and the image loss of the train step, for bench.py's cpu_baseline leg (the reference's own modules
                  # on the host cores; pure torch, they import nothing but utils.rigid_utils)
                  (R + "/utils/time_utils.py", "time_utils.pyc"), (R + "/utils/loss_utils.py", "loss_utils.pyc"),
                  # the configuration objects of train.py:858-906 (argparse groups + the YAML merge), for
                  # tests/test_reference_configs.py: the reference's own OptimizationParams / PipelineParams / ModelParams
                  # merged with its shipped YAML files drive Trainer.step
                  (R + "/arguments/__init__.py", "arguments.pyc"), (R + "/utils/system_utils.py", "system_utils.pyc")):
    py_compile.compile(src, cfile=os.path.join(out, name), dfile=os.path.basename(src), doraise=True)
print("build_ref: wrote pyref/", sorted(os.listdir(out)))
# the reference's shipped YAML configs (data files read by the test above at run time; oracle/_ref is git-ignored)
import shutil
cfg_out = os.path.join(os.path.dirname(out), "configs")
shutil.rmtree(cfg_out, ignore_errors=True)
shutil.copytree(R + "/configs", cfg_out)
print("build_ref: copied configs/", sum(len(f) for _, _, f in os.walk(cfg_out)), "files")
PY
