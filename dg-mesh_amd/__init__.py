"""dg-mesh_amd: MI355X-native (gfx950) implementation of the DG-Mesh training hot path.

Only what the path needs lives here:
  csrc/          hand-written HIP kernels + the C ABI (include/dgmesh_hip.h) -> lib/libdgmesh_hip.so
  _lib.py        ctypes loader (fails loudly when the library is missing: there is no fallback)
  rasterizer.py  `diff_gaussian_rasterization` API (GaussianRasterizationSettings, GaussianRasterizer, _C)
  knn.py         `simple_knn._C.distCUDA2`
  synthetic.py   seeded synthetic D-NeRF-like workloads (no datasets on the build / GPU box)
The directory name contains a hyphen (fixed by the project layout), so import it with
`importlib.import_module("dg-mesh_amd")` or through the top-level aliases `diff_gaussian_rasterization`,
`simple_knn` and `dgmesh_amd`.
"""
__all__ = ["synthetic"]
