"""Drop-in for the reference's `diff_gaussian_rasterization` package
(/root/reference/dgmesh/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py):
`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer` resolves to the
MI355X implementation in dg-mesh_amd/rasterizer.py."""
import importlib

_impl = importlib.import_module("dg-mesh_amd.rasterizer")
_C = _impl._C
GaussianRasterizationSettings = _impl.GaussianRasterizationSettings
GaussianRasterizer = _impl.GaussianRasterizer
rasterize_gaussians = _impl.rasterize_gaussians
_RasterizeGaussians = _impl._RasterizeGaussians
cpu_deep_copy_tuple = _impl.cpu_deep_copy_tuple
