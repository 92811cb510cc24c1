"""Drop-in replacement of `simple_knn._C.distCUDA2` (KNN/spatial.cu:15-26, KNN/ext.cpp:15-16) on top of
libdgmesh_hip.so.  (KNN/ = /root/reference/dgmesh/submodules/simple-knn/.)"""
import ctypes

import torch

from . import _lib


def distCUDA2(points):
    """points: (P,3) float32 on the GPU -> (P,) mean squared distance to the 3 nearest neighbours."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 needs a CUDA/HIP tensor (dg-mesh_amd has no CPU path)")
    P = points.size(0)
    pts = points.contiguous().float()
    means = torch.zeros((P,), dtype=torch.float32, device=points.device)  # torch::full({P}, 0.0), spatial.cu:21
    if P:
        L = _lib.lib()
        # scratch is caller-owned like every buffer of the C ABI: a torch allocation (256-byte aligned by the caching allocator)
        scratch = torch.empty((int(L.dgm_knn_scratch_bytes(P)),), dtype=torch.uint8, device=points.device)
        with _lib.device_guard(points.device):
            _lib.check(L.dgm_knn_mean_dist2(P, ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(means.data_ptr()),
                                            ctypes.c_void_p(scratch.data_ptr()), _lib.stream_ptr()))
    return means
