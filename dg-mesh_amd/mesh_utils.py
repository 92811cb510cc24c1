"""get_opacity_field_from_gaussians (R/utils/mesh_utils.py:7-76; R/ = /root/reference/dgmesh/) on the device: the opacity
field sum_g opacity_g exp(-0.5 d^T Sigma_g^-1 d) on a resolution^3 grid over [-bbox_scale, bbox_scale]^3, Gaussians
assigned to the num_blocks^3 grid blocks by the reference's rule (centre inside the block's box grown by
(2 / num_blocks) * relax_ratio, opacity above the threshold).  One HIP launch pair (csrc/opacity_field.hip) instead of the
reference's Python triple loop over 4096 blocks.  CUDA/HIP tensors only."""
import ctypes

import torch

from . import _lib


def get_opacity_field_from_gaussians(xyzs, rotations, scalings, opacities, resolution=256, num_blocks=16, relax_ratio=0.5,
                                     opacity_threshold=0.005, bbox_scale=1.25):
    if not xyzs.is_cuda:
        raise RuntimeError("get_opacity_field_from_gaussians needs CUDA/HIP tensors (dg-mesh_amd has no CPU path)")
    block_size = 2 / num_blocks
    assert resolution % block_size == 0  # the reference's (vacuous for these values) check, kept
    L = _lib.lib()
    dev = xyzs.device
    f = lambda t: t.detach().to(torch.float32).contiguous()
    xyzs, rotations, scalings, opacities = f(xyzs), f(rotations), f(scalings), f(opacities)
    P = xyzs.shape[0]
    coords = torch.linspace(-bbox_scale, bbox_scale, resolution).to(dev)   # computed like the reference, on the host
    occ = torch.empty([resolution] * 3, dtype=torch.float32, device=dev)
    scratch = torch.empty(L.dgm_opacity_field_scratch_bytes(P), dtype=torch.uint8, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    with _lib.device_guard(dev):
        _lib.check(L.dgm_opacity_field(P, vp(xyzs), vp(rotations), vp(scalings), vp(opacities), float(opacity_threshold),
                                       int(resolution), int(num_blocks), float(block_size * relax_ratio), vp(coords), vp(scratch),
                                       vp(occ), _lib.stream_ptr()))
    return occ
