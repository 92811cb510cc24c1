"""Fused image loss (L1 + D-SSIM) on libdgmesh_hip.so: one forward and one backward kernel instead of the
PyTorch graph of R/utils/loss_utils.py:18-19, 32-76 (five grouped 11x11 convolutions + their autograd)."""
import ctypes

import torch

from . import _lib


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        L = _lib.lib()
        image, gt = image.contiguous(), gt.contiguous()
        C, H, W = image.shape
        ws = torch.empty(L.dgm_image_loss_workspace_bytes(C, H, W), dtype=torch.uint8, device=image.device)
        out = torch.empty(3, dtype=torch.float32, device=image.device)
        with _lib.device_guard(image.device):
            _lib.check(L.dgm_image_loss_forward(
                ctypes.c_void_p(image.data_ptr()), ctypes.c_void_p(gt.data_ptr()), C, H, W, float(lambda_dssim),
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                _lib.stream_ptr()))
        ctx.save_for_backward(image, gt, ws)
        ctx.lam = float(lambda_dssim)
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.lib()
        image, gt, ws = ctx.saved_tensors
        C, H, W = image.shape
        g = grad_out.contiguous().float().reshape(1)
        d_image = torch.empty_like(image)
        with _lib.device_guard(image.device):
            _lib.check(L.dgm_image_loss_backward(
                ctypes.c_void_p(image.data_ptr()), ctypes.c_void_p(gt.data_ptr()), C, H, W, ctx.lam,
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(d_image.data_ptr()),
                _lib.stream_ptr()))
        return d_image, None, None


def image_loss(image, gt, lambda_dssim=0.2):
    """(1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt)) for (C,H,W) fp32 GPU tensors."""
    if not image.is_cuda:
        raise RuntimeError("image_loss needs CUDA/HIP tensors (dg-mesh_amd has no CPU path for its kernels)")
    return _ImageLoss.apply(image, gt, lambda_dssim)
