"""autograd.Functions over the per-Gaussian glue kernels (dgm_gaussian_apply_*, dgm_cycle_loss_*).

They replace, on the GPU, the one-op-per-kernel PyTorch expressions of the reference's render() prologue
(R/gaussian_renderer/__init__.py:77-95: get_xyz + d_xyz, get_scaling + d_scaling, get_rotation + d_rotation,
get_opacity) and of the cycle-consistency loss (R/train.py:221-238), and consume the deformation network's raw
(N, n_out) output directly, so the column slicing of that output needs no backward kernels either.
"""
import ctypes

import torch

from . import _lib


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return _lib.stream_ptr()


def _f32c(t, name):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"{name}: fp32 device tensor required (dg-mesh_amd has no CPU path for its kernels)")
    return t if t.is_contiguous() else t.contiguous()


class _GaussianApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, scaling, rotation, opacity, delta):
        L = _lib.lib()
        xyz, scaling, rotation = _f32c(xyz, "xyz"), _f32c(scaling, "scaling"), _f32c(rotation, "rotation")
        opacity, delta = _f32c(opacity, "opacity"), _f32c(delta, "delta")
        P, ld = xyz.shape[0], delta.shape[1]
        if delta.shape[0] != P or ld < 10:
            raise RuntimeError("gaussian_apply: delta must be (P, >= 10): [d_xyz | d_rotation | d_scaling | ...]")
        means, scales = torch.empty_like(xyz), torch.empty_like(scaling)
        rots, opac = torch.empty_like(rotation), torch.empty_like(opacity)
        with _lib.device_guard(xyz.device):
            _lib.check(L.dgm_gaussian_apply_forward(P, _vp(xyz), _vp(scaling), _vp(rotation), _vp(opacity), _vp(delta), ld,
                                                    _vp(means), _vp(scales), _vp(rots), _vp(opac), _stream()))
        ctx.save_for_backward(scaling, rotation, opacity)
        ctx.ld = ld
        return means, scales, rots, opac

    @staticmethod
    def backward(ctx, g_means, g_scales, g_rots, g_opac):
        L = _lib.lib()
        scaling, rotation, opacity = ctx.saved_tensors
        P, ld = scaling.shape[0], ctx.ld
        z = lambda g, like: torch.zeros_like(like) if g is None else _f32c(g, "grad")
        g_means, g_scales = z(g_means, scaling), z(g_scales, scaling)
        g_rots, g_opac = z(g_rots, rotation), z(g_opac, opacity)
        d_xyz, d_scaling = torch.empty_like(scaling), torch.empty_like(scaling)
        d_rotation, d_opacity = torch.empty_like(rotation), torch.empty_like(opacity)
        d_delta = torch.empty((P, ld), dtype=torch.float32, device=scaling.device)
        with _lib.device_guard(scaling.device):
            _lib.check(L.dgm_gaussian_apply_backward(P, _vp(scaling), _vp(rotation), _vp(opacity), _vp(g_means), _vp(g_scales),
                                                     _vp(g_rots), _vp(g_opac), _vp(d_xyz), _vp(d_scaling), _vp(d_rotation),
                                                     _vp(d_opacity), _vp(d_delta), ld, _stream()))
        return d_xyz, d_scaling, d_rotation, d_opacity, d_delta


def gaussian_apply(xyz, scaling, rotation, opacity, delta):
    """(means3D, scales, rotations, opacities) from the raw Gaussian parameters and the raw deformation output."""
    return _GaussianApply.apply(xyz, scaling, rotation, opacity, delta)


class _CycleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        L = _lib.lib()
        a, b = _f32c(a, "a"), _f32c(b, "b")
        if a.shape != b.shape or a.dim() != 2 or a.shape[1] < 10:
            raise RuntimeError("cycle_loss: two (N, >= 10) tensors required")
        N, ld = a.shape
        ws = torch.empty(L.dgm_cycle_loss_workspace_bytes(N), dtype=torch.uint8, device=a.device)
        out = torch.empty(4, dtype=torch.float32, device=a.device)
        with _lib.device_guard(a.device):
            _lib.check(L.dgm_cycle_loss_forward(N, _vp(a), _vp(b), ld, _vp(ws), _vp(out), _stream()))
        ctx.save_for_backward(a, b)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        a, b = ctx.saved_tensors
        N, ld = a.shape
        g = g.contiguous().float().reshape(1)
        d_a, d_b = torch.empty_like(a), torch.empty_like(b)
        with _lib.device_guard(a.device):
            _lib.check(L.dgm_cycle_loss_backward(N, _vp(a), _vp(b), ld, _vp(g), _vp(d_a), _vp(d_b), _stream()))
        return d_a, d_b


def cycle_loss(delta, delta_back):
    """(l1(-b_xyz, d_xyz) + l1(-b_rot, d_rot) + l1(-b_scale, d_scale)) / 3 on the raw head outputs."""
    return _CycleLoss.apply(delta, delta_back)


class _Se3Exp(torch.autograd.Function):
    """(N, >= 6) raw head outputs [w | v | ...] -> (N, 4, 4) rigid transforms (R/utils/time_utils.py:116-123 + rigid_utils.exp_se3)."""

    @staticmethod
    def forward(ctx, o):
        L = _lib.lib()
        o = _f32c(o, "screw rows")
        N, ld = o.shape
        if ld < 6:
            raise RuntimeError("se3_exp: rows need at least 6 columns (w, v)")
        T = torch.empty((N, 4, 4), dtype=torch.float32, device=o.device)
        with _lib.device_guard(o.device):
            _lib.check(L.dgm_se3_exp_forward(N, _vp(o), ld, _vp(T), _stream()))
        ctx.save_for_backward(o)
        return T

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dT):
        L = _lib.lib()
        (o,) = ctx.saved_tensors
        N, ld = o.shape
        dT = _f32c(dT, "grad")
        d_o = torch.zeros_like(o)  # (columns beyond w, v get no gradient from here)
        with _lib.device_guard(o.device):
            _lib.check(L.dgm_se3_exp_backward(N, _vp(o), ld, _vp(dT), _vp(d_o), ld, _stream()))
        return d_o


class _Se3Transform(torch.autograd.Function):
    """means3D = (T [xyz, 1])[:3] / (T [xyz, 1])[3]: the 6-DoF branch of render() (R/gaussian_renderer/__init__.py:68-75)."""

    @staticmethod
    def forward(ctx, T, xyz):
        L = _lib.lib()
        T, xyz = _f32c(T, "transforms"), _f32c(xyz, "xyz")
        N = xyz.shape[0]
        if tuple(T.shape) != (N, 4, 4):
            raise RuntimeError("se3_transform: transforms must be (N, 4, 4)")
        out = torch.empty_like(xyz)
        with _lib.device_guard(xyz.device):
            _lib.check(L.dgm_se3_transform_forward(N, _vp(T), _vp(xyz), _vp(out), _stream()))
        ctx.save_for_backward(T, xyz)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        L = _lib.lib()
        T, xyz = ctx.saved_tensors
        N = xyz.shape[0]
        g = _f32c(g, "grad")
        dT, d_xyz = torch.empty_like(T), torch.empty_like(xyz)
        with _lib.device_guard(xyz.device):
            _lib.check(L.dgm_se3_transform_backward(N, _vp(T), _vp(xyz), _vp(g), _vp(dT), _vp(d_xyz), _stream()))
        return dT, d_xyz


def se3_exp(o):
    """Raw (w, v) rows (first six columns of `o`) -> (N, 4, 4) transforms, on the HIP kernels."""
    return _Se3Exp.apply(o)


def se3_transform(T, xyz):
    return _Se3Transform.apply(T, xyz)
