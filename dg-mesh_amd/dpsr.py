"""Differentiable Poisson surface reconstruction (oriented points -> indicator grid) with the reference's module API:
    DPSR(res, sig=10, scale=True, shift=True)(V, N) -> phi          R/nvdiffrast_utils/dpsr.py:9-69
    point_rasterize(pts, vals, size), grid_interp(grid, pts)          R/nvdiffrast_utils/dpsr_utils.py:69-198
    laplace_regularizer_const(v_pos, t_pos_idx)                       R/nvdiffrast_utils/regularizer.py:40-60
(R/ = /root/reference/dgmesh/.)  The trilinear splat, the spectral solve between the two FFTs and the trilinear read-back
are HIP kernels of libdgmesh_hip (csrc/dpsr.hip) behind torch.autograd.Function, differentiable w.r.t. points and
normals; the FFTs are rocFFT through torch.fft on the current stream.  3-D only (the reference's only use), float32,
CUDA/HIP tensors only -- no CPU fallback.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


def _st():
    return _lib.stream_ptr()


def _chk_pts(pts):
    if not pts.is_cuda or pts.dtype != torch.float32 or pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("dpsr: points / values must be (n, 3) float32 CUDA/HIP tensors (dg-mesh_amd has no CPU path)")


class _Splat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, V, N, res):
        _chk_pts(V), _chk_pts(N)
        V, N = V.contiguous(), N.contiguous()
        grid = torch.empty((3, res, res, res), dtype=torch.float32, device=V.device)
        with _lib.device_guard(V.device):
            _lib.check(_lib.lib().dgm_dpsr_splat_forward(V.shape[0], res, _vp(V), _vp(N), _vp(grid), _st()))
        ctx.save_for_backward(V, N)
        ctx.res = res
        return grid

    @staticmethod
    def backward(ctx, dgrid):
        V, N = ctx.saved_tensors
        dgrid = dgrid.contiguous()
        dV, dN = torch.empty_like(V), torch.empty_like(N)
        with _lib.device_guard(V.device):
            _lib.check(_lib.lib().dgm_dpsr_splat_backward(V.shape[0], ctx.res, _vp(V), _vp(N), _vp(dgrid), _vp(dV), _vp(dN), _st()))
        return dV, dN, None


class _Interp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, phi, V):
        _chk_pts(V)
        phi, V = phi.contiguous(), V.contiguous()
        fv = torch.empty(V.shape[0], dtype=torch.float32, device=V.device)
        with _lib.device_guard(V.device):
            _lib.check(_lib.lib().dgm_dpsr_interp_forward(V.shape[0], phi.shape[0], _vp(phi), _vp(V), _vp(fv), _st()))
        ctx.save_for_backward(phi, V)
        return fv

    @staticmethod
    def backward(ctx, dfv):
        phi, V = ctx.saved_tensors
        dfv = dfv.contiguous()
        dphi, dV = torch.empty_like(phi), torch.empty_like(V)
        with _lib.device_guard(V.device):
            _lib.check(_lib.lib().dgm_dpsr_interp_backward(V.shape[0], phi.shape[0], _vp(phi), _vp(V), _vp(dfv), _vp(dphi), _vp(dV), _st()))
        return dphi, dV


class _Spectral(torch.autograd.Function):
    """Phi = sum_d (-i omega_d G / (Lap + 1e-6)) Nhat_d, Phi(0) = 0 on the half spectrum (dpsr.py:41-54)."""

    @staticmethod
    def forward(ctx, ras_s, res, sig):
        x = torch.view_as_real(ras_s.contiguous()).contiguous()          # (3, R, R, R/2+1, 2)
        out = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device)
        with _lib.device_guard(x.device):
            _lib.check(_lib.lib().dgm_dpsr_spectral(res, float(sig), _vp(x), _vp(out), 0, _st()))
        ctx.res, ctx.sig = res, float(sig)
        return torch.view_as_complex(out)

    @staticmethod
    def backward(ctx, dPhi):
        g = torch.view_as_real(dPhi.contiguous()).contiguous()
        out = torch.empty((3,) + tuple(g.shape), dtype=torch.float32, device=g.device)
        with _lib.device_guard(g.device):
            _lib.check(_lib.lib().dgm_dpsr_spectral(ctx.res, ctx.sig, _vp(g), _vp(out), 1, _st()))
        return torch.view_as_complex(out), None, None


def point_rasterize(pts, vals, size):
    """(batch, n, 3) points in (0, 1), (batch, n, 3) values -> (batch, 3, res, res, res)  (dpsr_utils.py:143-198)."""
    res = int(size[0])
    if tuple(int(s) for s in size) != (res, res, res):
        raise RuntimeError("dpsr: cubic 3-D grids only")
    return torch.stack([_Splat.apply(pts[b], vals[b], res) for b in range(pts.shape[0])], 0)


def grid_interp(grid, pts, batched=True):
    """grid (batch, res, res, res, 1), pts (batch, n, 3) -> (batch, n, 1)  (dpsr_utils.py:69-118; one feature)."""
    if not batched:
        grid, pts = grid.unsqueeze(0), pts.unsqueeze(0)
    if grid.shape[-1] != 1:
        raise RuntimeError("dpsr.grid_interp: one feature per cell (what DPSR uses)")
    out = torch.stack([_Interp.apply(grid[b, ..., 0], pts[b]) for b in range(pts.shape[0])], 0).unsqueeze(-1)
    return out if batched else out.squeeze(0)


class DPSR(nn.Module):
    def __init__(self, res, sig=10, scale=True, shift=True):
        super().__init__()
        self.res = tuple(int(r) for r in res)
        if len(self.res) != 3 or len(set(self.res)) != 1:
            raise RuntimeError("DPSR: cubic 3-D grids only")
        self.sig, self.dim, self.scale, self.shift = sig, 3, scale, shift

    def forward(self, V, N):
        assert V.shape == N.shape
        R = self.res[0]
        outs = []
        for b in range(V.shape[0]):
            ras_p = _Splat.apply(V[b], N[b], R)                                   # (3, R, R, R)
            ras_s = torch.fft.rfftn(ras_p, dim=(1, 2, 3))                        # rocFFT
            Phi = _Spectral.apply(ras_s, R, self.sig)
            phi = torch.fft.irfftn(Phi, s=self.res, dim=(0, 1, 2))
            if self.shift or self.scale:
                if self.shift:  # offset so that the field is zero at the points on average (dpsr.py:58-61)
                    phi = phi - torch.mean(_Interp.apply(phi, V[b]))
                fv0 = phi[0, 0, 0]
                if self.scale:
                    phi = -phi / torch.abs(fv0) * 0.5
            outs.append(phi)
        return torch.stack(outs, 0)


class _Laplace(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v_pos, faces):
        L = _lib.lib()
        v = v_pos.contiguous()
        V, F = v.shape[0], faces.shape[0]
        scratch = torch.empty(int(L.dgm_laplace_scratch_floats(V)), dtype=torch.float32, device=v.device)
        loss = torch.empty(1, dtype=torch.float32, device=v.device)
        with _lib.device_guard(v.device):
            _lib.check(L.dgm_laplace_forward(V, F, _vp(v), _vp(faces), _vp(scratch), _vp(loss), _st()))
        ctx.save_for_backward(faces, scratch)
        ctx.V = V
        return loss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dloss):
        faces, scratch = ctx.saved_tensors
        dl = dloss.reshape(1).contiguous().float()
        dv = torch.empty((ctx.V, 3), dtype=torch.float32, device=scratch.device)
        with _lib.device_guard(scratch.device):
            _lib.check(_lib.lib().dgm_laplace_backward(ctx.V, faces.shape[0], _vp(faces), _vp(scratch), _vp(dl), _vp(dv), _st()))
        return dv, None


def laplace_regularizer_const(v_pos, t_pos_idx):
    """Umbrella-operator Laplacian regulariser of a triangle mesh (regularizer.py:40-60).  Device tensors run the HIP kernels
    (csrc/dpsr.hip: dgm_laplace_forward / backward); host tensors -- the CPU tests' path -- the reference's index-op formulation."""
    if v_pos.is_cuda:
        if v_pos.dtype != torch.float32 or v_pos.dim() != 2 or v_pos.shape[1] != 3:
            raise RuntimeError("laplace_regularizer_const: v_pos must be (V, 3) float32")
        faces = t_pos_idx.to(device=v_pos.device, dtype=torch.int32).contiguous()
        if faces.dim() != 2 or faces.shape[1] != 3:
            raise RuntimeError("laplace_regularizer_const: t_pos_idx must be (F, 3)")
        if faces.numel():  # the kernels scatter with atomicAdd at 3 * index: an index outside [0, V) would be a silent stray write
            lo, hi = torch.aminmax(faces)  # (one small reduction + read-back per call; the mesh branch, not the train step's hot loop)
            if int(lo) < 0 or int(hi) >= v_pos.shape[0]:
                raise IndexError(f"laplace_regularizer_const: face index out of range [0, {v_pos.shape[0]})")
        return _Laplace.apply(v_pos, faces)
    return _laplace_regularizer_torch(v_pos, t_pos_idx)


def _laplace_regularizer_torch(v_pos, t_pos_idx):
    """The reference's formulation in torch index ops (host tensors; also the GPU test's comparison)."""
    idx = t_pos_idx.long()
    term = torch.zeros_like(v_pos)
    norm = torch.zeros_like(v_pos[..., 0:1])
    v0, v1, v2 = v_pos[idx[:, 0]], v_pos[idx[:, 1]], v_pos[idx[:, 2]]
    term = term.index_add(0, idx[:, 0], (v1 - v0) + (v2 - v0))
    term = term.index_add(0, idx[:, 1], (v0 - v1) + (v2 - v1))
    term = term.index_add(0, idx[:, 2], (v0 - v2) + (v1 - v2))
    two = torch.full((idx.shape[0], 1), 2.0, dtype=v_pos.dtype, device=v_pos.device)
    for c in range(3):
        norm = norm.index_add(0, idx[:, c], two)
    term = term / torch.clamp(norm, min=1.0)
    return torch.mean(term ** 2)
