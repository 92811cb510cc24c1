"""TEST-ONLY loader for oracle/_ref/libref_raster*.so: the REFERENCE's own rasterizer / simple-knn kernels, built for
gfx950 from /root/reference by oracle/build_ref.sh (binaries only travel to the GPU box)."""
import ctypes
import os

import numpy as np
import torch

from conftest import ROOT

_c = ctypes
_vp, _i, _f = _c.c_void_p, _c.c_int, _c.c_float


def available(variant=""):
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", f"libref_raster{variant}.so"))


def _hip():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("HIP runtime not loaded")


_LIBS = {}


def lib(variant=""):
    if variant not in _LIBS:
        torch.zeros(1, device="cuda")  # make sure torch's HIP runtime is the one in the process
        L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libref_raster{variant}.so"))
        L.ref_forward.restype = _i
        L.ref_forward.argtypes = [_i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f,
                                  _vp, _vp, _vp]
        L.ref_backward.restype = None
        L.ref_backward.argtypes = [_i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _f,
                                   _vp] + [_vp] * 10
        _LIBS[variant] = L
    return _LIBS[variant]


def _dev_bytes(ptr, n):
    out = np.empty(n, np.uint8)
    hip = _hip()
    hip.hipMemcpy.argtypes = [_vp, _vp, _c.c_size_t, _i]
    assert hip.hipMemcpy(out.ctypes.data_as(_vp), _vp(ptr), n, 2) == 0
    return out


def _al(x, a=128):
    return (x + a - 1) // a * a


def t(a):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def p(x):
    return None if x is None else _vp(x.data_ptr())


def forward(a, variant=""):
    """a: dict of conftest.raster_args.  Returns outputs + decoded reference state (numpy)."""
    L = lib(variant)
    P, W, H = a["means3D"].shape[0], a["W"], a["H"]
    T = {k: t(a[k]) for k in ("bg", "means3D", "sh", "colors_precomp", "opacities", "scales", "rotations",
                              "cov3D_precomp", "viewmatrix", "projmatrix", "campos")}
    M = 0 if a["sh"] is None else a["sh"].shape[1]
    color = torch.zeros(3, H, W, device="cuda")
    radii = torch.zeros(P, dtype=torch.int32, device="cuda")
    state = (ctypes.c_ulonglong * 6)()
    n = L.ref_forward(P, a["degree"], M, p(T["bg"]), W, H, p(T["means3D"]), p(T["sh"]), p(T["colors_precomp"]),
                      p(T["opacities"]), p(T["scales"]), a["scale_modifier"], p(T["rotations"]), p(T["cov3D_precomp"]),
                      p(T["viewmatrix"]), p(T["projmatrix"]), p(T["campos"]), a["tanfovx"], a["tanfovy"], p(color),
                      p(radii), state)
    torch.cuda.synchronize()
    out = dict(num_rendered=n, color=color.cpu().numpy(), radii=radii.cpu().numpy(), tensors=T)
    # ImageState::fromChunk (rasterizer_impl.cu:172-179): accum_alpha, n_contrib, ranges, each 128-byte aligned
    N = W * H
    base = state[4]
    o_alpha = _al(base) - base
    o_ncon = _al(base + o_alpha + 4 * N) - base
    o_rng = _al(base + o_ncon + 4 * N) - base
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    img = _dev_bytes(base, o_rng + 8 * tiles)
    out["final_T"] = img[o_alpha:o_alpha + 4 * N].view(np.float32).reshape(H, W).copy()
    out["n_contrib"] = img[o_ncon:o_ncon + 4 * N].view(np.uint32).reshape(H, W).copy()
    out["ranges"] = img[o_rng:o_rng + 8 * tiles].view(np.uint32).reshape(tiles, 2).copy()
    # BinningState::fromChunk (rasterizer_impl.cu:181-194): point_list first
    if n > 0:
        b = state[2]
        o_pl = _al(b) - b
        out["point_list"] = _dev_bytes(b, o_pl + 4 * n)[o_pl:].view(np.uint32).copy()
    else:
        out["point_list"] = np.zeros(0, np.uint32)
    return out


def backward(a, fwd, dL, variant=""):
    L = lib(variant)
    P, W, H = a["means3D"].shape[0], a["W"], a["H"]
    T = fwd["tensors"]
    M = 0 if a["sh"] is None else a["sh"].shape[1]
    z = lambda *s: torch.zeros(s, device="cuda")
    g = dict(dL_dmeans2D=z(P, 3), dL_dconic=z(P, 4), dL_dopacity=z(P, 1), dL_dcolors=z(P, 3), dL_dmeans3D=z(P, 3),
             dL_dcov3D=z(P, 6), dL_dsh=z(P, max(M, 1), 3), dL_dscales=z(P, 3), dL_drotations=z(P, 4))
    radii = t(fwd["radii"])
    dLt = t(dL)
    L.ref_backward(P, a["degree"], M, fwd["num_rendered"], p(T["bg"]), W, H, p(T["means3D"]), p(T["sh"]),
                   p(T["colors_precomp"]), p(T["scales"]), a["scale_modifier"], p(T["rotations"]), p(T["cov3D_precomp"]),
                   p(T["viewmatrix"]), p(T["projmatrix"]), p(T["campos"]), a["tanfovx"], a["tanfovy"], p(radii), p(dLt),
                   p(g["dL_dmeans2D"]), p(g["dL_dconic"]), p(g["dL_dopacity"]), p(g["dL_dcolors"]), p(g["dL_dmeans3D"]),
                   p(g["dL_dcov3D"]), p(g["dL_dsh"]), p(g["dL_dscales"]), p(g["dL_drotations"]))
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in g.items()}
    if M == 0:
        out["dL_dsh"] = out["dL_dsh"][:, :0]
    return out


def knn(points):
    L = lib("")
    L.ref_knn.restype = None
    L.ref_knn.argtypes = [_i, _vp, _vp]
    pts = t(points.astype(np.float32))
    out = torch.zeros(points.shape[0], device="cuda")
    L.ref_knn(points.shape[0], p(pts), p(out))
    torch.cuda.synchronize()
    return out.cpu().numpy()
