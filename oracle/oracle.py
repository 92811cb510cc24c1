"""ctypes/numpy front-end of the CPU ORACLE (oracle/dgr_oracle.c, oracle/knn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under the product package may import this module.

`forward` / `backward` restate the host orchestration of the reference:
  DGR/cuda_rasterizer/rasterizer_impl.cu:198-336 (Rasterizer::forward),
  DGR/cuda_rasterizer/rasterizer_impl.cu:340-434 (Rasterizer::backward),
  DGR/rasterize_points.cu:35-114, 117-196 (output allocation / zero-init).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = ctypes.c_float
_i = ctypes.c_int
_p = ctypes.c_void_p


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, s) for s in ("dgr_oracle.c", "knn_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_inclusive_scan.restype = ctypes.c_uint32
        _LIB.orc_get_higher_msb.restype = ctypes.c_uint32
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n):
    """OpenMP threads used by the oracle (bench.py reports this as cpu_baseline.cores)."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(n))
    except OSError:
        pass


def preprocess_fwd(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
                   colors_precomp, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy):
    means3D, scales, rotations, opacities, shs = map(_f32, (means3D, scales, rotations, opacities, shs))
    cov3D_precomp, colors_precomp = _f32(cov3D_precomp), _f32(colors_precomp)
    viewmatrix, projmatrix, campos = _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    out = dict(
        radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), np.float32), depths=np.zeros(P, np.float32),
        cov3D=np.zeros((P, 6), np.float32), rgb=np.zeros((P, 3), np.float32),
        conic_opacity=np.zeros((P, 4), np.float32), tiles_touched=np.zeros(P, np.uint32),
        clamped=np.zeros((P, 3), np.uint8))
    lib().orc_preprocess_fwd(
        _i(P), _i(D), _i(M), _ptr(means3D), _ptr(scales), _f(scale_modifier), _ptr(rotations), _ptr(opacities),
        _ptr(shs), _ptr(cov3D_precomp), _ptr(colors_precomp), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos),
        _i(W), _i(H), _f(tanfovx), _f(tanfovy), _ptr(out["radii"]), _ptr(out["means2D"]), _ptr(out["depths"]),
        _ptr(out["cov3D"]), _ptr(out["rgb"]), _ptr(out["conic_opacity"]), _ptr(out["tiles_touched"]),
        _ptr(out["clamped"]))
    return out


def bin_tiles(P, W, H, geom):
    """inclusive scan -> duplicateWithKeys -> stable sort -> identifyTileRanges."""
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    offs = np.zeros(P, np.uint32)
    R = int(lib().orc_inclusive_scan(_i(P), _ptr(geom["tiles_touched"]), _ptr(offs))) if P else 0
    keys = np.zeros(max(R, 1), np.uint64)
    plist = np.zeros(max(R, 1), np.uint32)
    ranges = np.zeros((tiles, 2), np.uint32)
    lib().orc_bin(_i(P), _i(W), _i(H), _ptr(geom["means2D"]), _ptr(geom["depths"]), _ptr(geom["radii"]),
                  _ptr(offs), ctypes.c_uint32(R), _ptr(keys), _ptr(plist), _ptr(ranges))
    return dict(point_offsets=offs, num_rendered=R, keys=keys[:R], point_list=plist[:R], ranges=ranges)


def render_fwd(W, H, binning, means2D, colors, conic_opacity, bg):
    out_color = np.zeros((3, H, W), np.float32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.uint32)
    fragile = np.zeros((H, W), np.uint8)
    plist = binning["point_list"] if binning["num_rendered"] else np.zeros(1, np.uint32)
    colors, bg = _f32(colors), _f32(bg)
    lib().orc_render_fwd(_ptr(binning["ranges"]), _ptr(plist), _i(W), _i(H), _ptr(means2D), _ptr(colors),
                         _ptr(conic_opacity), _ptr(bg), _ptr(out_color), _ptr(final_T), _ptr(n_contrib),
                         _ptr(fragile))
    return dict(out_color=out_color, final_T=final_T, n_contrib=n_contrib, fragile=fragile)


def forward(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
            projmatrix, tanfovx, tanfovy, H, W, sh, degree, campos):
    """Argument order of _C.rasterize_gaussians (DGR/rasterize_points.h:18-38), numpy in / dict out."""
    means3D = _f32(means3D)
    P = means3D.shape[0]
    sh = None if sh is None or np.size(sh) == 0 else _f32(sh)
    colors_precomp = None if colors_precomp is None or np.size(colors_precomp) == 0 else _f32(colors_precomp)
    cov3D_precomp = None if cov3D_precomp is None or np.size(cov3D_precomp) == 0 else _f32(cov3D_precomp)
    scales = None if scales is None or np.size(scales) == 0 else _f32(scales)
    rotations = None if rotations is None or np.size(rotations) == 0 else _f32(rotations)
    M = 0 if sh is None else sh.shape[1]
    if P == 0:
        return dict(num_rendered=0, color=np.zeros((3, H, W), np.float32), radii=np.zeros(0, np.int32))
    geom = preprocess_fwd(P, degree, M, means3D, scales, scale_modifier, rotations, opacities, sh, cov3D_precomp,
                          colors_precomp, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy)
    binning = bin_tiles(P, W, H, geom)
    feat = colors_precomp if colors_precomp is not None else geom["rgb"]
    img = render_fwd(W, H, binning, geom["means2D"], feat, geom["conic_opacity"], bg)
    return dict(num_rendered=binning["num_rendered"], color=img["out_color"], radii=geom["radii"], geom=geom,
                binning=binning, img=img)


def backward(fwd, bg, means3D, colors_precomp, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
             projmatrix, tanfovx, tanfovy, dL_dout_color, sh, degree, campos):
    """Mirrors _C.rasterize_gaussians_backward (DGR/rasterize_points.h:40-62); `fwd` = forward()'s dict."""
    means3D = _f32(means3D)
    P = means3D.shape[0]
    dL = _f32(dL_dout_color)
    H, W = dL.shape[1], dL.shape[2]
    sh = None if sh is None or np.size(sh) == 0 else _f32(sh)
    colors_precomp = None if colors_precomp is None or np.size(colors_precomp) == 0 else _f32(colors_precomp)
    cov3D_precomp = None if cov3D_precomp is None or np.size(cov3D_precomp) == 0 else _f32(cov3D_precomp)
    scales = None if scales is None or np.size(scales) == 0 else _f32(scales)
    rotations = None if rotations is None or np.size(rotations) == 0 else _f32(rotations)
    viewmatrix, projmatrix, campos, bg = map(_f32, (viewmatrix, projmatrix, campos, bg))
    M = 0 if sh is None else sh.shape[1]
    g = dict(dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dmeans2D=np.zeros((P, 3), np.float32),
             dL_dcolors=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
             dL_dopacity=np.zeros((P, 1), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
             dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
             dL_drotations=np.zeros((P, 4), np.float32))
    if P == 0:
        return g
    geom, binning, img = fwd["geom"], fwd["binning"], fwd["img"]
    feat = colors_precomp if colors_precomp is not None else geom["rgb"]
    cov3D = cov3D_precomp if cov3D_precomp is not None else geom["cov3D"]
    plist = binning["point_list"] if binning["num_rendered"] else np.zeros(1, np.uint32)
    lib().orc_render_bwd(_i(P), _ptr(binning["ranges"]), _ptr(plist), _i(W), _i(H), _ptr(bg), _ptr(geom["means2D"]),
                         _ptr(geom["conic_opacity"]), _ptr(feat), _ptr(img["final_T"]), _ptr(img["n_contrib"]),
                         _ptr(dL), _ptr(g["dL_dmeans2D"]), _ptr(g["dL_dconic"]), _ptr(g["dL_dopacity"]),
                         _ptr(g["dL_dcolors"]))
    focal_y = np.float32(H) / (np.float32(2.0) * np.float32(tanfovy))
    focal_x = np.float32(W) / (np.float32(2.0) * np.float32(tanfovx))
    lib().orc_cov2d_bwd(_i(P), _ptr(means3D), _ptr(geom["radii"]), _ptr(cov3D), _f(focal_x), _f(focal_y),
                        _f(tanfovx), _f(tanfovy), _ptr(viewmatrix), _ptr(g["dL_dconic"]), _ptr(g["dL_dmeans3D"]),
                        _ptr(g["dL_dcov3D"]))
    lib().orc_preprocess_bwd(_i(P), _i(degree), _i(M), _ptr(means3D), _ptr(geom["radii"]), _ptr(sh),
                             _ptr(geom["clamped"]), _ptr(scales), _ptr(rotations), _f(scale_modifier), _ptr(projmatrix),
                             _ptr(campos), _ptr(g["dL_dmeans2D"]), _ptr(g["dL_dmeans3D"]), _ptr(g["dL_dcolors"]),
                             _ptr(g["dL_dcov3D"]), _ptr(g["dL_dsh"]), _ptr(g["dL_dscales"]), _ptr(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    if P:
        lib().orc_mark_visible(_i(P), _ptr(means3D), _ptr(_f32(viewmatrix)), _ptr(_f32(projmatrix)), _ptr(out))
    return out.astype(bool)


def knn(points, brute=False):
    points = _f32(points)
    P = points.shape[0]
    out = np.zeros(P, np.float32)
    fn = lib().orc_knn_brute if brute else lib().orc_knn
    fn(_i(P), _ptr(points), _ptr(out))
    return out
