"""Helpers for the `-m gpu` parity tests: run the HIP path through its public API / C ABI and expose its
internal state (dgm_describe_state) as numpy arrays for comparison with the oracle."""
import ctypes
import importlib

import numpy as np
import torch

from conftest import pkg


def t(a, dev="cuda"):
    if a is None:
        return torch.empty(0, device=dev)
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


def hip_forward(a, debug=False):
    """a: dict from conftest.raster_args -> dict with outputs + decoded internal state (numpy)."""
    R = pkg("rasterizer")
    L = pkg("_lib")
    P, W, H = a["means3D"].shape[0], a["W"], a["H"]
    tens = dict(bg=t(a["bg"]), means3D=t(a["means3D"]), colors=t(a["colors_precomp"]), opac=t(a["opacities"]),
                scales=t(a["scales"]), rots=t(a["rotations"]), cov=t(a["cov3D_precomp"]), vm=t(a["viewmatrix"]),
                pm=t(a["projmatrix"]), sh=t(a["sh"]), campos=t(a["campos"]))
    n, color, radii, geom, binning, img = R._C.rasterize_gaussians(
        tens["bg"], tens["means3D"], tens["colors"], tens["opac"], tens["scales"], tens["rots"], a["scale_modifier"],
        tens["cov"], tens["vm"], tens["pm"], a["tanfovx"], a["tanfovy"], H, W, tens["sh"], a["degree"], tens["campos"],
        False, debug)
    torch.cuda.synchronize()
    # (capacity-mode forwards -- rasterizer.SYNC_FREE -- lay the binning buffer out for their capacity, not for the frame's R)
    cap_entry = R._CAP_OF.get(binning.data_ptr()) if binning.numel() else None
    n_layout = cap_entry[0] if cap_entry is not None and cap_entry[1] == n and cap_entry[2] == binning.numel() else n
    lay = L.StateLayout()
    L.check(L.lib().dgm_describe_state(P, W, H, n_layout, ctypes.byref(lay)))
    out = dict(num_rendered=n, color=color.cpu().numpy(), radii=radii.cpu().numpy(), tensors=tens,
               buffers=(geom, binning, img), layout=lay)
    if P == 0:
        return out
    tiles = lay.tiles_x * lay.tiles_y

    def view(buf, off, dtype, count):
        base = buf.data_ptr()
        pad = (-base) % 256  # the library aligns the chunk it is handed to 256 B
        raw = buf.cpu().numpy()
        return np.frombuffer(raw.tobytes(), dtype=dtype, count=count, offset=pad + off).copy()

    rec = view(geom, lay.rec, np.float32, P * 12).reshape(P, 12)
    out.update(
        means2D=rec[:, 0:2].copy(), conic_opacity=np.stack([rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5]], 1),
        rgb=rec[:, 6:9].copy(), rect=rec[:, 9].copy().view(np.uint32), rec_offs=rec[:, 10].copy().view(np.uint32),
        depths=view(geom, lay.depth, np.float32, P), radii_int=view(geom, lay.radii, np.int32, P),
        tiles_touched=view(geom, lay.tiles_touched, np.uint32, P), offs=view(geom, lay.offs, np.uint32, P),
        cov3D=view(geom, lay.cov3D, np.float32, P * 6).reshape(P, 6), clamped=view(geom, lay.clamped, np.uint8, P),
        counters=view(geom, lay.counters, np.uint32, 8), uctl=view(geom, lay.counters + 32, np.uint32, 64),
        tile_order=view(geom, lay.tile_count, np.uint32, tiles),
        nproc=view(img, lay.nproc, np.uint32, tiles), ulist_last=view(img, lay.ulist_last, np.uint32, tiles * 4).reshape(tiles, 4),
        final_T=view(img, lay.final_T, np.float32, W * H).reshape(H, W),
        n_contrib=view(img, lay.n_contrib, np.uint32, W * H).reshape(H, W),
        ranges=view(img, lay.ranges, np.uint32, tiles * 2).reshape(tiles, 2))
    if n > 0:
        ulog = 5 if n_layout < (1 << 20) else 6
        cap = (n_layout >> ulog) + 1
        point_list = view(binning, lay.point_list, np.uint32, n)
        # the gradient row of every list entry as render_bwd4 forms it (render_common.hpp instance_row): offs[g] + the position of the
        # entry's tile in g's rectangle, row by row
        tile_of = np.zeros(n, np.int64)
        for tl, (a0, a1) in enumerate(out["ranges"]):
            tile_of[a0:a1] = tl
        rect = out["rect"][point_list].astype(np.int64)
        xmin, ymin, w = rect & 1023, (rect >> 10) & 1023, rect >> 20
        upos = out["rec_offs"][point_list].astype(np.int64) + (tile_of // lay.tiles_x - ymin) * w + (tile_of % lay.tiles_x - xmin)
        out.update(point_list=point_list, upos=upos.astype(np.uint32),
                   ulist_full=view(binning, lay.ulist_full, np.uint32, cap * 4).reshape(cap, 4), unit_log2=ulog)
    else:
        out.update(point_list=np.zeros(0, np.uint32), upos=np.zeros(0, np.uint32))
    return out


def hip_backward(a, fwd, dL, debug=False):
    R = pkg("rasterizer")
    tn = fwd["tensors"]
    geom, binning, img = fwd["buffers"]
    radii = torch.as_tensor(fwd["radii"], device="cuda")
    g = R._C.rasterize_gaussians_backward(
        tn["bg"], tn["means3D"], radii, tn["colors"], tn["scales"], tn["rots"], a["scale_modifier"], tn["cov"],
        tn["vm"], tn["pm"], a["tanfovx"], a["tanfovy"], t(dL), tn["sh"], a["degree"], tn["campos"], geom,
        fwd["num_rendered"], binning, img, debug)
    torch.cuda.synchronize()
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations"]
    return {k: v.cpu().numpy() for k, v in zip(names, g)}


def rel_to_max(mine, ref):
    scale = np.abs(ref).max() + 1e-30
    return float(np.abs(mine.astype(np.float64) - ref.astype(np.float64)).max() / scale)
