"""Tile-list length histogram of the bench scene after n train steps: python tools/tile_hist.py cfg4 240"""
import ctypes
import importlib
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 240
bench.WORKLOAD = wl
dev = torch.device("cuda", 0)
tr, (P, W, H) = bench.build_scene(dev, 0, 1, "hip")
it0 = tr.opt.warm_up + 2000
for i in range(steps):
    tr.step(it0 + i)
torch.cuda.synchronize()
g = tr.g
R = importlib.import_module("dg-mesh_amd.rasterizer")
L = importlib.import_module("dg-mesh_amd._lib")
cam = tr.cameras[0]
with torch.no_grad():
    sh = g.get_features.contiguous()
    n, color, radii, geom, binning, img = R._C.rasterize_gaussians(
        tr.bg, g.get_xyz.contiguous(), torch.empty(0, device=dev), g.get_opacity.contiguous(), g.get_scaling.contiguous(),
        g.get_rotation.contiguous(), 1.0, torch.empty(0, device=dev), cam.world_view_transform, cam.full_proj_transform,
        math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, sh, g.active_sh_degree, cam.camera_center, False, False)
torch.cuda.synchronize()
lay = L.StateLayout()
L.check(L.lib().dgm_describe_state(P, W, H, n, ctypes.byref(lay)))
tiles = lay.tiles_x * lay.tiles_y
raw = img.cpu().numpy()
pad = (-img.data_ptr()) % 256
rg = np.frombuffer(raw.tobytes(), dtype=np.uint32, count=tiles * 2, offset=pad + lay.ranges).reshape(tiles, 2)
ln = (rg[:, 1] - rg[:, 0]).astype(np.int64)
print(wl, "after", steps, "steps: R", n, "tiles", tiles, "min/median/mean/max", ln.min(), int(np.median(ln)), int(ln.mean()), ln.max())
for lo, hi in [(0, 2048), (2049, 4096), (4097, 8192), (8193, 16384), (16385, 65536), (65537, 10 ** 9)]:
    m = (ln >= lo) & (ln <= hi)
    print(f"  {lo:>6}..{hi:<10} tiles {m.sum():>6}  entries {ln[m].sum():>10}")
sc = g.get_scaling
print("scaling mean/max", float(sc.mean()), float(sc.max()), "opacity mean", float(g.get_opacity.mean()))
# the same frame through the trainer's own path (deformation applied)
losses, pkg = tr.loss_terms(cam, it0 + steps)
torch.cuda.synchronize()
print("through loss_terms: keys", sorted(pkg.keys()))
for k in ("d_xyz", "d_rotation", "d_scaling", "means3D", "scales"):
    if k in pkg and torch.is_tensor(pkg[k]):
        v = pkg[k].detach()
        print("  ", k, "abs mean", float(v.abs().mean()), "abs max", float(v.abs().max()))
print("  radii>0", int((pkg["radii"] > 0).sum()), "radii mean", float(pkg["radii"][pkg["radii"] > 0].float().mean()), "max", int(pkg["radii"].max()))
with torch.no_grad():
    N = g.get_xyz.shape[0]
    ti = tr.time_input(cam, N, it0 + steps)
    print("time_input", tuple(ti.shape), float(ti.min()), float(ti.max()))
    raw = tr.deform.step_raw(g.get_xyz.detach(), ti)
    raw = raw[0] if isinstance(raw, (tuple, list)) else raw
    print("raw head output", tuple(raw.shape), "abs mean per column", [round(float(x), 5) for x in raw.abs().mean(0)])
    tr.deform.net.trunk_impl = "torch" if hasattr(tr.deform.net, "trunk_impl") else None
    try:
        ref = tr.deform.net(g.get_xyz.detach(), ti)
        print("torch trunk outputs abs mean", [round(float(o.abs().mean()), 5) for o in ref])
    except Exception as e:  # noqa
        print("torch trunk call failed:", e)
with torch.no_grad():
    n2, color, radii, geom, binning, img = R._C.rasterize_gaussians(
        tr.bg, (g.get_xyz + raw[:, 0:3]).contiguous(), torch.empty(0, device=dev), g.get_opacity.contiguous(),
        (g.get_scaling + raw[:, 7:10]).contiguous(), (g.get_rotation + raw[:, 3:7]).contiguous(), 1.0, torch.empty(0, device=dev),
        cam.world_view_transform, cam.full_proj_transform, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), H, W, sh,
        g.active_sh_degree, cam.camera_center, False, False)
torch.cuda.synchronize()
L.check(L.lib().dgm_describe_state(P, W, H, n2, ctypes.byref(lay)))
raw_img = img.cpu().numpy()
pad = (-img.data_ptr()) % 256
rg = np.frombuffer(raw_img.tobytes(), dtype=np.uint32, count=tiles * 2, offset=pad + lay.ranges).reshape(tiles, 2)
ln = (rg[:, 1] - rg[:, 0]).astype(np.int64)
print("DEFORMED: R", n2, "min/median/mean/max", ln.min(), int(np.median(ln)), int(ln.mean()), ln.max())
for lo, hi in [(0, 2048), (2049, 4096), (4097, 8192), (8193, 16384), (16385, 65536), (65537, 10 ** 9)]:
    m = (ln >= lo) & (ln <= hi)
    print(f"  {lo:>6}..{hi:<10} tiles {m.sum():>6}  entries {ln[m].sum():>10}")
