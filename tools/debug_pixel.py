import sys, os, importlib, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import raster_args, oracle_forward
import gpu_util as G
from oracle import oracle as orc
syn = importlib.import_module("dg-mesh_amd.synthetic")
kind, P, W, H, seed = ("trained", 4000, 160, 160, 3)
a = raster_args(syn, P, W, H, seed=seed, kind=kind)
fh = G.hip_forward(a)
fo = oracle_forward(orc, a)
img = fo["img"]
bad = (fh["n_contrib"] != img["n_contrib"]) & (img["fragile"] == 0)
ys, xs = np.nonzero(bad)
gx = (W + 15) // 16
print("bad quadrants:", sorted(set((int(y) // 8, int(x) // 8) for y, x in zip(ys, xs))))
def blend(px, py, lst, geom):
    T = 1.0; last = 0; contribs = []
    for k, g in enumerate(lst):
        x, y = geom["means2D"][g]; a_, b_, c_, o = geom["conic_opacity"][g]
        dx, dy = x - px, y - py
        power = -0.5 * (a_ * dx * dx + c_ * dy * dy) - b_ * dx * dy
        if power > 0: continue
        al = min(0.99, o * math.exp(power))
        if al < 1 / 255: continue
        if T * (1 - al) < 1e-4: break
        contribs.append((k + 1, round(al, 4))); T *= (1 - al); last = k + 1
    return T, last, contribs
for (y, x) in list(zip(ys, xs))[:1]:
    tile = (y // 16) * gx + x // 16
    r0, r1 = fo["binning"]["ranges"][tile]
    lst = fo["binning"]["point_list"][r0:r1]
    print("pixel", x, y, "tile", tile, "n", r1 - r0, "hip n_contrib", fh["n_contrib"][y, x], "T", fh["final_T"][y, x], "orc", img["n_contrib"][y, x], img["final_T"][y, x])
    for (dxo, dyo) in [(0, 0), (-8, 0), (8, 0), (0, -8), (0, 8), (-8, -8), (8, 8), (-16, 0), (0, -16), (16, 0), (0, 16)]:
        T, last, c = blend(x + dxo, y + dyo, lst, fo["geom"])
        print("   offset", dxo, dyo, "-> T", round(T, 5), "last", last, "ncontrib", len(c))
    # same pixel evaluated against the lists of neighbouring tiles
    for dt in (-gx, -1, 1, gx):
        t2 = tile + dt
        if 0 <= t2 < len(fo["binning"]["ranges"]):
            q0, q1 = fo["binning"]["ranges"][t2]
            T, last, c = blend(x, y, fo["binning"]["point_list"][q0:q1], fo["geom"])
            print("   with list of tile", t2, "n", q1 - q0, "-> T", round(T, 5), "last", last)
    print("  hip T row:", fh["final_T"][y, (x // 8) * 8:(x // 8) * 8 + 8])
    print("  orc T row:", img["final_T"][y, (x // 8) * 8:(x // 8) * 8 + 8])
