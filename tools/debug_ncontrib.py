import sys, os, importlib, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import raster_args, oracle_forward
import gpu_util as G
from oracle import oracle as orc
syn = importlib.import_module("dg-mesh_amd.synthetic")
for kind, P, W, H, seed in [("init", 3000, 200, 136, 1), ("trained", 4000, 160, 160, 3)]:
    a = raster_args(syn, P, W, H, seed=seed, kind=kind)
    fh = G.hip_forward(a)
    fo = oracle_forward(orc, a)
    img = fo["img"]
    frag = img["fragile"]
    dn = fh["n_contrib"].astype(np.int64) - img["n_contrib"].astype(np.int64)
    bad = (dn != 0)
    print(kind, "pixels", H * W, "fragile", int((frag != 0).sum()), "n_contrib mismatches", int(bad.sum()), "of which non-fragile", int((bad & (frag == 0)).sum()))
    err = np.abs(fh["color"] - fo["color"]).max(0)
    print("  color err max all", err.max(), "non-fragile", err[frag == 0].max(), "T err", np.abs(fh["final_T"] - img["final_T"]).max())
    ys, xs = np.nonzero(bad & (frag == 0))
    for y, x in list(zip(ys, xs))[:12]:
        print("   px", x, y, "hip", fh["n_contrib"][y, x], "orc", img["n_contrib"][y, x], "T hip", fh["final_T"][y, x], "T orc", img["final_T"][y, x], "cerr", err[y, x])
    if len(ys):
        print("  dn hist", np.unique(dn[bad & (frag == 0)], return_counts=True))
        print("  x%16 hist", np.bincount(xs % 16, minlength=16), " y%16", np.bincount(ys % 16, minlength=16))
