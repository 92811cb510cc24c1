"""Rasterizer-only micro-benchmark: per-stage device time (hipEvents inside the library) on a BASELINE config.
Usage: python tools/raster_bench.py [cfg2] [--kind init|trained] [--iters 20]"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("dg-mesh_amd.synthetic")
L = importlib.import_module("dg-mesh_amd._lib")
R = importlib.import_module("dg-mesh_amd.rasterizer")
from simple_knn._C import distCUDA2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cfg", nargs="?", default="cfg2")
    ap.add_argument("--kind", default="init")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--profile", type=int, default=1)
    args = ap.parse_args()
    c = syn.CONFIGS[args.cfg]
    P, W, H = c["P"], c["W"], c["H"]
    dev = "cuda"
    rng = np.random.RandomState(0)
    if args.kind == "trained":
        g = syn.make_gaussians(P, seed=0, kind="trained", dist2=np.full(P, 1e-4, np.float32))
    else:
        xyz = ((rng.rand(P, 3) * 2 - 1) * c.get("extent", 1.3)).astype(np.float32)
        t0 = time.time()
        d2 = distCUDA2(torch.tensor(xyz, device=dev))
        torch.cuda.synchronize()
        t1 = time.time()
        d2 = distCUDA2(torch.tensor(xyz, device=dev)).cpu().numpy()
        torch.cuda.synchronize()
        print(f"knn P={P}: first {1e3*(t1-t0):.2f} ms, second {1e3*(time.time()-t1):.2f} ms", flush=True)
        g = syn.make_gaussians(P, seed=0, kind="init", dist2=d2, extent=c.get("extent", 1.3))
        g["xyz"] = xyz
    a = syn.activate(g)
    cam = syn.config_camera(args.cfg, frame=3)
    T = lambda x: torch.tensor(x, device=dev)
    bg = T(np.ones(3, np.float32) if c["white_bg"] else np.zeros(3, np.float32))
    means3D, opac, scales, rots, sh = T(a["means3D"]), T(a["opacities"]), T(a["scales"]), T(a["rotations"]), T(a["shs"])
    vm, pm, campos = T(cam.world_view_transform), T(cam.full_proj_transform), T(cam.camera_center)
    tanx, tany = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    e = torch.empty(0, device=dev)
    dL = torch.randn(3, H, W, device=dev)
    L.lib().dgm_set_profiling(args.profile)
    acc = {}
    wall = []
    for it in range(args.iters + 3):
        torch.cuda.synchronize()
        t0 = time.time()
        n, color, radii, geom, binning, img = R._C.rasterize_gaussians(bg, means3D, e, opac, scales, rots, 1.0, e, vm, pm,
                                                                      tanx, tany, H, W, sh, 3, campos, False, False)
        fw = L.stage_ms() if args.profile else {}
        grads = R._C.rasterize_gaussians_backward(bg, means3D, radii, e, scales, rots, 1.0, e, vm, pm, tanx, tany, dL, sh,
                                                  3, campos, geom, n, binning, img, False)
        bw = L.stage_ms() if args.profile else {}
        torch.cuda.synchronize()
        if it >= 3:
            wall.append(time.time() - t0)
            for k, v in fw.items():
                if not k.endswith("bwd"):
                    acc.setdefault(k, []).append(v)
            for k, v in bw.items():
                if k.endswith("bwd"):
                    acc.setdefault(k, []).append(v)
    res = {k: float(np.median(v)) for k, v in acc.items()}
    res["wall_ms_fwd_bwd"] = float(np.median(wall) * 1e3)
    res.update(cfg=args.cfg, kind=args.kind, P=P, W=W, H=H, R=int(n), vis=float((radii > 0).float().mean()),
               meanT=float(0))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
