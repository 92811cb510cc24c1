"""Shader clock and socket power while the rasterizer (forward + backward, tools/raster_bench.py's frames) runs back to back:
python tools/power_probe_raster.py [seconds=4]   -> gpurun_out/r05_power_raster_<kind>.json  (needs tools/bin/smi_sampler)"""
import csv
import importlib
import json
import math
import os
import subprocess
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

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
OUT = os.path.join(ROOT, "gpurun_out")
sampler = os.path.join(ROOT, "tools", "bin", "smi_sampler")
dev = "cuda"
c = syn.CONFIGS["cfg2"]
P, W, H = c["P"], c["W"], c["H"]
for kind in ("init", "trained"):
    rng = np.random.RandomState(0)
    if kind == "trained":
        g = syn.make_gaussians(P, seed=0, kind="trained", dist2=np.full(P, 1e-4, np.float32))
    else:
        xyz = ((rng.rand(P, 3) * 2 - 1) * 1.3).astype(np.float32)
        d2 = distCUDA2(torch.tensor(xyz, device=dev)).cpu().numpy()
        g = syn.make_gaussians(P, seed=0, kind="init", dist2=d2)
        g["xyz"] = xyz
    a = syn.activate(g)
    cam = syn.config_camera("cfg2", frame=3)
    T = lambda x: torch.tensor(x, device=dev)
    bg = T(np.ones(3, np.float32))
    m3, op, sc, ro, sh = T(a["means3D"]), T(a["opacities"]), T(a["scales"]), T(a["rotations"]), T(a["shs"])
    vm, pm, cp = T(cam.world_view_transform), T(cam.full_proj_transform), T(cam.camera_center)
    tx, ty = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    e = torch.empty(0, device=dev)
    dL = torch.randn(3, H, W, device=dev)

    def frame():
        n, color, radii, geom, binning, img = R._C.rasterize_gaussians(bg, m3, e, op, sc, ro, 1.0, e, vm, pm, tx, ty, H, W, sh, 3, cp, False, False)
        R._C.rasterize_gaussians_backward(bg, m3, radii, e, sc, ro, 1.0, e, vm, pm, tx, ty, dL, sh, 3, cp, geom, n, binning, img, False)
        return n
    for _ in range(5):
        n = frame()
    torch.cuda.synchronize()
    path = os.path.join(OUT, f"r05_power_raster_{kind}.csv")
    proc = subprocess.Popen([sampler, path, "20", str(seconds * 3 + 20)]) if os.path.exists(sampler) else None
    time.sleep(0.3)
    t0 = time.time()
    frames = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            frame()
        frames += 20
    torch.cuda.synchronize()
    dur = time.time() - t0
    time.sleep(0.2)
    rows = []
    if proc is not None:
        proc.terminate()
        proc.wait()
        with open(path) as fh:
            rows = [{k: float(v) for k, v in r.items()} for r in csv.DictReader(l for l in fh if not l.startswith("#"))]
    keep = [r for r in rows if 0.8 <= r["t_s"] <= 0.3 + dur]
    st = lambda k: None if not keep else {"mean": float(np.mean([r[k] for r in keep])), "min": float(np.min([r[k] for r in keep])), "max": float(np.max([r[k] for r in keep]))}
    out = {"workload": f"cfg2 {kind} frame, rasterizer forward + backward back to back (the forward's R read-back included: the GPU idles while the host turns around)",
           "R": int(n), "frames": frames, "ms_per_frame_wall": 1e3 * dur / frames, "samples": len(keep), "gfxclk_mhz": st("gfxclk_mhz"),
           "current_socket_power_w": st("cur_socket_w"), "gfx_activity": st("gfx_activity")}
    json.dump(out, open(os.path.join(OUT, f"r05_power_raster_{kind}.json"), "w"), indent=1)
    print(kind, out["ms_per_frame_wall"], out["gfxclk_mhz"], out["current_socket_power_w"], out["gfx_activity"], flush=True)
