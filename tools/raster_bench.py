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
    ap.add_argument("--trace-fwd", action="store_true", help="per-wave timeline of render_fwd (needs a library built with -DRF_TRACE=1)")
    ap.add_argument("--trace", action="store_true", help="per-wave timeline of render_bwd4 (needs a library built with -DRB4_TRACE=1)")
    ap.add_argument("--hist", action="store_true", help="tile-list length / replay-bound statistics of the frame")
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
    if args.profile:  # the wall time of a forward + backward is taken WITHOUT the per-stage events (reading them synchronises)
        L.lib().dgm_set_profiling(0)
        wall = []
        for it in range(args.iters + 3):
            torch.cuda.synchronize()
            t0 = time.time()
            n, color, radii, geom, binning, img = R._C.rasterize_gaussians(bg, means3D, e, opac, scales, rots, 1.0, e, vm, pm,
                                                                          tanx, tany, H, W, sh, 3, campos, False, False)
            R._C.rasterize_gaussians_backward(bg, means3D, radii, e, scales, rots, 1.0, e, vm, pm, tanx, tany, dL, sh,
                                              3, campos, geom, n, binning, img, False)
            torch.cuda.synchronize()
            if it >= 3:
                wall.append(time.time() - t0)
        L.lib().dgm_set_profiling(args.profile)
    res["wall_ms_fwd_bwd"] = float(np.median(wall) * 1e3)
    res.update(cfg=args.cfg, kind=args.kind, P=P, W=W, H=H, R=int(n), vis=float((radii > 0).float().mean()),
               meanT=float(0))
    print(json.dumps(res), flush=True)
    if args.trace_fwd:
        import ctypes
        fn = L.lib().dgm_debug_rf_trace
        fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int], ctypes.c_int
        buf = np.zeros(4 * 65536 + 1, np.uint64)
        fn(buf.ctypes.data, buf.nbytes, 1)
        R._C.rasterize_gaussians(bg, means3D, e, opac, scales, rots, 1.0, e, vm, pm, tanx, tany, H, W, sh, 3, campos, False, False)
        fn(buf.ctypes.data, buf.nbytes, 0)
        cnt = min(int(buf[-1]), 65536)
        t = buf[:4 * cnt].reshape(cnt, 4)
        w0 = (t[:, 0] & ((1 << 48) - 1)).astype(np.int64)
        wd = (t[:, 0] >> 48).astype(np.int64)                       # lifetime on the 100 MHz clock
        hw, xcc = (t[:, 1] & 0xffffffff).astype(np.int64), (t[:, 1] >> 32).astype(np.int64) & 15
        dur, first = (t[:, 2] >> 32).astype(np.int64), (t[:, 2] & 0xffffffff).astype(np.int64)
        nlist, tested = (t[:, 3] >> 32).astype(np.int64), ((t[:, 3] >> 8) & 0xffffff).astype(np.int64)
        base = w0.min()
        us = lambda v: round(float(v) * 0.01, 2)
        pc = lambda v, ps=(10, 50, 90, 100): [int(np.percentile(v, p)) for p in ps]
        print("fwd waves", cnt, "| kernel span", us((w0 + wd).max() - base), "us | starts p50/p90/p99/max", [us(np.percentile(w0 - base, p)) for p in (50, 90, 99, 100)],
              "| ends p10/p50/p90", [us(np.percentile(w0 + wd - base, p)) for p in (10, 50, 90)])
        print("wave lifetime us p10/p50/p90/max", [us(np.percentile(wd, p)) for p in (10, 50, 90, 100)], "| shader cycles p10/p50/p90/max", pc(dur),
              "| cycles before the first blend p10/p50/p90/max", pc(first[first > 0]) if (first > 0).any() else None)
        loop16 = ((t[:, 1] >> 36) & 0xffff).astype(np.int64) * 16     # (render_fwd_async_kernel: cycles from entry to the end of the blend loop)
        busy = np.argsort(dur)[-40:]
        print("the 40 longest waves: lifetime cycles", pc(dur[busy], (0, 50, 100)), "| loop cycles", pc(loop16[busy], (0, 50, 100)), "| list length", pc(nlist[busy], (0, 50, 100)),
              "| tested", pc(tested[busy], (0, 50, 100)), "| cycles per tested entry", pc((loop16[busy] - first[busy]) / np.maximum(tested[busy], 1), (0, 50, 100)),
              "| cycles per LIST entry", pc(loop16[busy] / np.maximum(nlist[busy], 1), (0, 50, 100)))
        print("list length p10/p50/p90/max", pc(nlist), "| entries a wave tested p10/p50/p90/max", pc(tested),
              "| cycles per tested entry (after the first blend) p10/p50/p90", pc(((dur - first)[tested > 8] / tested[tested > 8]), (10, 50, 90)) if (tested > 8).any() else None)
        simd, cu, sh_, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        key = (((xcc * 8 + se) * 2 + sh_) * 16 + cu) * 4 + simd
        uniq, inv = np.unique(key, return_inverse=True)
        per = np.bincount(inv)
        work = np.bincount(inv, weights=tested)
        endt = np.zeros(len(uniq)); np.maximum.at(endt, inv, w0 + wd - base)
        print("SIMDs", len(uniq), "| waves per SIMD p10/p50/p90/max", pc(per), "| tested entries per SIMD p10/p50/p90/max", pc(work),
              "| last end per SIMD (us) p10/p50/p90/max", [us(np.percentile(endt, p)) for p in (10, 50, 90, 100)],
              "| corr(end, work)", round(float(np.corrcoef(endt, work)[0, 1]), 3))
        order = np.argsort(w0)
        late = order[-max(1, cnt // 20):]
        print("the last 5 % of the waves to start: list length p50", int(np.median(nlist[late])), "tested p50", int(np.median(tested[late])),
              "lifetime us p50", us(np.median(wd[late])), "| first 5 %: list length p50", int(np.median(nlist[order[:max(1, cnt // 20)]])))
    if args.trace:
        import ctypes
        fn = L.lib().dgm_debug_rb4_trace
        fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int], ctypes.c_int
        buf = np.zeros(4 * 65536 + 1, np.uint64)
        fn(buf.ctypes.data, buf.nbytes, 1)
        n2, color, radii, geom, binning, img = R._C.rasterize_gaussians(bg, means3D, e, opac, scales, rots, 1.0, e, vm, pm,
                                                                       tanx, tany, H, W, sh, 3, campos, False, False)
        R._C.rasterize_gaussians_backward(bg, means3D, radii, e, scales, rots, 1.0, e, vm, pm, tanx, tany, dL, sh, 3, campos,
                                          geom, n2, binning, img, False)
        fn(buf.ctypes.data, buf.nbytes, 0)
        cnt = int(buf[-1])
        t = buf[:4 * cnt].reshape(cnt, 4)
        # start / end on the constant 100 MHz clock (10 ns), duration in shader cycles (s_memtime is per CU: differences only)
        w0, w1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
        hw, xcc = (t[:, 2] & 0xffffffff).astype(np.int64), (t[:, 2] >> 32).astype(np.int64) & 15
        # (round 5: one record per PERSISTENT wave: its lifetime, the list entries it blended and the units it pulled)
        dur, blended = (t[:, 3] >> 32).astype(np.int64), (t[:, 3] >> 16).astype(np.int64) & 0xffff
        units = t[:, 3].astype(np.int64) & 0xffff
        simd, cu, sh_, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        base = w0.min()
        us = lambda v: round(float(v) * 0.01, 2)
        print("units per wave p10/p50/p90/max", [int(np.percentile(units, p)) for p in (10, 50, 90, 100)], "| waves without a unit", int((units == 0).sum()))
        print("waves", cnt, "| kernel span", us(w1.max() - base), "us | starts p50/p99/max", [us(np.percentile(w0 - base, p)) for p in (50, 99, 100)],
              "| ends p10/p50/p90", [us(np.percentile(w1 - base, p)) for p in (10, 50, 90)])
        print("wave duration (shader cycles) p10/p50/p90/max", [int(np.percentile(dur, p)) for p in (10, 50, 90, 100)], "| blended entries p50/max",
              int(np.median(blended)), int(blended.max()), "| cycles per blended entry p50", round(float(np.median(dur[blended > 8] / blended[blended > 8]))) if (blended > 8).any() else None)
        key = (((xcc * 8 + se) * 2 + sh_) * 16 + cu) * 4 + simd
        uniq, inv = np.unique(key, return_inverse=True)
        per = np.bincount(inv)
        work = np.bincount(inv, weights=blended)
        endt = np.zeros(len(uniq)); np.maximum.at(endt, inv, w1 - base)
        print("SIMDs that got a wave", len(uniq), "| waves per SIMD p10/p50/p90/max", [int(np.percentile(per, p)) for p in (10, 50, 90, 100)],
              "| blended entries per SIMD p10/p50/p90/max", [int(np.percentile(work, p)) for p in (10, 50, 90, 100)],
              "| last end per SIMD (us) p10/p50/p90/max", [us(np.percentile(endt, p)) for p in (10, 50, 90, 100)],
              "| entry spread max / p50", round(float(work.max() / max(np.median(work), 1.0)), 3))
        print("correlation(last end of a SIMD, its blended entries)", round(float(np.corrcoef(endt, work)[0, 1]), 3),
              "| waves per XCC", np.bincount(xcc, minlength=8).tolist())
    if args.hist:
        import ctypes
        lay = L.StateLayout()
        L.check(L.lib().dgm_describe_state(P, W, H, int(n), ctypes.byref(lay)))
        tiles = lay.tiles_x * lay.tiles_y
        raw = img.cpu().numpy().tobytes()
        pad = (-img.data_ptr()) % 256
        rg = np.frombuffer(raw, dtype=np.uint32, count=tiles * 2, offset=pad + lay.ranges).reshape(tiles, 2)
        ln = (rg[:, 1] - rg[:, 0]).astype(np.int64)
        npr = np.frombuffer(raw, dtype=np.uint32, count=tiles, offset=pad + lay.nproc).astype(np.int64)
        q = lambda v: [int(np.percentile(v, p)) for p in (50, 90, 99, 100)]
        print("tiles", tiles, "empty", int((ln == 0).sum()), "list length p50/p90/p99/max", q(ln), "replayed p50/p90/p99/max", q(npr),
              "sum replayed", int(npr.sum()), "units64", int(((npr + 63) // 64).sum()), "tiles > 4096:", int((ln > 4096).sum()))


if __name__ == "__main__":
    main()
