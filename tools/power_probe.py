"""Shader clock and socket power while ONE MLP kernel runs back to back (put the "power-limited" claim on record).

    bash tools/build_variant.sh probe "-DP4_PROBE=1"          # dg-mesh_amd/lib/variants/probe.so (dgm_p4_probe, variant-only)
    gcc -O2 tools/smi_sampler.c -I/opt/rocm/include -L/opt/rocm/lib -lrocm_smi64 -Wl,-rpath,/opt/rocm/lib -o tools/bin/smi_sampler
    python tools/power_probe.py [seconds=5] [N=100000]        # on the GPU box -> gpurun_out/r05_power_<kind>_<operands>.json

For each of {forward layer GEMM, weight gradient alone, paired backward launch, backward-data GEMM alone} x {random, all-zero
operands}: launch the kernel in a loop for >= `seconds`, sample the SMU's gpu_metrics (tools/smi_sampler.c, 20 ms period: gfxclk,
per-XCD gfxclks, uclk, socket power, hotspot, throttle status) beside it, and report the average launch duration (hipEvents
around the whole loop / launches), the clock / power statistics of the samples taken while the loop ran (first 0.5 s dropped),
and the kernel's fractions of the HBM and matrix-pipe roofs at 2.4 GHz AND at the measured clock.
"""
import csv
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
VARIANT = os.environ.get("DGM_PROBE_LIB", "probe")        # dg-mesh_amd/lib/variants/<name>.so (ablation builds: probe_abl<bits>)
TAG = os.environ.get("DGM_PROBE_TAG", "r05_power")        # output prefix under gpurun_out/
lib = ctypes.CDLL(os.path.join(ROOT, "dg-mesh_amd", "lib", "variants", VARIANT + ".so"))
lib.dgm_p4_probe.restype = ctypes.c_int
lib.dgm_p4_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.dgm_last_error.restype = ctypes.c_char_p
sampler = os.path.join(ROOT, "tools", "bin", "smi_sampler")
torch.zeros(1, device="cuda")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

KINDS = {0: ("fwd_gemm", "mlp_gemm4_kernel<16,1024,512,0>", 2.0 * N * 1024 + 32 * N + 256 * 1024, 1),
         1: ("dw", "mlp_dw4_kernel<8,8>", 2.0 * N * 1024 + 256 * 1024, 1),
         2: ("pair", "mlp_bwd_pair_kernel", 3.0 * N * 1024 + 32 * N + 2 * 256 * 1024, 2),
         3: ("bwd_gemm", "mlp_gemm4_kernel<16,1024,512,1>", 2.0 * N * 1024 + 32 * N + 256 * 1024, 1)}
# round 6: the one-wave-per-SIMD forms (4 waves x 64 columns; mlp_planes5.hpp)
KINDS[4] = ("fwd_gemm5", "mlp_gemm5_kernel<0,0>", KINDS[0][2], 1)
KINDS[5] = ("bwd_gemm5", "mlp_gemm5_kernel<0,1>", KINDS[3][2], 1)
# (kinds 6 / 7 -- the paired launch and the weight gradient in that form -- were measured once, 3.3x / 4.3x slower: tools/exp/dw5_pair5_r6.hip.txt)
KINDS[9] = ("copy", "torch copy_ (elementwise kernel), 102.4 MB -> 102.4 MB", 2.0 * N * 1024, 0)
COPY_SRC = torch.randn(N * 256, device="cuda")
COPY_DST = torch.empty_like(COPY_SRC)
LAYER_FLOPS = 2.0 * N * 256 * 256


def run(kind, zero):
    def launch(iters):
        if kind == 9:  # a plain device copy of the layer GEMM's byte count (102 MB in + 102 MB out): what streaming HBM alone draws
            for _ in range(iters):
                COPY_DST.copy_(COPY_SRC)
            return
        rc = lib.dgm_p4_probe(N, kind, iters, zero, stream)
        if rc != 0:
            raise RuntimeError(lib.dgm_last_error().decode())
    launch(20)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch(200)
    e1.record()
    torch.cuda.synchronize()
    est = e0.elapsed_time(e1) / 200 * 1e-3
    iters = int(seconds / est) + 1
    name = KINDS[kind][0] + ("_zero" if zero else "_random")
    path = os.path.join(OUT, f"{TAG}_{name}.csv")
    proc = subprocess.Popen([sampler, path, "20", str(seconds * 3 + 20)]) if os.path.exists(sampler) else None
    time.sleep(0.3)
    t_start = time.time()
    e0.record()
    done = 0
    while done < iters:  # chunks: the launch queue never holds more than ~0.25 s of work, so the host clock tracks the device
        n = min(iters - done, max(int(0.25 / est), 1))
        launch(n)
        done += n
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    t_end = time.time()
    avg_us = e0.elapsed_time(e1) / iters * 1e3
    time.sleep(0.2)
    rows = []
    if proc is not None:
        proc.terminate()
        proc.wait()
        try:
            with open(path) as fh:
                rd = [r for r in csv.DictReader(l for l in fh if not l.startswith("#"))]
            t_first = None
            for r in rd:
                rows.append({k: float(v) for k, v in r.items()})
        except Exception as ex:  # noqa
            rows = []
    # the sampler started 0.3 s before the loop; keep samples inside [start + 0.5 s, end]
    dur = t_end - t_start
    keep = [r for r in rows if 0.3 + 0.5 <= r["t_s"] <= 0.3 + dur]
    out = {"kernel": KINDS[kind][1], "operands": "all-zero" if zero else "random binary16 planes", "N": N, "launches": iters,
           "loop_seconds": dur, "avg_launch_us": avg_us, "samples": len(keep), "sample_period_ms": 20}

    def stat(key, valid=lambda v: v > 0):
        v = np.array([r[key] for r in keep if valid(r[key])])
        return None if v.size == 0 else {"mean": float(v.mean()), "min": float(v.min()), "max": float(v.max()), "p10": float(np.percentile(v, 10)),
                                          "p90": float(np.percentile(v, 90))}
    out["gfxclk_mhz"] = stat("gfxclk_mhz", lambda v: 0 < v < 60000)
    xc = [stat(f"xcd{i}", lambda v: 0 < v < 60000) for i in range(8)]
    out["gfxclk_mhz_per_xcd_mean"] = [x["mean"] if x else None for x in xc]
    out["uclk_mhz"] = stat("uclk_mhz", lambda v: 0 < v < 60000)
    out["current_socket_power_w"] = stat("cur_socket_w", lambda v: 0 < v < 60000)
    out["average_socket_power_w"] = stat("avg_socket_w", lambda v: 0 < v < 60000)
    out["hotspot_c"] = stat("hotspot_c", lambda v: 0 < v < 60000)
    out["rsmi_sclk_mhz"] = stat("rsmi_sclk_mhz")
    out["rsmi_power_w"] = stat("rsmi_power_w")
    thr = sorted({int(r["throttle"]) for r in keep})
    out["throttle_status_values"] = thr[:8]
    by, nl = KINDS[kind][2], KINDS[kind][3]
    t = avg_us * 1e-6
    mfma_tf = 3.0 * nl * LAYER_FLOPS / t / 1e12
    clk = None
    xs = [x for x in out["gfxclk_mhz_per_xcd_mean"] if x]
    if xs:
        clk = float(np.mean(xs))
    elif out["gfxclk_mhz"]:
        clk = out["gfxclk_mhz"]["mean"]
    out["algorithmic_bytes"] = by
    pw = out["current_socket_power_w"] or out["average_socket_power_w"] or out["rsmi_power_w"]
    if pw:
        out["energy_mJ_per_launch"] = pw["mean"] * t * 1e3
    out["cycles_per_launch_at_measured_sclk"] = clk * 1e6 * t if clk else None
    out["hbm_GBps"] = by / t / 1e9
    out["frac_hbm_8TBps"] = by / t / 8e12
    out["mfma_issued_TFLOPs"] = mfma_tf
    out["frac_mfma_pipe_at_2400MHz"] = mfma_tf / 2500.0
    if clk:
        out["measured_sclk_mhz"] = clk
        out["frac_mfma_pipe_at_measured_sclk"] = mfma_tf / (2500.0 * clk / 2400.0)
    json.dump(out, open(os.path.join(OUT, f"{TAG}_{name}.json"), "w"), indent=1)
    print(VARIANT, name, f"{avg_us:.2f} us/launch", "E", round(out.get("energy_mJ_per_launch", 0.0), 1), "mJ", "kcycles", clk and round(clk * avg_us / 1e3, 1), "sclk", clk and round(clk), "power", (out["current_socket_power_w"] or out["average_socket_power_w"] or {}).get("mean"),
          "samples", len(keep), flush=True)
    return out


if __name__ == "__main__":
    res = {}
    idle = None
    kinds = [int(k) for k in os.environ.get("DGM_PROBE_KINDS", "0,1,2,3").split(",")]
    zeros = [int(z) for z in os.environ.get("DGM_PROBE_ZERO", "0,1").split(",")]
    for kind in kinds:
        for zero in zeros:
            res[KINDS[kind][0] + ("_zero" if zero else "_random")] = run(kind, zero)
            time.sleep(1.0)
    json.dump(res, open(os.path.join(OUT, f"{TAG}_summary.json"), "w"), indent=1)
