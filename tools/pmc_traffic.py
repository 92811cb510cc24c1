"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, --kernel-trace only):
    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out dir> <workload>
Writes <out dir>/pmc_<name>.json for the kernels bench.py prices.  Units and the gfx950 correction follow
MI355X_MICROARCH.md: the counters are KB; FETCH_SIZE reports half of the bytes of wide coalesced streaming reads on
gfx950 and is doubled (calibration: the layer GEMM reads 100000 x 256 fp32 = 102.4 MB + 3.2 MB of masks + weights)."""
import collections
import csv
import json
import os
import sys

fetch_csv, write_csv, out_dir, workload = sys.argv[1:5]
N = int(sys.argv[5]) if len(sys.argv) > 5 else None            # rows of the MLP launches / Gaussians of the scene (bench.py checks it)
steps = float(sys.argv[6]) if len(sys.argv) > 6 else None      # train steps the profiled command ran (for launches per step)
commit = sys.argv[7] if len(sys.argv) > 7 else os.environ.get("DGM_COMMIT")
KERNELS = {"render_bwd4": "render_bwd4_kernel", "render_fwd": "render_fwd_kernel", "preprocess_bwd": "preprocess_bwd_kernel",
           "tile_sort_radix": "tile_sort_radix_kernel", "reduce_dw": "mlp_reduce_dw_all_kernel", "scatter": "dgm::scatter_kernel",
           "count_tiles": "count_tiles_kernel", "tile_scan": "tile_scan_kernel",
           # the plane arithmetic
           "gemm4_fwd": "mlp_gemm4_kernel<16, 1024, 512, 0, false, 8", "gemm4_bwd": "mlp_gemm4_kernel<16, 1024, 512, 1, false, 8",
           "dw4": "mlp_dw4_kernel<8, 8, 1024, 512, 1024, 512>", "gemm4_skip": "mlp_gemm4_kernel<16, 1024, 512, 2, false, 8>",
           "gemm4_l0": "mlp_gemm4_kernel<6, 384, 192, 0, true, 8>", "dw4_emb": "mlp_dw4_kernel<3, 8, 384, 192, 1024, 512>",
           "embed4": "mlp_embed4_kernel", "bwd_pair": "mlp_bwd_pair_kernel",
           # round 6: one time row per call folded into the biases
           "gemm5_skip": "mlp_gemm5_kernel<4, 0>", "gemm4_l0f": "mlp_gemm4_kernel<4, 256, 128, 0, false, 8",
           "dw4_emb64": "mlp_dw4_kernel<2, 8, 256, 128, 1024, 512>"}
# kernels whose reads are gathers of short records: the x2 streaming-read correction of FETCH_SIZE is not calibrated for them
GATHER = {"render_bwd4", "render_fwd", "preprocess_bwd", "tile_sort_radix", "scatter", "count_tiles", "tile_scan"}


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: tot[k] / len(n[k]) for k in tot}, {k: len(n[k]) for k in tot}


f, fn = per_kernel(fetch_csv, "FETCH_SIZE")
w, wn = per_kernel(write_csv, "WRITE_SIZE")
os.makedirs(out_dir, exist_ok=True)
for short, pat in KERNELS.items():
    fk = [k for k in f if pat in k]
    wk = [k for k in w if pat in k]
    if not fk or not wk:
        continue  # (not in this run, e.g. the unpaired layer kernels)
    fetch_kb, write_kb = f[fk[0]], w[wk[0]]
    gather = short in GATHER
    rec = {"workload": workload, "kernel": pat, "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
           "fetch_bytes": (1.0 if gather else 2.0) * fetch_kb * 1024.0, "write_bytes": write_kb * 1024.0,
           "launches_sampled": [fn[fk[0]], wn[wk[0]]], "N": N, "commit": commit,
           "launches_per_step": (fn[fk[0]] / steps) if steps else None,
           "method": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, over bench.py; "
                     "counters are KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950",
           "fetch_bytes_raw": fetch_kb * 1024.0, "fetch_bytes_doubled": 2.0 * fetch_kb * 1024.0,
           "fetch_correction": ("none (gathers of short records / indexed rows: the x2 streaming-read correction is uncalibrated "
                                "for this access pattern, so the raw counter is a LOWER bound and x2 an upper bound)") if gather
           else "x2 (wide coalesced streaming reads)"}
    json.dump(rec, open(os.path.join(out_dir, f"pmc_{short}.json"), "w"), indent=1)
    print(f"{short:16s} fetch {rec['fetch_bytes'] / 1e6:8.1f} MB  write {rec['write_bytes'] / 1e6:8.1f} MB per launch")
