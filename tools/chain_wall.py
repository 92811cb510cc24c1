"""Wall time of the forward binning chain from a rocprofv3 --kernel-trace CSV: kernels that run side by side (the tile-sort classes,
launched without a barrier between them) do not add up, so per-kernel averages say nothing about the chain.  Per render_fwd launch:
    chain = render_fwd.start - preprocess_fwd.start      sort = render_fwd.start - scatter.end
usage: python tools/chain_wall.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    names = ["preprocess_fwd_kernel", "scan_exclusive_kernel", "count_tiles_kernel", "tile_scan_kernel", "scatter_kernel",
             "tile_sort_radix_mid_kernel", "tile_sort_radix_kernel", "tile_sort_radix_big_kernel"]
    short = lambda n: next((k for k in names + ["render_fwd_kernel"] if "dgm::" + k + "(" in n or "dgm::" + k + "<" in n), None)
    chain, sort, dur = [], [], {k: [] for k in names}
    cur = {}
    for s, e, n in rows:
        k = short(n)
        if k is None:
            continue
        if k == "render_fwd_kernel":
            if "preprocess_fwd_kernel" in cur and "scatter_kernel" in cur:
                chain.append(s - cur["preprocess_fwd_kernel"][0])
                sort.append(s - cur["scatter_kernel"][1])
                for kk, (ss, ee) in cur.items():
                    dur[kk].append(ee - ss)
            cur = {}
        else:
            cur[k] = (s, e)
    skip = len(chain) // 5  # warm-up launches
    mean = lambda v: sum(v[skip:]) / max(1, len(v[skip:])) / 1e3
    out = {"launches": len(chain) - skip, "chain_us": mean(chain), "sort_wall_us": mean(sort),
           "kernel_us": {k: mean(v) for k, v in dur.items() if v}}
    out["kernel_sum_us"] = sum(out["kernel_us"].values())
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
