"""Aggregate the rocprofv3 --pmc passes written by scripts/gpu_pmc_sq.sh per kernel:
    python tools/pmc_sq_summary.py <tag>   ->  gpurun_out/pmc_sq_<tag>.json  (+ a compact table on stdout)
Per kernel: dispatches and the per-dispatch average of every counter, plus derived ratios
(units per MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves,
SQ_BUSY_CYCLES and SQ_VALU_MFMA_BUSY_CYCLES are cycles summed over SQs / SIMDs)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.join("gpurun_out")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(root, f"pmc_sq_{tag}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "")
        if "dgm::" not in k:
            continue
        k = k.split("(")[0][:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {}
for k, c in agg.items():
    e = {name: v / max(len(disp[k][name]), 1) for name, v in c.items()}
    e["dispatches"] = max(len(s) for s in disp[k].values())
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS",
                  "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA"):
            if n in e:
                e["frac_of_wave_cycles/" + n] = e[n] / wc
    if e.get("SQ_BUSY_CYCLES") and e.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        e["mfma_busy_over_sq_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"]
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs (render_bwd2: 1.246e7 for a 0.67 ms kernel = 8 x 1.56e6 cycles), so
    # the kernel's wall-clock cycles are GRBM_GUI_ACTIVE / 8; SQ cycle counters are summed over 1024 SIMDs (256 CU x 4).
    wall = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if wall:
        e["wall_cycles"] = wall
    if wall and e.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        e["mfma_util_of_chip"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (wall * 1024.0)
    if wall and e.get("SQ_ACTIVE_INST_VALU"):
        # quad-cycles of VALU execution summed over waves -> x4 cycles
        e["valu_busy_of_chip"] = 4.0 * e["SQ_ACTIVE_INST_VALU"] / (wall * 1024.0)
    out[k] = e
json.dump(out, open(os.path.join(root, f"pmc_sq_{tag}.json"), "w"), indent=1, sort_keys=True)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", kv[1].get("SQ_BUSY_CYCLES", 0)) * kv[1]["dispatches"]):
    print(f"{k[:56]:56s} n={e['dispatches']:4d} " + " ".join(
        f"{n.split('/')[-1][3:] if n.startswith('frac') else n}={e[n]:.3g}" for n in sorted(e)
        if n.startswith("frac_of") or n in ("mfma_util_of_chip", "valu_busy_of_chip", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU")))
