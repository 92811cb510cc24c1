"""Compact view of a rocprofv3 *_kernel_stats.csv: python tools/prof_summary.py <csv> [steps] [top]
The number of train steps the profile covers is taken from the trace itself -- one `render_bwd4_kernel` launch per step in
every phase (the mesh phase issues three `adam_kernel` launches per step, the Gaussian phase one: `adam_kernel` is only the
second choice) -- and the `steps` argument is the fallback for traces with neither (MLP-only runs)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
src = "argument"
for key in ("render_bwd4_kernel", "adam_kernel"):
    calls = [int(r["Calls"]) for r in rows if key in r["Name"]]
    if calls:
        steps, src = calls[0], key + " launches"
        break
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# total kernel time {tot/1e6:.2f} ms = {tot/1e6/steps:.3f} ms/step over {steps} steps ({src})")
print(f"# {'kernel':70s} {'calls':>6s} {'ms/step':>9s} {'avg_us':>9s} {'pct':>6s}")
for r in rows[:top]:
    n = r["Name"].replace("void ", "")
    print(f"{n[:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/steps:9.3f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
