"""Compact view of a rocprofv3 *_kernel_stats.csv: python tools/prof_summary.py <csv> [steps] [top] [setup_launches]
setup_launches (bench.py with teacher targets: 200 forward renders at set-up): kernels with exactly steps + setup_launches calls are
priced per step on their average launch time, and the set-up's share is taken out of the total.
The number of train steps the profile covers is taken from the trace itself -- one `render_bwd4_kernel` launch per step in
every phase (the mesh phase issues three `adam_kernel` launches per step, the Gaussian phase one: `adam_kernel` is only the
second choice) -- and the `steps` argument is the fallback for traces with neither (MLP-only runs)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
setup = int(sys.argv[4]) if len(sys.argv) > 4 else 0
src = "argument"
for key in ("render_bwd4_kernel", "adam_kernel"):
    calls = [int(r["Calls"]) for r in rows if key in r["Name"]]
    if calls:
        steps, src = calls[0], key + " launches"
        break
tot = sum(float(r["TotalDurationNs"]) for r in rows)
setup_ns = 0.0
if setup:
    for r in rows:
        if int(r["Calls"]) == steps + setup:
            share = float(r["AverageNs"]) * setup
            r["TotalDurationNs"] = str(float(r["TotalDurationNs"]) - share)
            setup_ns += share
    tot -= setup_ns
print(f"# total kernel time {tot/1e6:.2f} ms = {tot/1e6/steps:.3f} ms/step over {steps} steps ({src})"
      + (f"; {setup_ns/1e6:.2f} ms of {setup} set-up launches per forward kernel (teacher targets) taken out" if setup else ""))
print(f"# {'kernel':70s} {'calls':>6s} {'ms/step':>9s} {'avg_us':>9s} {'pct':>6s}")
for r in rows[:top]:
    n = r["Name"].replace("void ", "")
    print(f"{n[:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/steps:9.3f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
