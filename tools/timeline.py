"""Per-step timeline of a rocprofv3 kernel trace (scripts/gpu_trace.sh writes gpurun_out/trace/tail.csv: start, end, queue, name).
Prints, for the last full steps, where each queue (HIP stream) is busy and how the chip is shared between them."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace/tail.csv")))
ev = [(int(r["start"]), int(r["end"]), r["queue"], r["name"]) for r in rows]
ev.sort()
# a step starts at mlp_timenet_fwd / embed4 of the deformation network following an adam_kernel
starts = [i for i, e in enumerate(ev) if "adam_kernel" in e[3]]
for a, b in list(zip(starts[:-1], starts[1:]))[-3:]:
    step = ev[a + 1:b + 1]
    t0 = step[0][0]
    print(f"--- step of {(step[-1][1] - t0) / 1e3:.0f} us, {len(step)} kernels")
    queues = sorted({e[2] for e in step})
    for q in queues:
        qs = [e for e in step if e[2] == q]
        busy = sum(e[1] - e[0] for e in qs)
        print(f"  queue {q}: {len(qs)} kernels, busy {busy / 1e3:.0f} us, from {(qs[0][0] - t0) / 1e3:.0f} to {(qs[-1][1] - t0) / 1e3:.0f} us")
    if "-v" in sys.argv:
        for e in step:
            print(f"    {(e[0] - t0) / 1e3:8.1f} {(e[1] - e[0]) / 1e3:7.1f}  q{e[2]}  {e[3][:50]}")
