"""rocprofv3 kernel trace (csv) of the bench command: is the step GPU-bound?  Finds the optimizer steps (one `sumsq_kernel` launch each), and for the
steps of the timed region reports wall time first-kernel-start -> last-kernel-end, the sum of kernel durations, the idle time between consecutive
kernels and the largest gaps.   python tools/trace_gaps.py trace.csv"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows)
marks = [i for i, k in enumerate(ks) if "sumsq_kernel" in k[2]]
print(f"{len(ks)} kernels, {len(marks)} optimizer steps in the trace")
steps = []
for a, b in zip(marks, marks[1:]):                      # a step = from the kernel after one gradient-norm launch to the next one's AdamW launches (approximately: norm to norm)
    seg = ks[a:b]
    wall = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [(seg[i + 1][0] - seg[i][1], seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1)]
    idle = sum(max(0, g[0]) for g in gaps)
    steps.append((wall, busy, idle, len(seg), sorted(gaps, reverse=True)[:3]))
core = [s for s in steps if 0.8 * statistics.median(x[0] for x in steps) < s[0] < 1.2 * statistics.median(x[0] for x in steps)]
print(f"{len(core)} regular steps: wall {statistics.median(s[0] for s in core) / 1e6:.3f} ms (median), kernels {statistics.median(s[1] for s in core) / 1e6:.3f} ms, "
      f"idle between kernels {statistics.median(s[2] for s in core) / 1e6:.3f} ms = {100 * statistics.median(s[2] / s[0] for s in core):.2f} %, "
      f"{statistics.median(s[3] for s in core):.0f} launches per step, mean gap {1e-3 * statistics.median(s[2] / s[3] for s in core):.2f} us")
w = core[len(core) // 2]
print("largest gaps of one step:", "; ".join(f"{g / 1e3:.1f} us after {a} before {b}" for g, a, b in w[4]))
