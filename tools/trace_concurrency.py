"""rocprofv3 kernel trace (csv) of the bench command with the two half-batch streams: how many kernels are in flight, for how much of a step?
Steps are delimited by the gradient-norm launch (`sumsq_kernel`, one per optimizer step).  Prints, for the regular steps, the share of wall time with
0 / 1 / 2 / >= 3 kernels in flight, the queues the kernels ran on, and per kernel family the time it ran alone and beside another kernel.
python tools/trace_concurrency.py trace.csv"""
import csv, sys, statistics, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Queue_Id", "?")) for r in rows)
marks = [i for i, k in enumerate(ks) if "sumsq_kernel" in k[2]]
steps = []
for a, b in zip(marks, marks[1:]):
    seg = ks[a:b]
    t0, t1 = seg[0][0], max(k[1] for k in seg)
    ev = []
    for j, (s, e, n, q) in enumerate(seg):
        ev.append((s, 1, j)); ev.append((e, -1, j))
    ev.sort()
    live, last, hist = set(), t0, collections.Counter()
    alone, paired = collections.Counter(), collections.Counter()
    for t, d, j in ev:
        dt = t - last
        if dt > 0:
            hist[min(len(live), 3)] += dt
            for i in live:
                (alone if len(live) == 1 else paired)[seg[i][2]] += dt
        last = t
        live.add(j) if d > 0 else live.discard(j)
    gaps, live, last_end = [], 0, None
    cur_end = None
    for idx_, (s_, e_, n_, q_) in enumerate(seg):                # idle intervals (no kernel in flight): between the running maximum of the ends and the next start
        if cur_end is not None and s_ > cur_end:
            gaps.append((s_ - cur_end, prev_n, n_, (cur_end - t0) / 1e6, idx_))
        if cur_end is None or e_ > cur_end:
            cur_end, prev_n = e_, n_
    steps.append((t1 - t0, hist, alone, paired, collections.Counter(k[3] for k in seg), len(seg), sorted(gaps, reverse=True)[:10], sum(g[0] for g in gaps), len(gaps), seg))
med = statistics.median(s[0] for s in steps)
core = [s for s in steps if 0.8 * med < s[0] < 1.2 * med]
print(f"{len(ks)} kernels, {len(marks)} optimizer steps, {len(core)} regular steps of {med / 1e6:.2f} ms (median wall), {statistics.median(s[5] for s in core):.0f} launches per step")
for n in range(4):
    print(f"  {'>= 3' if n == 3 else n} kernel(s) in flight: {100 * statistics.median(s[1][n] / s[0] for s in core):5.1f} % of the step")
print("  queues (launches per step):", dict(core[len(core) // 2][4]))
fam = collections.Counter()
for s in core:
    for k, v in s[2].items(): fam[k] += v
    for k, v in s[3].items(): fam[k] += v
print(f"  {'kernel':50s} {'ms/step':>8s} {'alone':>8s} {'beside':>8s}")
for k, _ in fam.most_common(16):
    a = sum(s[2][k] for s in core) / len(core) / 1e6
    p = sum(s[3][k] for s in core) / len(core) / 1e6
    print(f"  {k:50s} {a + p:8.2f} {a:8.2f} {p:8.2f}")
w = core[len(core) // 2]
print(f"  idle intervals of one step: {w[8]} totalling {w[7] / 1e3:.0f} us; the largest:")
for g, a, b, off, idx in w[6]:
    ctx0 = " | ".join(f"{k[2][:28]}@q{k[3]}" for k in w[9][max(0, idx - 4):idx])
    ctx1 = " | ".join(f"{k[2][:28]}@q{k[3]}" for k in w[9][idx:idx + 4])
    print(f"    {g / 1e3:8.1f} us at {off:7.2f} ms into the step\n        before: {ctx0}\n        after:  {ctx1}")
