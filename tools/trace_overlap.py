"""rocprofv3 kernel trace (csv) of a run with side-stream kernels: while a side-stream kernel (name contains `fabric_emulate` or
`oneRankReduce` or `ccl`) is running, how much of that time is some OTHER queue's kernel running too?   python tools/trace_overlap.py trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ks.sort()
side = [k for k in ks if any(t in k[2] for t in ("fabric_emulate", "oneRankReduce", "ccl"))]
main = [k for k in ks if k not in side]
qs = collections.Counter(k[3] for k in side), collections.Counter(k[3] for k in main)
print("queues of the side kernels:", dict(qs[0]), " of the others:", dict(qs[1].most_common(3)))
import bisect
starts = [k[0] for k in main]
tot_side = cov = 0
for s0, s1, n, q in side:
    if "fabric" not in n:
        continue
    tot_side += s1 - s0
    i = max(0, bisect.bisect_left(starts, s0) - 3)
    segs = []
    while i < len(main) and main[i][0] < s1:
        a, b = max(main[i][0], s0), min(main[i][1], s1)
        if b > a:
            segs.append((a, b))
        i += 1
    segs.sort()
    cur_a = cur_b = None
    for a, b in segs:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                cov += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        cov += cur_b - cur_a
print(f"emulator kernels: {tot_side / 1e6:.1f} ms in total; another queue's kernel was running during {cov / 1e6:.1f} ms of it ({100.0 * cov / max(1, tot_side):.0f} %)")
# the gaps of the main queue: time between the end of a main kernel and the start of the next, summed, inside / outside emulator intervals
gap_in = gap_out = 0
for (a0, a1, _, _), (b0, b1, _, _) in zip(main, main[1:]):
    g = b0 - a1
    if g <= 0 or g > 5e6:
        continue
    mid = (a1 + b0) // 2
    inside = any(s0 <= mid <= s1 for s0, s1, n, q in side if "fabric" in n and s0 - 2e6 < mid < s1 + 2e6)
    if inside: gap_in += g
    else: gap_out += g
print(f"idle gaps between consecutive kernels of the other queues: {gap_in / 1e6:.1f} ms while an emulator kernel runs, {gap_out / 1e6:.1f} ms otherwise")
