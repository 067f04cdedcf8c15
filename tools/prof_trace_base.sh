#!/bin/bash
# kernel trace (start / end stamps) of the whisper-base B=8 graph-replayed step -> gpurun_out/base_trace_summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/proft
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/proft -- python $R/bench.py --model whisper-base --batch 8 --graph --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/proft_bench.log 2>&1
f=$(ls $R/gpurun_out/proft/*/*kernel_trace.csv | head -1)
python - "$f" > $R/gpurun_out/base_trace_summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# steps are delimited by adamw_hyper_kernel (one per optimizer step)
idx = [i for i, e in enumerate(ev) if e[2].startswith("adamw_hyper_kernel")]
print("kernels", len(ev), "optimizer steps", len(idx))
for a, b in list(zip(idx, idx[1:]))[-4:]:
    seg = ev[a:b]
    wall = seg[-1][1] - seg[0][0] if False else ev[b][0] - ev[a][0]
    busy = sum(e[1] - e[0] for e in seg)
    gaps = sorted((seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)), reverse=True)
    print(f"step: {len(seg)} kernels, wall {wall/1e6:.3f} ms, kernel time {busy/1e6:.3f} ms, idle {100*(1-busy/wall):.1f} %, median gap {gaps[len(gaps)//2]/1e3:.2f} us, 10 largest gaps {[round(g/1e3,1) for g in gaps[:10]]}")
a, b = idx[-2], idx[-1]
c = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev[a:b]:
    c[n.split("(")[0][:70]][0] += 1; c[n.split("(")[0][:70]][1] += e - s
for n, (k, t) in sorted(c.items(), key=lambda x: -x[1][1])[:32]:
    print(f"{n:72s} {k:5d} {t/1e3:9.1f} us")
PY
cat $R/gpurun_out/base_trace_summary.txt
rm -rf $R/gpurun_out/proft
