#!/bin/bash
# L2 hit rate per kernel of the bench step: one --pmc pass (kernel-trace only) with TCC_HIT_sum / TCC_MISS_sum
# -> gpurun_out/pmc_l2_summary.json (hit rate = HIT / (HIT + MISS), MI355X_MICROARCH.md "L2")
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_l2
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_l2.log 2>&1
f=$(ls $R/gpurun_out/pmc_l2/*/*counter_collection.csv 2>/dev/null | head -1)
python - "$f" > $R/gpurun_out/pmc_l2_summary.json <<'PY'
import csv, sys, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = (r.get("Kernel_Name") or "").split("(")[0].replace("void ", "")
        agg[name][r.get("Counter_Name")][0] += 1; agg[name][r.get("Counter_Name")][1] += float(r.get("Counter_Value") or 0)
out = {}
for k, cs in agg.items():
    h, m = cs["TCC_HIT_sum"], cs["TCC_MISS_sum"]
    if h[0] == 0: continue
    out[k] = {"launches": h[0], "TCC_HIT_per_launch": h[1] / h[0], "TCC_MISS_per_launch": m[1] / max(1, m[0]),
              "l2_hit_rate": h[1] / max(1.0, h[1] + m[1])}
keys = sorted(out, key=lambda k: -(out[k]["TCC_HIT_per_launch"] + out[k]["TCC_MISS_per_launch"]) * out[k]["launches"])
print(json.dumps({k: out[k] for k in keys[:24]}, indent=1))
PY
rm -rf $R/gpurun_out/pmc_l2
python -c "
import json; d=json.load(open('$R/gpurun_out/pmc_l2_summary.json'))
for k,v in list(d.items())[:14]: print(k[:46].ljust(46), 'L2 hit rate %.3f' % v['l2_hit_rate'], 'requests/launch %.2e' % (v['TCC_HIT_per_launch']+v['TCC_MISS_per_launch']))"
