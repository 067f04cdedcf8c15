#!/bin/bash
# MFMA-busy and LDS-conflict counters of the bench step (PMC passes with kernel-trace only), summarised per kernel.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcm_$i
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcm_$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcm_$i.log 2>&1
  f=$(ls $R/gpurun_out/pmcm_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f $R/gpurun_out/pmcm_$i.csv; else echo "pass $i: no counter file"; tail -3 $R/gpurun_out/pmcm_$i.log; fi
  rm -rf $R/gpurun_out/pmcm_$i
done
python - <<PY
import csv, collections, json, glob
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob("$R/gpurun_out/pmcm_*.csv")):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = (r.get("Kernel_Name") or "").split("(")[0].replace("void ", "")
            c = r.get("Counter_Name"); v = float(r.get("Counter_Value") or 0)
            agg[name][c][0] += 1; agg[name][c][1] += v
out = {}
for k, cs in agg.items():
    m = {c: v[1] / max(1, v[0]) for c, v in cs.items()}
    m["launches"] = max(v[0] for v in cs.values())
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        m["mfma_busy_frac_of_simd_cycles"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024)
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    out[k] = m
keys = sorted(out, key=lambda k: -out[k].get("GRBM_GUI_ACTIVE", 0) * out[k]["launches"])
json.dump({k: out[k] for k in keys[:24]}, open("$R/gpurun_out/pmc_mfma_summary.json", "w"), indent=1)
for k in keys[:12]:
    print(k[:48], {a: (round(b, 4) if b < 10 else int(b)) for a, b in out[k].items()})
PY
rm -f $R/gpurun_out/pmcm_*.csv
