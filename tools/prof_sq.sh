#!/bin/bash
# Where a kernel's wave cycles go: SQ counters (PMC passes with kernel-trace only) over any command,
# summarised per kernel -> gpurun_out/sq_pmc_summary.json.   bash tools/prof_attn_pmc.sh [bench_attn args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
PFX=$1; shift; CMD="$@"     # usage: bash tools/prof_sq.sh <kernel-name-prefix> <command...>
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/sqpmc_$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sqpmc_$i -- $CMD > $R/gpurun_out/sqpmc_$i.log 2>&1
  f=$(ls $R/gpurun_out/sqpmc_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f $R/gpurun_out/sqpmc_$i.csv; else echo "pass $i: no counter file"; tail -5 $R/gpurun_out/sqpmc_$i.log; fi
  rm -rf $R/gpurun_out/sqpmc_$i
done
python - <<PY
import csv, collections, json, glob
PFX = "$PFX"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob("$R/gpurun_out/sqpmc_*.csv")):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = (r.get("Kernel_Name") or "").split("(")[0].replace("void ", "")
            if not name.startswith(PFX): continue
            c = r.get("Counter_Name"); v = float(r.get("Counter_Value") or 0)
            agg[name][c][0] += 1; agg[name][c][1] += v
out = {}
for k, cs in agg.items():
    m = {c: v[1] / max(1, v[0]) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m: m[c + "/WAVE_CYCLES"] = m[c] / wc
    out[k] = m
json.dump(out, open("$R/gpurun_out/sq_pmc_summary.json", "w"), indent=1)
for k, m in out.items():
    print(k)
    for a, b in sorted(m.items()): print("   %-34s %s" % (a, round(b, 4) if b < 10 else int(b)))
PY
rm -f $R/gpurun_out/sqpmc_*.csv
