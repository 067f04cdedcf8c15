#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04c; mkdir -p $O
DICOW_LN_FOLD=1 bash tools/prof_encfwd.sh > $O/prof_fold.txt 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $O/encfwd_fold.csv
DICOW_LN_FOLD=0 bash tools/prof_encfwd.sh > $O/prof_nofold.txt 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $O/encfwd_nofold.csv
python - <<'PY'
import csv
for tag in ("fold","nofold"):
    rows=list(csv.DictReader(open(f"gpurun_out/r04c/encfwd_{tag}.csv")))
    print(tag)
    for r in rows[:14]:
        print(f"  {r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us  {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
