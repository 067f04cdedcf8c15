#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04f; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_lnfold.py -q -x > $O/lnfold_tests.txt 2>&1; tail -4 $O/lnfold_tests.txt)
python tools/ab_lnfold.py > $O/ab_lnfold.txt 2>&1; tail -5 $O/ab_lnfold.txt
DICOW_LN_FOLD=1 bash tools/prof_encfwd.sh > $O/prof_fold.txt 2>&1; cp gpurun_out/encfwd_kernel_stats.csv $O/encfwd_fold.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r04f/encfwd_fold.csv")))
for r in rows[:8]:
    print(f"  {r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us  {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
