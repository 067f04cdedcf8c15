#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04h; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_optimizer.py tests/test_gpu_graph.py tests/test_gpu_trajectory.py tests/test_gpu_dp.py tests/test_gpu_rccl.py tests/test_gpu_model.py -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt)
for i in 1 2; do python bench.py --no-cpu-baseline --steps 12 --warmup 4 > $O/bench_$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_median'])"; done
bash tools/prof_step.sh > /dev/null 2>&1; python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/kernel_stats.csv")))
for r in rows:
    n=r['Name']
    if 'at::' in n or 'rocclr' in n or 'multi_tensor' in n or 'foreach' in n.lower():
        print(f"{n[:110]:110s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
