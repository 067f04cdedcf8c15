#!/bin/bash
# HBM traffic counters (separate passes, kernel-trace only -- never combined with sys/hip/hsa traces) of the bench step.
# Writes gpurun_out/pmc_<counter>.csv (per-dispatch rows) and gpurun_out/pmc_summary.json (tools/pmc_summary.py).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1
  f=$(ls $R/gpurun_out/pmc_$c/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f $R/gpurun_out/pmc_$c.csv; fi
  rm -rf $R/gpurun_out/pmc_$c
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_FETCH_SIZE.csv $R/gpurun_out/pmc_WRITE_SIZE.csv > $R/gpurun_out/pmc_summary.json
rm -f $R/gpurun_out/pmc_FETCH_SIZE.csv $R/gpurun_out/pmc_WRITE_SIZE.csv      # per-dispatch tables are large; keep the summary
head -c 1500 $R/gpurun_out/pmc_summary.json
