#!/bin/bash
# rocprofv3 kernel-trace summary of the bench step -> gpurun_out/prof/ (copy the *_kernel_stats.csv into profiles/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/prof_bench.log 2>&1
f=$(ls $R/gpurun_out/prof/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/kernel_stats.csv
rm -rf $R/gpurun_out/prof
head -30 $R/gpurun_out/kernel_stats.csv | cut -c1-150
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-200
