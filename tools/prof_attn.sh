#!/bin/bash
# per-kernel times of tools/bench_attn.py under rocprofv3 (kernel trace): bash tools/prof_attn.sh [lib.so]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$1" ] && export DICOW_HIP_LIB=$R/$1
rm -rf $R/gpurun_out/pa
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pa -- python $R/tools/bench_attn.py > $R/gpurun_out/pa.log 2>&1
f=$(ls $R/gpurun_out/pa/*/*kernel_stats.csv | head -1)
grep attn_ $f | awk -F'","' '{gsub(/"/,"",$1); printf "%-60s calls %s avg %.1f us\n", substr($1,1,60), $2, $4/1000}'
rm -rf $R/gpurun_out/pa
