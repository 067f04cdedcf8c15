#!/bin/bash
# rocprofv3 kernel-trace summary of the decoding path (tools/bench_decode_ctc.py) -> gpurun_out/decode_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/tools/bench_decode_ctc.py > $R/gpurun_out/prof_decode.log 2>&1
f=$(ls $R/gpurun_out/prof/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/decode_kernel_stats.csv
rm -rf $R/gpurun_out/prof
head -16 $R/gpurun_out/decode_kernel_stats.csv | cut -c1-160
tail -3 $R/gpurun_out/prof_decode.log | cut -c1-300
