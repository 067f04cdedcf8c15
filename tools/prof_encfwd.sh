#!/bin/bash
# rocprofv3 kernel-trace summary of the encoder forward alone -> gpurun_out/encfwd_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_enc
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_enc -- python $R/tools/enc_fwd.py 10 > $R/gpurun_out/prof_encfwd.log 2>&1
f=$(ls $R/gpurun_out/prof_enc/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/encfwd_kernel_stats.csv
rm -rf $R/gpurun_out/prof_enc
head -24 $R/gpurun_out/encfwd_kernel_stats.csv | cut -c1-160
tail -1 $R/gpurun_out/prof_encfwd.log | cut -c1-200
