#!/bin/bash
# Round 4, first call: the full GPU test run on the tree with the replica sync / CE sum / B=16 gradient pins, the default bench
# line (now with encoder_forward_train) and the in-situ comparison with hipBLASLt re-taken on this tree.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04a; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; tail -15 $O/gpu_tests.txt)
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
python tools/gemm_vs_hipblaslt.py $O/gemm_vs_hipblaslt.json > $O/gemm_vs_hipblaslt.txt 2>&1; tail -30 $O/gemm_vs_hipblaslt.txt
