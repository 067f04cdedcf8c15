#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04b; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_lnfold.py -q -x > $O/lnfold_tests.txt 2>&1; tail -25 $O/lnfold_tests.txt)
python tools/ab_lnfold.py > $O/ab_lnfold.txt 2>&1; tail -8 $O/ab_lnfold.txt
(timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -15 $O/gpu_tests.txt)
