#!/bin/bash
# round 4, call 14: log-mel with the DFT folded about sample 200 -- goldens + timing
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04l; mkdir -p $O
timeout 600 python -m pytest tests/ -x -q -m gpu -k "logmel or from_audio or augment or features" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 200 python tools/bench_logmel.py tools/libv_lmdirect.so > $O/bench_logmel.txt 2>&1; cat $O/bench_logmel.txt
