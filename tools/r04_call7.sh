#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04g; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -q -k "tn" > $O/tn_tests.txt 2>&1; tail -8 $O/tn_tests.txt)
REPS=3 python tools/ab_step.py w4=ts-asr-whisper_amd/libdicow_hip.so w8=tools/libv_tn8.so > $O/ab_step_tn.txt 2>&1; tail -6 $O/ab_step_tn.txt
