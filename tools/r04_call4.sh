#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04d; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "logmel" > $O/logmel_tests.txt 2>&1; tail -5 $O/logmel_tests.txt)
python tools/bench_logmel.py tools/libv_lmdirect.so > $O/bench_logmel.txt 2>&1; tail -4 $O/bench_logmel.txt
DICOW_HIP_LIB=tools/libv_prof.so python tools/profile_ntr.py > $O/ntr_tile_timeline.txt 2>&1; tail -30 $O/ntr_tile_timeline.txt
ATTN_LOG2=1 REPS=3 python tools/ab_attn.py shipped=ts-asr-whisper_amd/libdicow_hip.so cts=tools/libv_acts.so cts1b=tools/libv_acts1b.so > $O/ab_attn.txt 2>&1; cat $O/ab_attn.txt
(DICOW_HIP_LIB=tools/libv_acts1b.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -k "attn or attention" > $O/attn_tests_cts1b.txt 2>&1; tail -4 $O/attn_tests_cts1b.txt)
