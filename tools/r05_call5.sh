#!/bin/bash
# round 5, call 5: gemm_ntq_kernel (320 x 256 tiles): parity under the all-shapes build, per-shape and in-situ A/B, timeline
mkdir -p gpurun_out/r05e
O=gpurun_out/r05e
L=ts-asr-whisper_amd/libdicow_hip.so
DICOW_HIP_LIB=$PWD/tools/libv_ntqall.so timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_nt or gemm_epilogues or gemm_identity" > $O/tests_ntq.txt 2>&1
tail -5 $O/tests_ntq.txt
ROUNDS=3 timeout 900 python tools/ab_nt2.py shipped=$L ntq=tools/libv_ntqall.so > $O/ab_ntq_shapes.txt 2>&1
cat $O/ab_ntq_shapes.txt
REPS=4 timeout 600 python tools/ab_encfwd.py shipped=$L ntq14=tools/libv_ntq14.so > $O/ab_encfwd.txt 2>&1
cat $O/ab_encfwd.txt
REPS=3 timeout 900 python tools/ab_step.py shipped=$L ntq14=tools/libv_ntq14.so > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
DICOW_HIP_LIB=$PWD/tools/libv_ntqp.so timeout 300 python tools/profile_ntr.py > $O/timeline_ntq.txt 2>&1
grep -A2 "N5120" $O/timeline_ntq.txt
