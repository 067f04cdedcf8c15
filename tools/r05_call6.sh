#!/bin/bash
mkdir -p gpurun_out/r05f
O=gpurun_out/r05f
L=ts-asr-whisper_amd/libdicow_hip.so
DICOW_HIP_LIB=$PWD/tools/libv_ntqall.so timeout 300 python tools/diag_ntq.py > $O/diag_asm.txt 2>&1; cat $O/diag_asm.txt
DICOW_HIP_LIB=$PWD/tools/libv_ntqnoasm.so timeout 300 python tools/diag_ntq.py > $O/diag_noasm.txt 2>&1; cat $O/diag_noasm.txt
REPS=3 timeout 600 python tools/ab_encfwd.py shipped=$L ntq14=tools/libv_ntq14.so ntqpe=tools/libv_ntqpe.so > $O/ab_encfwd.txt 2>&1
cat $O/ab_encfwd.txt
