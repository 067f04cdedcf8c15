#!/bin/bash
mkdir -p gpurun_out/r05h
O=gpurun_out/r05h
L=ts-asr-whisper_amd/libdicow_hip.so
REPS=4 timeout 900 python tools/ab_step.py shipped=$L ntq2=tools/libv_ntq2.so ntq6=tools/libv_ntq6.so > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
