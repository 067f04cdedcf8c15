#!/bin/bash
# round 5, call 3: tile-walk (GM) sweep and DMA issue distribution of the ring kernel, in situ
mkdir -p gpurun_out/r05c
O=gpurun_out/r05c
L=ts-asr-whisper_amd/libdicow_hip.so
REPS=3 timeout 900 python tools/ab_encfwd.py shipped=$L gm4=tools/libv_gm4.so gm6=tools/libv_gm6.so gm16=tools/libv_gm16.so db8=tools/libv_db8.so db62=tools/libv_db62.so > $O/ab_encfwd.txt 2>&1
cat $O/ab_encfwd.txt
REPS=3 timeout 1200 python tools/ab_step.py shipped=$L gm4=tools/libv_gm4.so gm16=tools/libv_gm16.so db8=tools/libv_db8.so > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
