#!/bin/bash
# round 5, call 4: full GPU test suite on the current tree + the snake MFMA order A/B
mkdir -p gpurun_out/r05d
O=gpurun_out/r05d
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1
tail -15 $O/gpu_tests.txt
L=ts-asr-whisper_amd/libdicow_hip.so
REPS=4 timeout 600 python tools/ab_encfwd.py shipped=$L snake=tools/libv_snake.so > $O/ab_snake_encfwd.txt 2>&1
cat $O/ab_snake_encfwd.txt
REPS=3 timeout 900 python tools/ab_step.py shipped=$L snake=tools/libv_snake.so > $O/ab_snake_step.txt 2>&1
cat $O/ab_snake_step.txt
