#!/bin/bash
# round 5, call 1: probes (hand-off, two-workgroup DMA), gemm_nt2_kernel correctness under the variant build, per-shape and in-situ A/B
mkdir -p gpurun_out/r05a
O=gpurun_out/r05a
export TMPDIR=/tmp
timeout 120 tools/probe_dma2 > $O/probe_dma2.txt 2>&1
timeout 300 tools/probe_handoff > $O/probe_handoff.txt 2>&1
DICOW_HIP_LIB=$PWD/tools/libv_nt2.so timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -x -q -m gpu -k "gemm_nt or gemm_epilogues or gemm_identity" > $O/tests_nt2.txt 2>&1
tail -5 $O/tests_nt2.txt
ROUNDS=3 timeout 900 python tools/ab_nt2.py shipped=ts-asr-whisper_amd/libdicow_hip.so nt2=tools/libv_nt2.so nt2d=tools/libv_nt2d.so nt2u=tools/libv_nt2u.so > $O/ab_nt2.txt 2>&1
cat $O/ab_nt2.txt
timeout 900 python tools/ab_encfwd.py shipped=ts-asr-whisper_amd/libdicow_hip.so nt2=tools/libv_nt2.so nt2d=tools/libv_nt2d.so > $O/ab_encfwd.txt 2>&1
cat $O/ab_encfwd.txt
head -20 $O/probe_dma2.txt
cat $O/probe_handoff.txt
