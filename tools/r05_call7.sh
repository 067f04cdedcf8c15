#!/bin/bash
mkdir -p gpurun_out/r05g
O=gpurun_out/r05g
DICOW_HIP_LIB=$PWD/tools/libv_ntqall.so timeout 300 python tools/diag_ntq.py > $O/diag_asm.txt 2>&1; cat $O/diag_asm.txt
DICOW_HIP_LIB=$PWD/tools/libv_ntqall.so timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -q -m gpu -k "gemm_nt or gemm_epilogues or gemm_identity" > $O/tests_ntq.txt 2>&1
tail -8 $O/tests_ntq.txt
