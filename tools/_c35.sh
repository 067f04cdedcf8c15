cd $GRAFT_REPO_ROOT
export DICOW_HIP_LIB=$PWD/tools/libva_nw8.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -4
unset DICOW_HIP_LIB
REPS=3 timeout 900 python tools/ab_attn.py nw4=tools/libva_nw4.so nw8=tools/libva_nw8.so 2>&1 | tail -20
