cd $GRAFT_REPO_ROOT
export DICOW_HIP_LIB=$PWD/tools/libva_dkv8.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -3
unset DICOW_HIP_LIB
REPS=4 timeout 900 python tools/ab_attn.py base=tools/libva_base.so dkv8=tools/libva_dkv8.so 2>&1 | tail -20
