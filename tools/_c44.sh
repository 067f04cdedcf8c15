cd $GRAFT_REPO_ROOT
export DICOW_HIP_LIB=$PWD/tools/libva_dkvjit.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -2
unset DICOW_HIP_LIB
REPS=4 timeout 900 python tools/ab_attn.py base=tools/libva_base.so dkvjit=tools/libva_dkvjit.so 2>&1 | tail -20
