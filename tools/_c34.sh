cd $GRAFT_REPO_ROOT
export DICOW_HIP_LIB=$PWD/tools/libva_pipe.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attn" 2>&1 | tail -4
unset DICOW_HIP_LIB
REPS=3 timeout 900 python tools/ab_attn.py old=tools/libva_old.so pipe=tools/libva_pipe.so nosched=tools/libva_nosched.so pipe3=tools/libva_pipe3.so latev0=tools/libva_latev0.so 2>&1 | tail -20
