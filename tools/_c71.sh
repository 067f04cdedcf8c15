cd $GRAFT_REPO_ROOT
T=$PWD/tools
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_realdims.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -2
REPS=4 timeout 900 python tools/ab_encfwd.py i0=$T/libv_i0.so i1=$T/libv_i1.so 2>&1 | grep -v amdgpu.ids | tail -3
