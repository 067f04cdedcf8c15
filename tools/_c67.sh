cd $GRAFT_REPO_ROOT
T=$PWD/tools
DICOW_HIP_LIB=$T/libv_b1.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_realdims.py -m gpu -q -x -k "attn or attention or rd_" 2>&1 | tail -2
for rep in 1 2; do for v in b0 b1; do DICOW_HIP_LIB=$T/libv_$v.so ATTN_LOG2=1 timeout 120 python tools/bench_attn.py 2>/dev/null | grep "attn_bwd" | sed "s/attn_bwd/$v bwd/" | cut -c1-90; done; done
REPS=3 timeout 900 python tools/ab_step.py b0=$T/libv_b0.so b1=$T/libv_b1.so 2>&1 | grep -v amdgpu.ids | tail -3
