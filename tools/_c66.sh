cd $GRAFT_REPO_ROOT
T=$PWD/tools
DICOW_HIP_LIB=$T/libv_t1.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -2
for rep in 1 2; do for v in old t0 t1; do DICOW_HIP_LIB=$T/libv_$v.so ATTN_LOG2=1 timeout 120 python tools/bench_attn.py 2>/dev/null | grep "attn_fwd" | sed "s/attn_fwd/$v fwd/" | cut -c1-70; done; done
REPS=3 timeout 900 python tools/ab_encfwd.py old=$T/libv_old.so t0=$T/libv_t0.so t1=$T/libv_t1.so 2>&1 | grep -v amdgpu.ids | tail -4
