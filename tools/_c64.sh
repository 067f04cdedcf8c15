cd $GRAFT_REPO_ROOT
T=$PWD/tools
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_realdims.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -2
for v in off w8 w4 w16; do echo "== $v"; DICOW_HIP_LIB=$T/libv_$v.so timeout 200 python tools/bench_rows.py 2>/dev/null | grep "bwd LN only"; done
REPS=3 timeout 900 python tools/ab_step.py off=$T/libv_off.so w8=$T/libv_w8.so w4=$T/libv_w4.so w16=$T/libv_w16.so 2>&1 | grep -v amdgpu.ids | tail -5
