cd $GRAFT_REPO_ROOT
T=$PWD/tools
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_realdims.py -m gpu -q -x -k "ln or fddt or layer or rd_ or row or invariance or inference" 2>&1 | tail -2
for v in off r1w8 r2w8 r1w16 r1w4 r2w16; do echo "== $v"; DICOW_HIP_LIB=$T/libv_$v.so timeout 200 python tools/bench_rows.py 2>/dev/null | grep "fwd FDDT+LN"; done
REPS=3 timeout 900 python tools/ab_encfwd.py off=$T/libv_off.so r1w8=$T/libv_r1w8.so r2w8=$T/libv_r2w8.so r1w16=$T/libv_r1w16.so r1w4=$T/libv_r1w4.so r2w16=$T/libv_r2w16.so 2>&1 | grep -v amdgpu.ids | tail -7
