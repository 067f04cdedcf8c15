cd $GRAFT_REPO_ROOT
T=$PWD/tools
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "ln or fddt or layer" 2>&1 | tail -2
for v in off r2w4 r1w4 r4w4 r2w8 r1w8; do echo "== $v"; DICOW_HIP_LIB=$T/libv_$v.so timeout 200 python tools/bench_rows.py 2>/dev/null | grep "fwd LN only"; done
REPS=3 timeout 900 python tools/ab_encfwd.py off=$T/libv_off.so r2w4=$T/libv_r2w4.so r1w4=$T/libv_r1w4.so r4w4=$T/libv_r4w4.so r2w8=$T/libv_r2w8.so r1w8=$T/libv_r1w8.so 2>&1 | grep -v amdgpu.ids | tail -7
