cd $GRAFT_REPO_ROOT
for v in d1 d2 d2r2 d2r4 d1 d2; do echo "== $v"; DICOW_HIP_LIB=tools/libvf_$v.so python tools/bench_rows.py 2>&1 | grep -E "bwd FDDT\+LN \(full\)|bwd FDDT\+LN no"; done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -k "fddt or row or ln" 2>&1 | tail -3
