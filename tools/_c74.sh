cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fddt_epilogue or gemm" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_realdims.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/enc_fwd.py 10 2>/dev/null | tail -1 | cut -c1-110; done
