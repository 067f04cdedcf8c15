cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -2
for rep in 1 2 3; do
python tools/bench_attn.py 2>/dev/null | grep attn_fwd | cut -c1-80
ATTN_LOG2=1 python tools/bench_attn.py 2>/dev/null | grep attn_fwd | sed 's/attn_fwd/attn_fwd[log2]/' | cut -c1-80
done
