cd $GRAFT_REPO_ROOT
export DICOW_HIP_LIB=$PWD/tools/libva_tmpl.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -2
unset DICOW_HIP_LIB
for rep in 1 2 3; do
for v in prev tmpl; do
DICOW_HIP_LIB=$PWD/tools/libva_$v.so ATTN_LOG2=1 timeout 120 python tools/bench_attn.py 2>/dev/null | grep "attn_bwd" | sed "s/attn_bwd/$v bwd/" | cut -c1-60
done
done
