cd $GRAFT_REPO_ROOT
export DICOW_HIP_LIB=$PWD/tools/libva_pksum.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attn" 2>&1 | tail -2
for rep in 1 2 3; do
for v in base pksum; do
DICOW_HIP_LIB=$PWD/tools/libva_$v.so ATTN_LOG2=1 python tools/bench_attn.py 2>/dev/null | grep "attn_fwd" | sed "s/attn_fwd/$v/" | cut -c1-70
done
done
