cd $GRAFT_REPO_ROOT
export DICOW_HIP_LIB=$PWD/tools/libva_spec.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -3
unset DICOW_HIP_LIB
for rep in 1 2 3; do
for v in base spec; do
DICOW_HIP_LIB=$PWD/tools/libva_$v.so ATTN_LOG2=1 timeout 120 python tools/bench_attn.py 2>/dev/null | grep "attn_fwd" | sed "s/attn_fwd/$v/" | cut -c1-70
done
done
for v in base spec; do DICOW_HIP_LIB=$PWD/tools/libva_$v.so timeout 300 python tools/enc_fwd.py 2>/dev/null | tail -1 | cut -c1-100; done
for v in base spec; do DICOW_HIP_LIB=$PWD/tools/libva_$v.so timeout 300 python tools/enc_fwd.py 2>/dev/null | tail -1 | cut -c1-100; done
