cd $GRAFT_REPO_ROOT
T=$PWD/tools
for rep in 1 2; do for v in d2 d3; do DICOW_HIP_LIB=$T/libv_$v.so ATTN_LOG2=1 timeout 120 python tools/bench_attn.py 2>/dev/null | grep "attn_bwd" | sed "s/attn_bwd/$v bwd/" | cut -c1-90; done; done
