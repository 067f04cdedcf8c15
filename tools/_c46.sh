cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
python tools/bench_attn.py 2>/dev/null | grep attn_fwd | cut -c1-80
ATTN_LOG2=1 python tools/bench_attn.py 2>/dev/null | grep attn_fwd | sed 's/attn_fwd/attn_fwd[log2 fast]/' | cut -c1-80
done
