cd $GRAFT_REPO_ROOT
for v in pg0 pg1 pg2 pg3; do echo "== $v"; DICOW_HIP_LIB=tools/libv_$v.so python tools/profile_ntr.py 2>&1 | grep -E "gelu|N5120 K1280 plain"; done
echo "== attention ablations"
REPS=2 python tools/ab_attn.py base=tools/libva_base.so nofma=tools/libva_nofma.so nosum=tools/libva_nosum.so nofmasum=tools/libva_nofmasum.so noexp=tools/libva_noexp.so
echo "== enc fwd in-situ: new gelu (default lib) vs old (g0) vs none (g3)"
for r in 1 2; do
python tools/enc_fwd.py 20 | tail -1
DICOW_HIP_LIB=tools/libv_g0.so python tools/enc_fwd.py 20 | tail -1
DICOW_HIP_LIB=tools/libv_g2.so python tools/enc_fwd.py 20 | tail -1
DICOW_HIP_LIB=tools/libv_g3.so python tools/enc_fwd.py 20 | tail -1
done
