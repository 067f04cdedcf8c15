cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=4 timeout 900 python tools/ab_encfwd.py base=$T/libv_base.so xs2=$T/libv_xs2.so xs5=$T/libv_xs5.so xs12=$T/libv_xs12.so xs30=$T/libv_xs30.so resnt=$T/libv_resnt.so rownt=$T/libv_rownt.so bothnt=$T/libv_bothnt.so 2>&1 | grep -v amdgpu.ids | tail -12
for v in prof profxs5 profxs12; do echo "== $v"; DICOW_HIP_LIB=$T/libv_$v.so timeout 300 python tools/profile_ntr.py 2>/dev/null | grep -A1 "N1280 K1280 res\|N1280 K5120 res\|N3840 K1280 plain\|geluinf" | cut -c1-220; done
