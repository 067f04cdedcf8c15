cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=4 timeout 900 python tools/ab_encfwd.py base=$T/libv_base.so f1=$T/libv_f1.so f2=$T/libv_f2.so f3=$T/libv_f3.so 2>&1 | grep -v amdgpu.ids | tail -5
