cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=5 timeout 900 python tools/ab_encfwd.py s0=$T/libv_s0.so s1=$T/libv_s1.so 2>&1 | grep -v amdgpu.ids | tail -3
