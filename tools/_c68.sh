cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=3 timeout 900 python tools/ab_step.py w0=$T/libv_w0.so w1=$T/libv_w1.so w2=$T/libv_w2.so w4=$T/libv_w4.so w8=$T/libv_w8.so 2>&1 | grep -v amdgpu.ids | tail -6
