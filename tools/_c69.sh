cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=5 timeout 900 python tools/ab_step.py w0=$T/libv_w0.so w2=$T/libv_w2.so w10=$T/libv_w10.so w14=$T/libv_w14.so 2>&1 | grep -v amdgpu.ids | tail -5
