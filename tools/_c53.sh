cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=4 timeout 900 python tools/ab_encfwd.py base=$T/libv_base.so bothnt=$T/libv_bothnt.so gnt=$T/libv_gnt.so ant=$T/libv_ant.so bnt=$T/libv_bnt.so c32nt=$T/libv_c32nt.so c16nt=$T/libv_c16nt.so rstnt=$T/libv_rstnt.so rynt=$T/libv_rynt.so 2>&1 | grep -v amdgpu.ids | tail -12
