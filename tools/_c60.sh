cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=3 timeout 900 python tools/ab_step.py base=$T/libv_base.so anl=$T/libv_anl.so ans=$T/libv_ans.so anb=$T/libv_anb.so p1=$T/libv_p1.so p2=$T/libv_p2.so p3=$T/libv_p3.so p4=$T/libv_p4.so 2>&1 | grep -v amdgpu.ids | tail -9
