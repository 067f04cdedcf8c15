cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=3 timeout 900 python tools/ab_step.py base=$T/libv_base.so n1=$T/libv_n1.so n2=$T/libv_n2.so n2q=$T/libv_n2q.so n3=$T/libv_n3.so n4=$T/libv_n4.so n5=$T/libv_n5.so n6=$T/libv_n6.so n7=$T/libv_n7.so 2>&1 | grep -v amdgpu.ids | tail -12
