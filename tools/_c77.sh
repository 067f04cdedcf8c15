cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=5 timeout 900 python tools/ab_encfwd.py g4=$T/libv_g4.so g2=$T/libv_g2.so 2>&1 | grep -v amdgpu.ids | tail -3
