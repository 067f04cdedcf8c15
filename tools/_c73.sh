cd $GRAFT_REPO_ROOT
T=$PWD/tools
REPS=5 timeout 900 python tools/ab_encfwd.py e0=$T/libv_e0.so e1=$T/libv_e1.so 2>&1 | grep -v amdgpu.ids | tail -3
echo "--- without the per-layer FDDT (plain LayerNorm 1):"
ENC_NO_FDDT=1 REPS=5 timeout 900 python tools/ab_encfwd.py e0=$T/libv_e0.so e1=$T/libv_e1.so 2>&1 | grep -v amdgpu.ids | tail -3
