cd $GRAFT_REPO_ROOT
REPS=3 timeout 600 python tools/ab_attn.py p0=tools/libva_p0.so zs=tools/libva_zs.so > gpurun_out/c31_ab.txt 2>&1
ATTN_LOG2=1 REPS=3 timeout 600 python tools/ab_attn.py l2b=tools/libva_l2b.so >> gpurun_out/c31_ab.txt 2>&1
cat gpurun_out/c31_ab.txt
