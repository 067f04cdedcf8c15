cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/c7_tests.txt 2>&1; tail -6 gpurun_out/c7_tests.txt
echo "== attention fwd variants"
REPS=2 python tools/ab_attn.py w2=tools/libva_w2.so w3=tools/libva_w3.so w2i=tools/libva_w2i.so w3i=tools/libva_w3i.so w3if=tools/libva_w3if.so w4i=tools/libva_w4i.so
