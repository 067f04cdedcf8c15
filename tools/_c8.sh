cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -k "attn or attention or gelu" > gpurun_out/c8_tests.txt 2>&1; tail -6 gpurun_out/c8_tests.txt
timeout 1500 python -m pytest tests/test_gpu_realdims.py tests/test_gpu_model.py -q -x > gpurun_out/c8_tests2.txt 2>&1; tail -6 gpurun_out/c8_tests2.txt
echo "== attention fwd: plain mode (default lib) vs log2 mode variants"
REPS=2 python tools/ab_attn.py plain=ts-asr-whisper_amd/libdicow_hip.so
ATTN_LOG2=1 REPS=2 python tools/ab_attn.py msum1=tools/libva_msum1.so msum0=tools/libva_msum0.so msum1w3=tools/libva_msum1w3.so
echo "== enc fwd in-situ"
for r in 1 2 3; do python tools/enc_fwd.py 20 | tail -1; done
