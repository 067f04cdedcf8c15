cd $GRAFT_REPO_ROOT
export DICOW_HIP_LIB=$PWD/tools/libva_occ4.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -k "attn or attention" 2>&1 | tail -2
unset DICOW_HIP_LIB
REPS=4 timeout 900 python tools/ab_attn.py base=tools/libva_base.so occ4=tools/libva_occ4.so 2>&1 | tail -20
for v in base occ4; do DICOW_HIP_LIB=$PWD/tools/libva_$v.so python tools/enc_fwd.py 2>/dev/null | tail -1 | cut -c1-120; done
for v in base occ4; do DICOW_HIP_LIB=$PWD/tools/libva_$v.so python tools/enc_fwd.py 2>/dev/null | tail -1 | cut -c1-120; done
