out=gpurun_out/r06x1; mkdir -p $out
export TMPDIR=/tmp
DICOW_HIP_LIB=$PWD/tools/libv_fatom1.so timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn" 2>&1 | tail -5 | tee $out/attn_tests_fatom1.txt
DICOW_HIP_LIB=$PWD/tools/libv_fatom1.so timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "attention_at_bench_shape" 2>&1 | tail -5 | tee -a $out/attn_tests_fatom1.txt
for rep in 1 2; do for v in base fatom1 fatom2; do l=$PWD/ts-asr-whisper_amd/libdicow_hip.so; [ $v != base ] && l=$PWD/tools/libv_$v.so
  echo "== $v" | tee -a $out/attn_bench.txt; DICOW_HIP_LIB=$l ATTN_LOG2=1 ATTN_BWD_REPS=2 timeout 300 python tools/bench_attn.py 2>&1 | grep "attn_bwd" | tee -a $out/attn_bench.txt; done; done
