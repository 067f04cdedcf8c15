cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_realdims.py -m gpu -q -x -k "gemm or rd_base or rd_tiny" 2>&1 | tail -3
python tools/bench_base_shapes.py 2>/dev/null | tail -14
DICOW_HIP_LIB=$PWD/tools/libv_old128.so python tools/bench_base_shapes.py 2>/dev/null | tail -14 | cut -c1-75
for rep in 1 2; do
for v in "" tools/libv_old128.so; do
  if [ -n "$v" ]; then export DICOW_HIP_LIB=$PWD/$v; else unset DICOW_HIP_LIB; fi
  python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('base ${v:-new}', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
done
