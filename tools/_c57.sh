cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_realdims.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base graph', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; done
bash tools/prof_step.sh --model whisper-base --batch 8 2>&1 | grep -E "gemm_nt64|gemm_nt128t|Name" | cut -c1-140
