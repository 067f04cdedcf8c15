cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_realdims.py tests/test_gpu_model.py -q -k "deep_contraction or rd_ or gemm_nt or f7" 2>&1 | tail -4
for i in 1 2; do python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base graph', d['value'], d['ms_per_step'])"; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('turbo', d['value'], d['ms_per_step'], d['roofline']['frac'])"
