cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_realdims.py tests/test_gpu_dp.py -q -k "f6 or f8 or scb or se or ctc_se" 2>&1 | tail -3
for i in 1 2; do python bench.py --se --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('se', d['value'], d['ms_per_step'], d['kernels'])" | cut -c1-120; done
