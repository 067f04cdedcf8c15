cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_realdims.py tests/test_gpu_graph.py tests/test_gpu_kernels.py -q -k "rd_ or f7 or f8 or graph or cast" 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('turbo', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
bash tools/prof_step.sh > /dev/null 2>&1; grep -E "cast_transpose|adamw|Fill|sumsq" gpurun_out/kernel_stats.csv | cut -c1-150
