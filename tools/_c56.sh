cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_base_shapes.py 2>&1 | grep -v amdgpu.ids
python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base graph', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
