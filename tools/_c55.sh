cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3)
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['encoder_forward'])"
