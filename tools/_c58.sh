cd $GRAFT_REPO_ROOT
T=$PWD/tools
for rep in 1 2; do
for v in nb k512 k1024 deep k512d off; do
DICOW_HIP_LIB=$T/libv_$v.so python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done; done
