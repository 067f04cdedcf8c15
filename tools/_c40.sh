cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in old128 t1 t2 t3; do
  export DICOW_HIP_LIB=$PWD/tools/libv_$v.so
  python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('base $v', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
done
