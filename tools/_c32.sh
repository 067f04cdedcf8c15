cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" tools/libv_big90.so tools/libv_big40.so; do
  if [ -n "$v" ]; then export DICOW_HIP_LIB=$PWD/$v; else unset DICOW_HIP_LIB; fi
  python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('base ${v:-shipped}', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
done
