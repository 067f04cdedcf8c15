cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" tools/libva_base.so; do
  if [ -n "$v" ]; then export DICOW_HIP_LIB=$PWD/$v; else unset DICOW_HIP_LIB; fi
  python bench.py --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('turbo ${v:-occ4}', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['encoder_forward']['ms'])"
  python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('base ${v:-occ4}', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
done
