cd $GRAFT_REPO_ROOT
run() {
  python bench.py --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('turbo $1', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['encoder_forward']['ms'], d['loss'] if 'loss' in d else '')"
  python tools/enc_fwd.py 2>/dev/null | tail -1 | cut -c1-100
}
for rep in 1 2; do
run log2
sed -i 's/^QK_LOG2 = True/QK_LOG2 = False/' ts-asr-whisper_amd/engine.py
run plain
sed -i 's/^QK_LOG2 = False/QK_LOG2 = True/' ts-asr-whisper_amd/engine.py
done
python bench.py --model whisper-base --batch 8 --graph --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('base log2', d['value'], d['ms_per_step'])"
