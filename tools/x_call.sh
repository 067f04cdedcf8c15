# scratch call: whisper-base B = 8 hipGraph step with the two half-batch streams forked INSIDE the capture (parallel graph branches)
out=gpurun_out/r06x12; mkdir -p $out
export TMPDIR=/tmp
pr() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d.get('value'), d.get('ms_per_step'), d.get('ms_per_step_median'), 'loss', d.get('loss'), d.get('error'))"; }
for rep in 1 2 3; do for v in none fwd fwd_dec all; do
  e="DICOW_SPLIT_IN_CAPTURE=0"
  [ $v = fwd ] && e="DICOW_SPLIT_IN_CAPTURE=1 DICOW_SPLIT_FWD_MIN_ROWS=6000 DICOW_SPLIT_DEC=0 DICOW_SPLIT_BWD=0"
  [ $v = fwd_dec ] && e="DICOW_SPLIT_IN_CAPTURE=1 DICOW_SPLIT_FWD_MIN_ROWS=6000 DICOW_SPLIT_BWD=0"
  [ $v = all ] && e="DICOW_SPLIT_IN_CAPTURE=1 DICOW_SPLIT_FWD_MIN_ROWS=6000"
  env $e timeout 300 python bench.py --model whisper-base --batch 8 --graph --no-extra --no-cpu-baseline --no-power --steps 30 --warmup 5 2>$out/err_${v}.txt | tail -n 1 > $out/b_${v}_$rep.json
  pr $out/b_${v}_$rep.json $v | tee -a $out/ab.txt; tail -n 3 $out/err_${v}.txt | cut -c1-300 >> $out/errs.txt
done; done
