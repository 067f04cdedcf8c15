# scratch call (timing only): what do the small reduction launches inside the chains cost on one stream and under the two half-batch streams?
out=gpurun_out/r06x16; mkdir -p $out
export TMPDIR=/tmp
pr() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['ms_per_step_median'], 'loss', d['loss'], (d.get('power') or {}).get('sclk_mhz_mean'))"; }
for rep in 1 2; do for v in base noreduce; do for s in 1 0; do
  l=$PWD/ts-asr-whisper_amd/libdicow_hip.so; [ $v = noreduce ] && l=$PWD/tools/libv_noreduce.so
  DICOW_HIP_LIB=$l DICOW_SPLIT_FWD=$s timeout 300 python bench.py --steps 15 --warmup 4 --no-extra --no-cpu-baseline --no-one-stream-ref --profile-steps 1 2>$out/err.txt | tail -n 1 > $out/b_${v}_${s}_$rep.json
  pr $out/b_${v}_${s}_$rep.json ${v}_split$s | tee -a $out/ab.txt
done; done; done
tail -n 3 $out/err.txt | cut -c1-300
