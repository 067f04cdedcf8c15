# scratch call: which stream launches a layer's pooled weight gradients under the two half-batch streams: alt (default) | main | third (a queue of its own)
out=gpurun_out/r06x15; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_forward.py -x -q 2>&1 | tail -n 5 | tee $out/tests.txt
pr() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['ms_per_step_median'], 'loss', d['loss'], (d.get('power') or {}).get('sclk_mhz_mean'))"; }
for rep in 1 2 3; do for v in alt third main; do
  DICOW_SPLIT_BWD_WGRAD=$v timeout 300 python bench.py --steps 15 --warmup 4 --no-extra --no-cpu-baseline --no-one-stream-ref --profile-steps 1 2>/dev/null | tail -n 1 > $out/b_${v}_$rep.json
  pr $out/b_${v}_$rep.json $v | tee -a $out/ab.txt
done; done
