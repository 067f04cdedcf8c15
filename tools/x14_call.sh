# scratch call: CU budget of the persistent GEMMs under the two half-batch streams (each stream's launch limited to part of the chip)
out=gpurun_out/r06x14; mkdir -p $out
export TMPDIR=/tmp
pr() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['ms_per_step_median'], 'train-fwd', (d.get('encoder_forward_train') or {}).get('ms'), 'inf-fwd', (d.get('encoder_forward') or {}).get('ms'), (d.get('power') or {}).get('sclk_mhz_mean'))"; }
for rep in 1 2; do for c in 0 128 160 192 224; do
  timeout 300 python bench.py --gemm-cus $c --steps 12 --warmup 4 --no-extra --no-cpu-baseline --no-one-stream-ref 2>/dev/null | tail -n 1 > $out/b_${c}_$rep.json
  pr $out/b_${c}_$rep.json cus$c | tee -a $out/ab.txt
done; done
