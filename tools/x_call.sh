out=gpurun_out/r06x10; mkdir -p $out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $out/gpu_tests.txt
pr() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['ms_per_step_median'], 'roof', d['roofline']['frac'], 'train-fwd', (d.get('encoder_forward_train') or {}).get('ms'), (d.get('power') or {}).get('sclk_mhz_mean'))"; }
for rep in 1 2; do for v in all_on bwd_off all_off; do e="DICOW_SPLIT_BWD=1"; [ $v = bwd_off ] && e="DICOW_SPLIT_BWD=0"; [ $v = all_off ] && e="DICOW_SPLIT_FWD=0"
  env $e timeout 600 python bench.py --no-extra --no-cpu-baseline 2>$out/bench_$v.err | tail -1 > $out/bench_${v}_$rep.json; pr $out/bench_${v}_$rep.json $v | tee -a $out/ab.txt
done; done
