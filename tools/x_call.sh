out=gpurun_out/r06x8; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_forward.py -x -q 2>&1 | tail -15 | tee $out/split_fwd_tests.txt
pr() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['ms_per_step_median'], 'roof', d['roofline']['frac'], 'train-fwd', (d.get('encoder_forward_train') or {}).get('ms'), (d.get('power') or {}).get('sclk_mhz_mean'))"; }
for rep in 1 2; do for v in dec1 dec0; do e=1; [ $v = dec0 ] && e=0
  DICOW_SPLIT_DEC=$e timeout 600 python bench.py --no-extra --no-cpu-baseline 2>$out/bench_$v.err | tail -1 > $out/bench_${v}_$rep.json; pr $out/bench_${v}_$rep.json $v | tee -a $out/ab.txt
done; done
for rep in 1 2; do for v in se1 se0; do e=1; [ $v = se0 ] && e=0
  DICOW_SPLIT_FWD=$e timeout 600 python bench.py --se --steps 8 --warmup 3 --no-extra --no-cpu-baseline 2>$out/bench_$v.err | tail -1 > $out/bench_${v}_$rep.json; pr $out/bench_${v}_$rep.json $v | tee -a $out/ab.txt
done; done
